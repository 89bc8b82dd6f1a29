// inst_rate.hip -- issue cost of individual gfx950 VALU instructions, in units of
// "wave64 v_add_u32" (8 waves/SIMD, 8 independent chains, registers only).
//   hipcc --offload-arch=gfx950 -O3 -o inst_rate inst_rate.hip && ./inst_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned long long u64;

// 8 chains: c_i = OP(c_i, a, b)
#define KERNEL3(NAME, ASM)                                                                         \
  __global__ void NAME(unsigned *out, int iters, unsigned seed) {                                  \
    unsigned c0 = threadIdx.x + seed, c1 = c0 * 3, c2 = c0 ^ 5, c3 = c0 + 7, c4 = c0 * 9, c5 = ~c0, c6 = c0 >> 1, c7 = c0 << 1; \
    unsigned a = blockIdx.x * 2654435761u + seed, b = threadIdx.x * 40503u + 1;                   \
    for (int i = 0; i < iters; ++i) {                                                              \
      _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                             \
        asm volatile(ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7)               \
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) \
                     : "v"(a), "v"(b));                                                            \
      }                                                                                            \
    }                                                                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;           \
  }
#define S(x) #x
#define A_ADD(d) "v_add_u32 " S(d) ", " S(d) ", %8\n\t"
#define A_XOR(d) "v_xor_b32 " S(d) ", " S(d) ", %8\n\t"
#define A_MIN(d) "v_min_u32 " S(d) ", " S(d) ", %8\n\t"
#define A_MAXI(d) "v_max_i32 " S(d) ", " S(d) ", %8\n\t"
#define A_SUB(d) "v_sub_u32 " S(d) ", " S(d) ", %8\n\t"
#define A_AND(d) "v_and_b32 " S(d) ", " S(d) ", %8\n\t"
#define A_ADD3(d) "v_add3_u32 " S(d) ", " S(d) ", %8, %9\n\t"
#define A_OR3(d) "v_or3_b32 " S(d) ", " S(d) ", %8, %9\n\t"
#define A_MAX3(d) "v_max3_i32 " S(d) ", " S(d) ", %8, %9\n\t"
#define A_BFI(d) "v_bfi_b32 " S(d) ", %8, " S(d) ", %9\n\t"
#define A_XAD(d) "v_xad_u32 " S(d) ", " S(d) ", %8, %9\n\t"
#define A_LSHLADD(d) "v_lshl_add_u32 " S(d) ", " S(d) ", 3, %9\n\t"
#define A_BITOP3(d) "v_bitop3_b32 " S(d) ", " S(d) ", %8, %9 bitop3:0xbe\n\t"   /* d | (a ^ b) */
#define A_BCNT(d) "v_bcnt_u32_b32 " S(d) ", %8, " S(d) "\n\t"
#define A_PKADD(d) "v_pk_add_u16 " S(d) ", " S(d) ", %8\n\t"
#define A_PKMIN(d) "v_pk_min_u16 " S(d) ", " S(d) ", %8\n\t"
#define A_PKMAXI(d) "v_pk_max_i16 " S(d) ", " S(d) ", %8\n\t"
#define A_MULLO(d) "v_mul_lo_u32 " S(d) ", " S(d) ", %8\n\t"
#define A_MAD24(d) "v_mad_u32_u24 " S(d) ", " S(d) ", %8, %9\n\t"
#define A_PERM(d) "v_perm_b32 " S(d) ", " S(d) ", %8, %9\n\t"
#define A_ALIGNBIT(d) "v_alignbit_b32 " S(d) ", " S(d) ", " S(d) ", 19\n\t"
#define A_SAD(d) "v_sad_u32 " S(d) ", %8, %9, " S(d) "\n\t"
#define A_SADU8(d) "v_sad_u8 " S(d) ", %8, %9, " S(d) "\n\t"
#define A_MED3(d) "v_med3_i32 " S(d) ", " S(d) ", %8, %9\n\t"
#define A_CMP_VCC_ADDC(d) "v_cmp_eq_u32_e32 vcc, %8, " S(d) "\n\tv_addc_co_u32_e32 " S(d) ", vcc, 0, " S(d) ", vcc\n\t"
#define A_CMP_VCC(d) "v_cmp_eq_u32_e32 vcc, %8, " S(d) "\n\t"
#define A_CMP_SGPR(d) "v_cmp_eq_u32_e64 s[20:21], %8, " S(d) "\n\t"
#define A_CNDMASK(d) "v_cndmask_b32_e64 " S(d) ", " S(d) ", %8, s[20:21]\n\t"
#define A_ADDC(d) "v_addc_co_u32_e64 " S(d) ", s[22:23], 0, " S(d) ", s[20:21]\n\t"
#define A_ADDCO(d) "v_add_co_u32_e64 " S(d) ", s[22:23], %8, " S(d) "\n\t"
#define A_MBCNT(d) "v_mbcnt_lo_u32_b32 " S(d) ", %8, " S(d) "\n\t"
#define A_DOT4(d) "v_dot4_u32_u8 " S(d) ", %8, %9, " S(d) "\n\t"
#define A_DOT2(d) "v_dot2_u32_u16 " S(d) ", %8, %9, " S(d) "\n\t"
#define A_CMPCLASS_NE(d) "v_cmp_ne_u32_e64 s[20:21], %8, " S(d) "\n\t"
#define A_SUBB(d) "v_subb_co_u32_e64 " S(d) ", s[22:23], " S(d) ", 0, s[20:21]\n\t"
#define A_FFBH(d) "v_ffbh_u32 " S(d) ", " S(d) "\n\t"
#define A_MINMAX(d) "v_min_i32 " S(d) ", " S(d) ", %8\n\tv_max_i32 " S(d) ", " S(d) ", %9\n\t"

KERNEL3(k_add, A_ADD) KERNEL3(k_xor, A_XOR) KERNEL3(k_min, A_MIN) KERNEL3(k_maxi, A_MAXI) KERNEL3(k_sub, A_SUB)
KERNEL3(k_and, A_AND) KERNEL3(k_add3, A_ADD3) KERNEL3(k_or3, A_OR3) KERNEL3(k_max3, A_MAX3) KERNEL3(k_bfi, A_BFI)
KERNEL3(k_xad, A_XAD) KERNEL3(k_lshladd, A_LSHLADD) KERNEL3(k_bitop3, A_BITOP3) KERNEL3(k_bcnt, A_BCNT)
KERNEL3(k_pkadd, A_PKADD) KERNEL3(k_pkmin, A_PKMIN) KERNEL3(k_pkmaxi, A_PKMAXI) KERNEL3(k_mullo, A_MULLO)
KERNEL3(k_mad24, A_MAD24) KERNEL3(k_perm, A_PERM) KERNEL3(k_alignbit, A_ALIGNBIT) KERNEL3(k_sad, A_SAD)
KERNEL3(k_sadu8, A_SADU8) KERNEL3(k_med3, A_MED3) KERNEL3(k_cmp_vcc_addc, A_CMP_VCC_ADDC) KERNEL3(k_cmp_vcc, A_CMP_VCC)
KERNEL3(k_cmp_sgpr, A_CMP_SGPR) KERNEL3(k_cndmask, A_CNDMASK) KERNEL3(k_addc, A_ADDC) KERNEL3(k_addco, A_ADDCO)
KERNEL3(k_mbcnt, A_MBCNT) KERNEL3(k_dot4, A_DOT4) KERNEL3(k_dot2, A_DOT2) KERNEL3(k_subb, A_SUBB) KERNEL3(k_ffbh, A_FFBH)

#define A_MAXF(d) "v_max_f32 " S(d) ", " S(d) ", %8\n\t"
#define A_MINF(d) "v_min_f32 " S(d) ", " S(d) ", %8\n\t"
#define A_MAX3F(d) "v_max3_f32 " S(d) ", " S(d) ", %8, %9\n\t"
#define A_MED3F(d) "v_med3_f32 " S(d) ", " S(d) ", %8, %9\n\t"
#define A_MAXIMUM3F(d) "v_maximum3_f32 " S(d) ", " S(d) ", %8, %9\n\t"
#define A_ADDF(d) "v_add_f32 " S(d) ", " S(d) ", %8\n\t"
#define A_FMAF(d) "v_fma_f32 " S(d) ", " S(d) ", %8, %9\n\t"
#define A_OR(d) "v_or_b32 " S(d) ", " S(d) ", %8\n\t"
#define A_LSHL(d) "v_lshlrev_b32 " S(d) ", 3, " S(d) "\n\t"
#define A_ASHR(d) "v_ashrrev_i32 " S(d) ", 3, " S(d) "\n\t"
#define A_MOV(d) "v_mov_b32 " S(d) ", %8\n\t"
#define A_MAXU(d) "v_max_u32 " S(d) ", " S(d) ", %8\n\t"
#define A_MAXI16(d) "v_max_i16 " S(d) ", " S(d) ", %8\n\t"
#define A_MAXU16(d) "v_max_u16 " S(d) ", " S(d) ", %8\n\t"
#define A_MAXF16(d) "v_max_f16 " S(d) ", " S(d) ", %8\n\t"
#define A_PKMAXF16(d) "v_pk_max_f16 " S(d) ", " S(d) ", %8\n\t"
#define A_ANDOR(d) "v_and_or_b32 " S(d) ", " S(d) ", %8, %9\n\t"
#define A_ADD_SGPR(d) "v_add_u32 " S(d) ", s20, " S(d) "\n\t"
#define A_ADD_LIT(d) "v_add_u32 " S(d) ", 0x12345, " S(d) "\n\t"
#define A_ADD_INL(d) "v_add_u32 " S(d) ", 17, " S(d) "\n\t"
#define A_MAXF_SGPR(d) "v_max_f32 " S(d) ", s20, " S(d) "\n\t"
#define A_CNDMASK_VCC(d) "v_cndmask_b32_e32 " S(d) ", " S(d) ", %8, vcc\n\t"
#define A_ADDMAXF(d) "v_add_u32 " S(d) ", " S(d) ", %8\n\tv_max_f32 " S(d) ", " S(d) ", %9\n\t"
#define A_BITOP3_PAR(d) "v_bitop3_b32 " S(d) ", " S(d) ", %8, %8 bitop3:0xbe\n\t"
KERNEL3(k_maxf, A_MAXF) KERNEL3(k_minf, A_MINF) KERNEL3(k_max3f, A_MAX3F) KERNEL3(k_med3f, A_MED3F) KERNEL3(k_maximum3f, A_MAXIMUM3F)
KERNEL3(k_addf, A_ADDF) KERNEL3(k_fmaf, A_FMAF) KERNEL3(k_or, A_OR) KERNEL3(k_lshl, A_LSHL) KERNEL3(k_ashr, A_ASHR) KERNEL3(k_mov, A_MOV)
KERNEL3(k_maxu, A_MAXU) KERNEL3(k_maxi16, A_MAXI16) KERNEL3(k_maxu16, A_MAXU16) KERNEL3(k_maxf16, A_MAXF16) KERNEL3(k_pkmaxf16, A_PKMAXF16)
KERNEL3(k_andor, A_ANDOR) KERNEL3(k_add_sgpr, A_ADD_SGPR) KERNEL3(k_add_lit, A_ADD_LIT) KERNEL3(k_add_inl, A_ADD_INL) KERNEL3(k_maxf_sgpr, A_MAXF_SGPR)
KERNEL3(k_cndmask_vcc, A_CNDMASK_VCC) KERNEL3(k_addmaxf, A_ADDMAXF)

typedef void (*kern_t)(unsigned *, int, unsigned);
double run(kern_t kern, int blocks, int iters, unsigned *out) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters / 4, 1u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e-3;
}

int main() {
  unsigned *out; CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(unsigned)));
  const int iters = 20000, blocks = 256 * 8;
  struct { const char *name; kern_t k; int inst; } list[] = {
      {"v_add_u32", k_add, 1}, {"v_xor_b32", k_xor, 1}, {"v_min_u32", k_min, 1}, {"v_max_i32", k_maxi, 1},
      {"v_sub_u32", k_sub, 1}, {"v_and_b32", k_and, 1}, {"v_add3_u32", k_add3, 1}, {"v_or3_b32", k_or3, 1},
      {"v_max3_i32", k_max3, 1}, {"v_med3_i32", k_med3, 1}, {"v_bfi_b32", k_bfi, 1}, {"v_xad_u32", k_xad, 1},
      {"v_lshl_add_u32", k_lshladd, 1}, {"v_bitop3_b32", k_bitop3, 1}, {"v_bcnt_u32_b32", k_bcnt, 1},
      {"v_mbcnt_lo_u32_b32", k_mbcnt, 1}, {"v_ffbh_u32", k_ffbh, 1},
      {"v_pk_add_u16", k_pkadd, 1}, {"v_pk_min_u16", k_pkmin, 1}, {"v_pk_max_i16", k_pkmaxi, 1},
      {"v_mul_lo_u32", k_mullo, 1}, {"v_mad_u32_u24", k_mad24, 1}, {"v_perm_b32", k_perm, 1},
      {"v_alignbit_b32", k_alignbit, 1}, {"v_sad_u32", k_sad, 1}, {"v_sad_u8", k_sadu8, 1},
      {"v_dot4_u32_u8", k_dot4, 1}, {"v_dot2_u32_u16", k_dot2, 1},
      {"v_cmp_eq_u32_e32 vcc", k_cmp_vcc, 1}, {"v_cmp_eq_u32_e64 sgpr", k_cmp_sgpr, 1},
      {"v_cndmask_b32 (sgpr sel)", k_cndmask, 1}, {"v_addc_co_u32 (sgpr in/out)", k_addc, 1},
      {"v_add_co_u32 (sgpr out)", k_addco, 1}, {"v_subb_co_u32", k_subb, 1},
      {"v_cmp_e32 vcc + v_addc_e32 (pair)", k_cmp_vcc_addc, 2},
      {"v_max_f32", k_maxf, 1}, {"v_min_f32", k_minf, 1}, {"v_max3_f32", k_max3f, 1}, {"v_med3_f32", k_med3f, 1},
      {"v_maximum3_f32", k_maximum3f, 1}, {"v_add_f32", k_addf, 1}, {"v_fma_f32", k_fmaf, 1}, {"v_or_b32", k_or, 1},
      {"v_lshlrev_b32", k_lshl, 1}, {"v_ashrrev_i32", k_ashr, 1}, {"v_mov_b32", k_mov, 1}, {"v_max_u32", k_maxu, 1},
      {"v_max_i16", k_maxi16, 1}, {"v_max_u16", k_maxu16, 1}, {"v_max_f16", k_maxf16, 1}, {"v_pk_max_f16", k_pkmaxf16, 1},
      {"v_and_or_b32", k_andor, 1}, {"v_add_u32 (sgpr src0)", k_add_sgpr, 1}, {"v_add_u32 (literal)", k_add_lit, 1},
      {"v_add_u32 (inline const)", k_add_inl, 1}, {"v_max_f32 (sgpr src0)", k_maxf_sgpr, 1}, {"v_cndmask_b32_e32 vcc", k_cndmask_vcc, 1},
      {"v_add_u32 + v_max_f32 (dependent pair)", k_addmaxf, 2},
  };
  double base = 0;
  printf("%-36s %12s %10s\n", "instruction", "Ginst*64/s", "cost(v_add=1)");
  for (auto &e : list) {
    double t = run(e.k, blocks, iters, out);
    double rate = (double)blocks * 256 * iters * 16 * 8 * e.inst / t;  // lane-instructions per second
    if (base == 0) base = rate;
    printf("%-36s %12.1f %10.2f\n", e.name, rate / 1e9, base / rate);
  }
  return 0;
}
