// nw_rows.hip -- the NW row block alone: the generated rows (tools/gen_nw_asm.py) against the compiled row (nw_row_ck as in nw_kernels.hip)
// under identical surroundings -- 256-thread workgroups at 4 waves per SIMD, a lane = one sequence2, a wave walks 16 sequence1 rows of a
// 64-row LDS tile, results folded into a checksum -- so that generator variants can be timed without the kernel around them.
//   hipcc --offload-arch=gfx950 -O3 -DNW_N=20 -DNW_INC='"../../dynaalign_amd/csrc/nw_rows_p20.inc"' -DNW_BIND='"../../dynaalign_amd/csrc/nw_rows_p20_bind.inc"' -o nw_rows nw_rows.hip
//   ./nw_rows            prints ms per 2*10^12 cells (the 100k headline set's direct sweep) for both forms + whether the checksums agree
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#ifndef NW_N
#define NW_N 20
#endif
#include NW_BIND
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
constexpr int N = NW_N, S = 13, S2 = 15, LB = 7, NEG = -24000;
constexpr int32_t PRI = 3 << S;

template <bool FIRST>
__device__ __forceinline__ void row_ck(int32_t (&VM)[N], int32_t (&XP)[N], const uint32_t (&boff)[N], const char *tab_row, int32_t vm_diag0,
                                       int32_t left0, int32_t kx, int32_t ky, int32_t ixf_first, int32_t pay_mask, int32_t pri_clear) {
  int32_t vmd = vm_diag0, vml = left0, ypl = left0;
#pragma unroll
  for (int c = 0; c < N; ++c) {
    const int32_t e = *reinterpret_cast<const int32_t *>(tab_row + boff[c]);
    const int32_t ixf = FIRST ? ixf_first : max(VM[c] + kx, XP[c]);
    const int32_t iyf = max(vml + ky, ypl);
    const int32_t vd = vmd + e;
    const int32_t w = max(max(vd, ixf), iyf);
    const int32_t vmn = w & pri_clear;
    vmd = VM[c];
    VM[c] = vmn;
    XP[c] = __builtin_amdgcn_bitop3_b32(ixf, vmn, pay_mask, 0xD8);
    ypl = __builtin_amdgcn_bitop3_b32(iyf, vmn, pay_mask, 0xD8);
    vml = vmn;
  }
}

template <int FORM>   // 0: compiled row (residue read per row), 1: generated rows, 2: compiled row with the row-ahead residue read of round 3 (scalar row offset)
__global__ __launch_bounds__(256, 4) void k_rows(const uint8_t *codes, int iters, int go, int ge, uint32_t *out) {
  __shared__ int32_t tabk[576];
  __shared__ uint8_t rowcodes[64][N];
  __shared__ int32_t rowlen[64];
  const int goe = go + ge;
  for (int e = threadIdx.x; e < 576; e += 256) {
    const int a = e / 24, b = e % 24;
    const int sc = (a == b) ? 5 : ((a * 7 + b * 3) % 9) - 4;
    tabk[e] = ((sc + 2 * ge) << S2) + (2 << S) + 1 + ((a == b) ? (1 << LB) : 0);
  }
  for (int r = threadIdx.x >> 2; r < 64; r += 64) {
    if ((threadIdx.x & 3) == 0) rowlen[r] = N;
    for (int q = threadIdx.x & 3; q < N; q += 4) rowcodes[r][q] = codes[(blockIdx.x * 64 + r) * N % 4096 + q] % 20;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t boff[N];
#pragma unroll
  for (int c = 0; c < N; ++c) boff[c] = (uint32_t)(codes[(lane * 131 + blockIdx.x * 17 + c * 29) % 4096] % 20) * 4u;
  __syncthreads();
  typedef __attribute__((address_space(3))) const uint8_t lds_u8_t;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    for (int rr = 0; rr < 16; ++rr) {
      const int lr = wave * 16 + rr;
      const int32_t m = rowlen[lr];
      int32_t VM[N], XP[N];
      if constexpr (FORM == 1) {
        const uint32_t m_s = __builtin_amdgcn_readfirstlane((uint32_t)m);
        const uint32_t rc_s = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8_t *)&rowcodes[lr][0]);
        const uint32_t tb_s = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8_t *)reinterpret_cast<const uint8_t *>(tabk));
        const int32_t kx_s = ((ge - goe) << S2) + (1 << S), ky_s = (ge - goe) << S2, pm_s = (1 << S) - 1, pc_s = ~PRI;
        const int32_t vi_s = (ge - go) << S2, l0_s = NEG << S2, xf_s = ((NEG - min(goe, ge)) << S2) + (1 << S);
        CAT(NW_ASM_DECL_, NW_N)
        asm volatile(
#include NW_INC
            : CAT(NW_ASM_OUTS_, NW_N)
            : [m] "s"(m_s), [rc] "s"(rc_s), [tb] "s"(tb_s), [kx] "s"(kx_s), [ky] "s"(ky_s), [pm] "s"(pm_s), [pc] "s"(pc_s),
              [vi] "s"(vi_s), [l0] "s"(l0_s), [xf] "s"(xf_s), CAT(NW_ASM_INS_, NW_N)
            : CAT(NW_ASM_CLOBBERS_, NW_N));
        CAT(NW_ASM_COPY_, NW_N)
      } else {
#pragma unroll
        for (int c = 0; c < N; ++c) { VM[c] = (ge - go) << S2; XP[c] = 0; }
        auto in_vgpr = [](int32_t x) { int32_t v; asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(x)); return v; };
        const int32_t kx = in_vgpr(((ge - goe) << S2) + (1 << S)), ky = in_vgpr((ge - goe) << S2);
        const int32_t pay_mask = in_vgpr((1 << S) - 1), pri_clear = in_vgpr(~PRI);
        const int32_t ixf_first = ((NEG - min(goe, ge)) << S2) + (1 << S);
        const char *tab_bytes = reinterpret_cast<const char *>(tabk);
        const uint32_t rc_addr = (uint32_t)(uintptr_t)(lds_u8_t *)&rowcodes[lr][0];
        uint32_t code_v = 0u;
        if (FORM == 2 && m > 0) asm volatile("ds_read_u8 %0, %1" : "=v"(code_v) : "v"(rc_addr) : "memory");
        for (int32_t r = 1; r <= m; ++r) {
          uint32_t row_off;
          if (FORM == 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(code_v) :: "memory");
            row_off = __builtin_amdgcn_readfirstlane(code_v) * 96u;
            asm volatile("ds_read_u8 %0, %1" : "=v"(code_v) : "v"(rc_addr + (uint32_t)(r < m ? r : r - 1)) : "memory");
          } else {
            row_off = (uint32_t)rowcodes[lr][r - 1] * 96u;
          }
          const char *tab_row = tab_bytes + row_off;
          const int32_t vm_diag0 = (r == 1) ? 0 : ((ge - go) << S2);
          const int32_t left0 = NEG << S2;
          if (r == 1) row_ck<true>(VM, XP, boff, tab_row, vm_diag0, left0, kx, ky, ixf_first, pay_mask, pri_clear);
          else row_ck<false>(VM, XP, boff, tab_row, vm_diag0, left0, kx, ky, ixf_first, pay_mask, pri_clear);
        }
      }
#pragma unroll
      for (int c = 0; c < N; ++c) acc = acc * 31u + (uint32_t)VM[c];
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
  const int blocks = 256 * 4 * 4, iters = argc > 1 ? atoi(argv[1]) : 8;
  uint8_t *h = (uint8_t *)malloc(8192), *d;
  unsigned s = 7;
  for (int i = 0; i < 8192; ++i) { s = s * 1664525u + 1013904223u; h[i] = (uint8_t)(s >> 24); }
  CHECK(hipMalloc(&d, 8192)); CHECK(hipMemcpy(d, h, 8192, hipMemcpyHostToDevice));
  uint32_t *o[3];
  for (int f = 0; f < 3; ++f) CHECK(hipMalloc(&o[f], blocks * 256 * 4));
  double ms[3];
  for (int pass = 0; pass < 2; ++pass)
    for (int f = 0; f < 3; ++f) {
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      CHECK(hipEventRecord(e0));
      if (f == 1) hipLaunchKernelGGL(k_rows<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 10, 4, o[1]);
      else if (f == 2) hipLaunchKernelGGL(k_rows<2>, dim3(blocks), dim3(256), 0, 0, d, iters, 10, 4, o[2]);
      else hipLaunchKernelGGL(k_rows<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 10, 4, o[0]);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float t; CHECK(hipEventElapsedTime(&t, e0, e1));
      ms[f] = t;
    }
  uint32_t *h0 = (uint32_t *)malloc(blocks * 256 * 4), *h1 = (uint32_t *)malloc(blocks * 256 * 4), *h2 = (uint32_t *)malloc(blocks * 256 * 4);
  CHECK(hipMemcpy(h0, o[0], blocks * 256 * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h1, o[1], blocks * 256 * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(h2, o[2], blocks * 256 * 4, hipMemcpyDeviceToHost));
  long bad = 0, bad2 = 0;
  for (int i = 0; i < blocks * 256; ++i) { bad += h0[i] != h1[i]; bad2 += h0[i] != h2[i]; }
  const double cells = (double)blocks * 256 * iters * 16 * N * N;
  printf("N=%d, per 2e12 cells: compiled %.1f ms, compiled with the row-ahead residue read %.1f ms, generated %.1f ms; checksums differ in %ld / %ld lanes\n",
         N, ms[0] * 2e12 / cells, ms[2] * 2e12 / cells, ms[1] * 2e12 / cells, bad2, bad);
  return 0;
}
