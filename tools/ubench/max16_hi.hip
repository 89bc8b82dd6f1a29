// max16_hi.hip -- can the NW gap-state maxes run as FULL-RATE 16-bit maxes on the HIGH half of a 32-bit key?
// (v_max_i16 is full rate on gfx950, v_max_i32 is not: profiles/r04_a_ubench_inst_rate.txt.)  Checks, for the op_sel and SDWA
// forms of v_max_i16 on the high halves: (1) semantics -- result.hi = max(a.hi, b.hi) signed, result.lo = the destination's old low
// half (preserved) -- and (2) the issue rate against v_add_u32.
//   hipcc --offload-arch=gfx950 -O3 -o max16_hi max16_hi.hip && ./max16_hi
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k_sem(const unsigned *a, const unsigned *b, unsigned *o_opsel, unsigned *o_sdwa, unsigned *o_opsel3, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = a[i], y = b[i], d1 = x, d2 = x, d3 = 0xdead0000u | (x & 0xffffu);
  asm volatile("v_max3_i16 %0, %0, %1, %1 op_sel:[1,1,1,1]" : "+v"(d1) : "v"(y));
  asm volatile("v_max_i16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(d2) : "v"(y));
  asm volatile("v_max3_i16 %0, %1, %2, %2 op_sel:[1,1,1,1]" : "+v"(d3) : "v"(x), "v"(y));   // dst differs from both sources: its low half must survive
  o_opsel[i] = d1; o_sdwa[i] = d2; o_opsel3[i] = d3;
}

#define S(x) #x
#define KERNEL(NAME, ASM)                                                                          \
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
#define A_ADD(d) "v_add_u32 " S(d) ", " S(d) ", %8\n\t"
#define A_MAX16(d) "v_max_i16 " S(d) ", " S(d) ", %8\n\t"
#define A_MAX16_E64(d) "v_max_i16_e64 " S(d) ", " S(d) ", %8\n\t"
#define A_MAX16_HI(d) "v_max3_i16 " S(d) ", " S(d) ", %8, %8 op_sel:[1,1,1,1]\n\t"
#define A_MAX16_SDWA(d) "v_max_i16_sdwa " S(d) ", " S(d) ", %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n\t"
#define A_MAX32_SDWA(d) "v_max_i32_sdwa " S(d) ", " S(d) ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
#define A_ADD_SDWA(d) "v_add_u32_sdwa " S(d) ", " S(d) ", %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n\t"
#define A_ADD16_HI(d) "v_add_i16 " S(d) ", " S(d) ", %8 op_sel:[1,1,1]\n\t"
#define A_PKMAX(d) "v_pk_max_i16 " S(d) ", " S(d) ", %8\n\t"
#define A_MAX3_I16(d) "v_max3_i16 " S(d) ", " S(d) ", %8, %9\n\t"
#define A_MAX3_I16_HI(d) "v_max3_i16 " S(d) ", " S(d) ", %8, %9 op_sel:[1,1,1,1]\n\t"
#define A_MAX32(d) "v_max_i32 " S(d) ", " S(d) ", %8\n\t"
#define A_CELL(d) "v_add_u32 " S(d) ", " S(d) ", %8\n\tv_max_i16_sdwa " S(d) ", " S(d) ", %9 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n\t"
KERNEL(k_add, A_ADD) KERNEL(k_max16, A_MAX16) KERNEL(k_max16_e64, A_MAX16_E64) KERNEL(k_max16_hi, A_MAX16_HI) KERNEL(k_max16_sdwa, A_MAX16_SDWA)
KERNEL(k_max32_sdwa, A_MAX32_SDWA) KERNEL(k_add_sdwa, A_ADD_SDWA) KERNEL(k_add16_hi, A_ADD16_HI) KERNEL(k_pkmax, A_PKMAX)
KERNEL(k_max3_i16, A_MAX3_I16) KERNEL(k_max3_i16_hi, A_MAX3_I16_HI) KERNEL(k_max32, A_MAX32) KERNEL(k_cell, A_CELL)

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
  const int n = 1 << 16;
  unsigned *ha = (unsigned *)malloc(n * 4), *hb = (unsigned *)malloc(n * 4), *h1 = (unsigned *)malloc(n * 4), *h2 = (unsigned *)malloc(n * 4), *h3 = (unsigned *)malloc(n * 4);
  unsigned s = 12345;
  for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; ha[i] = s; s = s * 1664525u + 1013904223u; hb[i] = s; }
  ha[0] = 0x7fff1234; hb[0] = 0x8000abcd; ha[1] = 0x8000ffff; hb[1] = 0x80000000; ha[2] = 0xffff0000; hb[2] = 0x0000ffff;
  unsigned *da, *db, *d1, *d2, *d3;
  CHECK(hipMalloc(&da, n * 4)); CHECK(hipMalloc(&db, n * 4)); CHECK(hipMalloc(&d1, n * 4)); CHECK(hipMalloc(&d2, n * 4)); CHECK(hipMalloc(&d3, n * 4));
  CHECK(hipMemcpy(da, ha, n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb, n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_sem, dim3(n / 256), dim3(256), 0, 0, da, db, d1, d2, d3, n);
  CHECK(hipMemcpy(h1, d1, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h2, d2, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h3, d3, n * 4, hipMemcpyDeviceToHost));
  long bad1 = 0, bad2 = 0, bad3 = 0;
  for (int i = 0; i < n; ++i) {
    short ah = (short)(ha[i] >> 16), bh = (short)(hb[i] >> 16);
    unsigned hi = (unsigned)(unsigned short)(ah > bh ? ah : bh) << 16;
    unsigned want = hi | (ha[i] & 0xffffu);
    bad1 += h1[i] != want; bad2 += h2[i] != want; bad3 += h3[i] != want;
  }
  printf("semantics: result.hi = max_i16(a.hi, b.hi), result.lo = old dst.lo over %d random pairs: v_max3_i16 op_sel form %ld wrong, SDWA form %ld wrong, op_sel with a third dst register %ld wrong\n", n, bad1, bad2, bad3);
  printf("  e.g. a=%08x b=%08x -> op_sel %08x sdwa %08x\n", ha[0], hb[0], h1[0], h2[0]);

  unsigned *out; CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(unsigned)));
  const int iters = 20000, blocks = 256 * 8;
  struct { const char *name; kern_t k; int inst; } list[] = {
      {"v_add_u32", k_add, 1}, {"v_max_i32", k_max32, 1}, {"v_max_i16 (e32, low halves)", k_max16, 1}, {"v_max_i16_e64", k_max16_e64, 1},
      {"v_max3_i16 a,b,b op_sel:[1,1,1,1] (high halves)", k_max16_hi, 1}, {"v_max_i16_sdwa WORD_1 preserve", k_max16_sdwa, 1},
      {"v_max_i32_sdwa DWORD", k_max32_sdwa, 1}, {"v_add_u32_sdwa WORD_1 preserve", k_add_sdwa, 1}, {"v_add_i16 op_sel:[1,1,1]", k_add16_hi, 1},
      {"v_pk_max_i16", k_pkmax, 1}, {"v_max3_i16", k_max3_i16, 1}, {"v_max3_i16 op_sel hi", k_max3_i16_hi, 1},
      {"v_add_u32 + v_max_i16_sdwa hi (dependent pair)", k_cell, 2},
  };
  double base = 0;
  printf("%-44s %12s %10s\n", "instruction", "Ginst*64/s", "cost(v_add=1)");
  for (auto &e : list) {
    double t = run(e.k, blocks, iters, out);
    double rate = (double)blocks * 256 * iters * 16 * 8 * e.inst / t;
    if (base == 0) base = rate;
    printf("%-44s %12.1f %10.2f\n", e.name, rate / 1e9, base / rate);
  }
  return 0;
}
