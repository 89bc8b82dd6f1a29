// bank_parity.hip -- does the parity (VGPR bank) of an instruction's source registers change its issue rate on gfx950?
// DESIGN.md (round 1) found v_bitop3_b32 at half rate when its three sources are VGPRs of equal index parity; this asks the same of
// the other instructions of the NW cell (v_max3_i32, v_max_i32, v_add_u32, v_and_b32) with explicit registers:
// 8 chains d_i = OP(d_i, A, B) with d_i in v0,v2,..,v14 (even) or v1,v3,..,v15 (odd) and A / B even or odd.
//   hipcc --offload-arch=gfx950 -O3 -o bank_parity bank_parity.hip && ./bank_parity
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v20","v21","v22","v23"
// D0..D7: destination = first source; A, B: the other sources
#define BODY3(OP, D0,D1,D2,D3,D4,D5,D6,D7, A, B, EXTRA) \
  OP " " D0 ", " D0 ", " A ", " B EXTRA "\n\t" OP " " D1 ", " D1 ", " A ", " B EXTRA "\n\t" OP " " D2 ", " D2 ", " A ", " B EXTRA "\n\t" OP " " D3 ", " D3 ", " A ", " B EXTRA "\n\t" \
  OP " " D4 ", " D4 ", " A ", " B EXTRA "\n\t" OP " " D5 ", " D5 ", " A ", " B EXTRA "\n\t" OP " " D6 ", " D6 ", " A ", " B EXTRA "\n\t" OP " " D7 ", " D7 ", " A ", " B EXTRA "\n\t"
#define BODY2(OP, D0,D1,D2,D3,D4,D5,D6,D7, A) \
  OP " " D0 ", " D0 ", " A "\n\t" OP " " D1 ", " D1 ", " A "\n\t" OP " " D2 ", " D2 ", " A "\n\t" OP " " D3 ", " D3 ", " A "\n\t" \
  OP " " D4 ", " D4 ", " A "\n\t" OP " " D5 ", " D5 ", " A "\n\t" OP " " D6 ", " D6 ", " A "\n\t" OP " " D7 ", " D7 ", " A "\n\t"
#define EVEN "v0","v2","v4","v6","v8","v10","v12","v14"
#define ODD "v1","v3","v5","v7","v9","v11","v13","v15"

#define KERNEL(NAME, TEXT)                                                                  \
  __global__ void NAME(unsigned *out, int iters, unsigned seed) {                           \
    asm volatile("v_mov_b32 v0, %0\n\tv_mov_b32 v1, %0\n\tv_mov_b32 v2, %0\n\tv_mov_b32 v3, %0\n\tv_mov_b32 v4, %0\n\tv_mov_b32 v5, %0\n\tv_mov_b32 v6, %0\n\tv_mov_b32 v7, %0\n\t" \
                 "v_mov_b32 v8, %0\n\tv_mov_b32 v9, %0\n\tv_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\tv_mov_b32 v12, %0\n\tv_mov_b32 v13, %0\n\tv_mov_b32 v14, %0\n\tv_mov_b32 v15, %0\n\t" \
                 "v_mov_b32 v20, %1\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %2\n\tv_mov_b32 v23, %2" :: "v"(threadIdx.x + seed), "v"(blockIdx.x * 7u + 1u), "v"(threadIdx.x * 3u + 5u) : CLOB); \
    for (int i = 0; i < iters; ++i) {                                                       \
      asm volatile(TEXT TEXT TEXT TEXT TEXT TEXT TEXT TEXT TEXT TEXT TEXT TEXT TEXT TEXT TEXT TEXT ::: CLOB);  \
    }                                                                                       \
    unsigned r;                                                                             \
    asm volatile("v_add_u32 %0, v0, v1\n\tv_add_u32 %0, %0, v2\n\tv_add_u32 %0, %0, v3\n\tv_add_u32 %0, %0, v15\n\tv_add_u32 %0, %0, v14" : "=v"(r) :: CLOB); \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                         \
  }

#define X3(OP, EXTRA, TAG)                                                                                         \
  KERNEL(k_##TAG##_eee, BODY3(OP, "v0","v2","v4","v6","v8","v10","v12","v14", "v20", "v22", EXTRA))               \
  KERNEL(k_##TAG##_eeo, BODY3(OP, "v0","v2","v4","v6","v8","v10","v12","v14", "v20", "v23", EXTRA))               \
  KERNEL(k_##TAG##_eoo, BODY3(OP, "v0","v2","v4","v6","v8","v10","v12","v14", "v21", "v23", EXTRA))               \
  KERNEL(k_##TAG##_ooo, BODY3(OP, "v1","v3","v5","v7","v9","v11","v13","v15", "v21", "v23", EXTRA))
#define X2(OP, TAG)                                                                                                \
  KERNEL(k_##TAG##_ee, BODY2(OP, "v0","v2","v4","v6","v8","v10","v12","v14", "v20"))                              \
  KERNEL(k_##TAG##_eo, BODY2(OP, "v0","v2","v4","v6","v8","v10","v12","v14", "v21"))                              \
  KERNEL(k_##TAG##_oo, BODY2(OP, "v1","v3","v5","v7","v9","v11","v13","v15", "v21"))
X3("v_max3_i32", "", max3) X3("v_bitop3_b32", " bitop3:0xd8", bitop3) X3("v_add3_u32", "", add3)
X2("v_add_u32", add) X2("v_max_i32", max) X2("v_and_b32", and)
// bank = index mod 4?  sources v20 (0 mod 4) / v22 (2 mod 4) with destinations 0 mod 4 only
KERNEL(k_bitop3_mod4_same, BODY3("v_bitop3_b32", "v0","v4","v8","v12","v0","v4","v8","v12", "v20", "v20", " bitop3:0xd8"))
KERNEL(k_bitop3_mod4_diff, BODY3("v_bitop3_b32", "v0","v4","v8","v12","v0","v4","v8","v12", "v22", "v22", " bitop3:0xd8"))

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
  struct { const char *name; kern_t k; } list[] = {
      {"v_add_u32  dst/src0 even, src1 even", k_add_ee}, {"v_add_u32  even, odd", k_add_eo}, {"v_add_u32  odd, odd", k_add_oo},
      {"v_max_i32  even, even", k_max_ee}, {"v_max_i32  even, odd", k_max_eo}, {"v_max_i32  odd, odd", k_max_oo},
      {"v_and_b32  even, even", k_and_ee}, {"v_and_b32  even, odd", k_and_eo}, {"v_and_b32  odd, odd", k_and_oo},
      {"v_max3_i32 even, even, even", k_max3_eee}, {"v_max3_i32 even, even, odd", k_max3_eeo}, {"v_max3_i32 even, odd, odd", k_max3_eoo}, {"v_max3_i32 odd, odd, odd", k_max3_ooo},
      {"v_bitop3   even, even, even", k_bitop3_eee}, {"v_bitop3   even, even, odd", k_bitop3_eeo}, {"v_bitop3   even, odd, odd", k_bitop3_eoo}, {"v_bitop3   odd, odd, odd", k_bitop3_ooo},
      {"v_add3_u32 even, even, even", k_add3_eee}, {"v_add3_u32 even, even, odd", k_add3_eeo}, {"v_add3_u32 even, odd, odd", k_add3_eoo}, {"v_add3_u32 odd, odd, odd", k_add3_ooo},
      {"v_bitop3   all sources = 0 mod 4", k_bitop3_mod4_same}, {"v_bitop3   0 mod 4, 2 mod 4, 2 mod 4", k_bitop3_mod4_diff},
  };
  double base = 0;
  printf("%-44s %12s %10s\n", "instruction, source register parities", "Ginst*64/s", "cost(v_add=1)");
  for (auto &e : list) {
    double t = run(e.k, blocks, iters, out);
    double rate = (double)blocks * 256 * iters * 16 * 8 / t;
    if (base == 0) base = rate;
    printf("%-44s %12.1f %10.2f\n", e.name, rate / 1e9, base / rate);
  }
  return 0;
}
