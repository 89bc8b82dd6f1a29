// bank_conflict.hip -- does the VGPR bank (index mod 4) of the three sources of v_bitop3_b32 matter?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define OP(D, A, B) "v_bitop3_b32 v" #D ", v" #D ", v" #A ", v" #B " bitop3:0xf6\n\t"
#define OPS(D, A, B) "v_bitop3_b32 v" #D ", v" #D ", s" #A ", v" #B " bitop3:0xf6\n\t"
#define OP2(D, A) "v_xor_b32 v" #D ", v" #D ", v" #A "\n\t"
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","s40","s41"
#define KERN(NAME, BODY)                                                           \
  __global__ void NAME(unsigned *out, int iters) {                                 \
    unsigned r = 0;                                                                \
    for (int i = 0; i < iters; ++i) {                                              \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) asm volatile(BODY ::: CLOB);   \
    }                                                                              \
    asm volatile("v_mov_b32 %0, v40" : "=v"(r));                                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                \
  }
// D regs: 40,44,48,52,56,60,64,68 (bank 0) ; 8 independent chains
// (d,a,b) banks = (0,1,2)
KERN(k_012, OP(40,41,42) OP(44,45,46) OP(48,49,50) OP(52,53,54) OP(56,57,58) OP(60,61,62) OP(64,65,66) OP(68,69,70))
// (0,0,1): a in bank 0 (another D-bank reg), b bank 1
KERN(k_001, OP(40,44,41) OP(44,48,45) OP(48,52,49) OP(52,56,53) OP(56,60,57) OP(60,64,61) OP(64,68,65) OP(68,40,69))
// (0,1,1)
KERN(k_011, OP(40,41,45) OP(44,45,49) OP(48,49,53) OP(52,53,57) OP(56,57,61) OP(60,61,65) OP(64,65,69) OP(68,69,41))
// (0,0,0)
KERN(k_000, OP(40,44,48) OP(44,48,52) OP(48,52,56) OP(52,56,60) OP(56,60,64) OP(60,64,68) OP(64,68,40) OP(68,40,44))
// sgpr a: (d bank0, s, b bank1) and (d bank0, s, b bank0)
KERN(k_s01, OPS(40,40,41) OPS(44,40,45) OPS(48,40,49) OPS(52,40,53) OPS(56,41,57) OPS(60,41,61) OPS(64,41,65) OPS(68,41,69))
KERN(k_s00, OPS(40,40,44) OPS(44,40,48) OPS(48,40,52) OPS(52,40,56) OPS(56,41,60) OPS(60,41,64) OPS(64,41,68) OPS(68,41,40))
// v_xor (2 sources): different / same bank
KERN(k_x01, OP2(40,41) OP2(44,45) OP2(48,49) OP2(52,53) OP2(56,57) OP2(60,61) OP2(64,65) OP2(68,69))
KERN(k_x00, OP2(40,44) OP2(44,48) OP2(48,52) OP2(52,56) OP2(56,60) OP2(60,64) OP2(64,68) OP2(68,40))
typedef void (*kern_t)(unsigned *, int);
int main() {
  unsigned *out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
  struct { const char *n; kern_t k; } L[] = {{"bitop3 banks d,a,b = 0,1,2", k_012}, {"bitop3 banks 0,0,1", k_001}, {"bitop3 banks 0,1,1", k_011},
    {"bitop3 banks 0,0,0", k_000}, {"bitop3 d=0, a=SGPR, b=1", k_s01}, {"bitop3 d=0, a=SGPR, b=0", k_s00}, {"v_xor banks 0,1", k_x01}, {"v_xor banks 0,0", k_x00}};
  const int iters = 20000, blocks = 256 * 8;
  for (auto &e : L) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, iters / 4);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-32s %7.1f T lane-ops/s\n", e.n, (double)blocks * 256 * iters * 64 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
