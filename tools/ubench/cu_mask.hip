// cu_mask.hip -- can the two halves of the pipelined duplicate route be given DISJOINT CUs?  (Today the compare's band kernel and the
// store-bound row expansion share every CU -- 2 + 1 workgroups -- and the expansion runs 16 % slower while the compare is resident.)
// hipExtStreamCreateWithCUMask: streams whose kernels may only run on the CUs of a bit mask.  This measures
//   (1) the row-fill kernel of store_bw (1024 threads, one row at a time) on masks of 8/8, 6/8, 5/8, 4/8, 3/8 of the CUs -- how many
//       CUs does the part's write rate need? -- with bit i of every group of 8 consecutive mask bits set for i < k;
//   (2) where those CUs are: XCC_ID / CU_ID (HW_ID registers) of the workgroups of a masked launch;
//   (3) the same fill while a VALU-bound kernel runs on the complementary mask.
//   hipcc --offload-arch=gfx950 -O3 -o cu_mask cu_mask.hip && ./cu_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(1024) void k_rows(double *out, int n, int64_t ld, double v, unsigned *ticket) {
  __shared__ int s_row;
  for (;;) {                                            // rows from a ticket counter: any number of resident workgroups
    __syncthreads();
    if (threadIdx.x == 0) s_row = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    const int r = s_row;
    if (r >= n) break;
    double *row = out + (int64_t)r * ld;
    for (int j2 = threadIdx.x; j2 < (n >> 1); j2 += 1024) {
      d2 x = {v + j2, v - j2};
      __builtin_nontemporal_store(x, reinterpret_cast<d2 *>(row + 2 * j2));
    }
  }
}
__global__ __launch_bounds__(256) void k_where(unsigned *xcc_cu) {   // one record per workgroup
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    xcc_cu[blockIdx.x] = ((xcc & 0xf) << 16) | ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4);   // XCC, CU_ID [11:8], SE_ID [15:13]
  }
  __builtin_amdgcn_s_sleep(100);
}
__global__ __launch_bounds__(256) void k_valu(unsigned *out, int iters) {   // VALU-bound filler (bitop3 chains), 4 workgroups per CU
  unsigned a = threadIdx.x, b = blockIdx.x * 3 + 1, c = 0x9e3779b9u, d = a ^ b;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) { a = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); b = __builtin_amdgcn_bitop3_b32(b, c, d, 0x96); c += a; d ^= b; }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}

static hipStream_t masked(int k, bool complement, int n_cus) {   // bits i of every 8 with i < k (or >= k)
  std::vector<uint32_t> m((n_cus + 31) / 32, 0u);
  for (int i = 0; i < n_cus; ++i) if (((i & 7) < k) != complement) m[i >> 5] |= 1u << (i & 31);
  hipStream_t s;
  CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data()));
  return s;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, n = 99328;
  double *out; CHECK(hipMalloc(&out, (size_t)n * n * 8));
  unsigned *ticket, *rec, *sink;
  CHECK(hipMalloc(&ticket, 64)); CHECK(hipMalloc(&rec, 4096 * 4)); CHECK(hipMalloc(&sink, 4096 * 256 * 4));
  printf("%d CUs; writing a %d x %d float64 matrix (%.1f GB) row by row\n", cus, n, n, (double)n * n * 8 / 1e9);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int k : {8, 6, 5, 4, 3}) {
    hipStream_t s = masked(k, false, cus);
    // where do the workgroups of a masked launch run?
    CHECK(hipMemsetAsync(rec, 0xff, 4096 * 4, s));
    hipLaunchKernelGGL(k_where, dim3(2048), dim3(256), 0, s, rec);
    std::vector<unsigned> h(2048);
    CHECK(hipMemcpyAsync(h.data(), rec, 2048 * 4, hipMemcpyDeviceToHost, s)); CHECK(hipStreamSynchronize(s));
    int per_xcc[16] = {0}; bool seen[16][8][16] = {};
    int distinct = 0;
    for (unsigned v : h) { const int x = (v >> 16) & 15, se = (v >> 4) & 7, cu = v & 15; if (!seen[x][se][cu]) { seen[x][se][cu] = true; ++distinct; ++per_xcc[x]; } }
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipMemsetAsync(ticket, 0, 4, s));
      CHECK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(k_rows, dim3(cus), dim3(1024), 0, s, out, n, (int64_t)n, 1.0, ticket);
      CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("mask %d/8: %3d distinct CUs seen (per XCC:", k, distinct);
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("); fill alone %.3f ms = %.2f TB/s", best, (double)n * n * 8 / 1e9 / best);
    if (k < 8) {   // the same fill with the VALU kernel on the complementary CUs
      hipStream_t c = masked(k, true, cus);
      hipLaunchKernelGGL(k_valu, dim3(4096), dim3(256), 0, c, sink, 6000);
      CHECK(hipMemsetAsync(ticket, 0, 4, s));
      CHECK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(k_rows, dim3(cus), dim3(1024), 0, s, out, n, (int64_t)n, 1.0, ticket);
      CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      CHECK(hipStreamSynchronize(c));
      printf("; beside a VALU-bound kernel on the other CUs %.3f ms", ms);
      CHECK(hipStreamDestroy(c));
    }
    printf("\n");
    CHECK(hipStreamDestroy(s));
  }
  // for comparison: both kernels on unmasked streams (sharing every CU)
  hipStream_t a, b; CHECK(hipStreamCreate(&a)); CHECK(hipStreamCreate(&b));
  hipLaunchKernelGGL(k_valu, dim3(4096), dim3(256), 0, b, sink, 6000);
  CHECK(hipMemsetAsync(ticket, 0, 4, a));
  CHECK(hipEventRecord(e0, a));
  hipLaunchKernelGGL(k_rows, dim3(cus), dim3(1024), 0, a, out, n, (int64_t)n, 1.0, ticket);
  CHECK(hipEventRecord(e1, a)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); CHECK(hipStreamSynchronize(b));
  printf("no masks, VALU-bound kernel on another stream: fill %.3f ms\n", ms);
  return 0;
}
