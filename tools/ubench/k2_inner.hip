// k2_inner.hip -- ceiling of the K2 inner structure (LDS b128 reads + v_bitop3) without
// global traffic.  Variants: how many LDS reads per 256 bitop3, barrier or not, waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ unsigned or_xor(unsigned d, unsigned a, unsigned b) { return __builtin_amdgcn_bitop3_b32(d, a, b, 0xF6); }

// NA = a-reads per seg (8 = real), NB = b-reads per seg (8 = real); BARRIER: __syncthreads every 4 segs
template <int NA, int NB, bool BARRIER, int MINW, bool SWAP = false, bool ILP = false>
__global__ __launch_bounds__(256, MINW) void k(unsigned *out, int iters) {
  __shared__ __attribute__((aligned(16))) uint4 lds[3 * 1024];   // 48 KiB
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tx = ((wave & 1) << 3) + (lane & 7), ty = ((wave >> 1) << 3) + (lane >> 3);
  for (int i = tid; i < 3 * 1024; i += 256) lds[i] = make_uint4(i * 2654435761u, i ^ 0x1234567, i * 40503u, ~i);
  __syncthreads();
  const int base_a = ty * 4, base_b = 512 + tx * 4, xa = (ty >> 2) & 3, xb = (tx >> 2) & 3;
  unsigned d[8][8];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) d[r][c] = 0;
  for (int it = 0; it < iters; ++it) {
    const uint4 *S = lds + (it % 3) * 1024;
    if (BARRIER) __syncthreads();
#pragma unroll 1
    for (int seg = 0; seg < 4; ++seg) {
      const uint4 *Sa = S + base_a + (seg ^ xa);
      const uint4 *Sb = S + base_b + (seg ^ xb);
      uint4 a[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) a[r] = (r < NA) ? Sa[r * 64] : a[r % (NA ? NA : 1)];
      if (NA == 0) {
#pragma unroll
        for (int r = 0; r < 8; ++r) a[r] = make_uint4(d[r][0] + seg, d[r][1], d[r][2], d[r][3]);
      }
      uint4 b;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (c < NB) b = Sb[c * 64];
        else if (NB == 0) b = make_uint4(d[0][c], d[1][c] + seg, d[2][c], d[3][c]);
        if (!ILP) {
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            unsigned v = or_xor(d[r][c], a[r].x, SWAP ? b.y : b.x);
            v = or_xor(v, a[r].y, SWAP ? b.x : b.y);
            v = or_xor(v, a[r].z, SWAP ? b.w : b.z);
            d[r][c] = or_xor(v, a[r].w, SWAP ? b.z : b.w);
          }
        } else {  // 8 independent chains between dependent ops
#pragma unroll
          for (int r = 0; r < 8; ++r) d[r][c] = or_xor(d[r][c], a[r].x, b.y);
#pragma unroll
          for (int r = 0; r < 8; ++r) d[r][c] = or_xor(d[r][c], a[r].y, b.x);
#pragma unroll
          for (int r = 0; r < 8; ++r) d[r][c] = or_xor(d[r][c], a[r].z, b.w);
#pragma unroll
          for (int r = 0; r < 8; ++r) d[r][c] = or_xor(d[r][c], a[r].w, b.z);
        }
      }
    }
  }
  unsigned acc = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc += d[r][c];
  out[blockIdx.x * 256 + tid] = acc;
}

template <typename K>
void run(const char *name, K kern, int blocks_per_cu, unsigned *out) {
  const int iters = 2000, blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters / 4);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double ops = (double)blocks * 256 * iters * 4 * 256;   // lane-bitop3 ops
  printf("%-44s blocks/CU=%d  %8.2f ms  %7.1f T lane-bitop3/s\n", name, blocks_per_cu, ms, ops / (ms * 1e-3) / 1e12);
}

int main() {
  unsigned *out; CHECK(hipMalloc(&out, 256 * 4 * 256 * sizeof(unsigned)));
  for (int bpc : {1, 2, 3}) {
    run("no LDS reads, no barrier", k<0, 0, false, 3>, bpc, out);
    run("8a+8b b128 reads/seg, no barrier", k<8, 8, false, 3>, bpc, out);
    run("8a+8b b128 reads/seg, barrier/4seg", k<8, 8, true, 3>, bpc, out);
    run("8a+8b swapped pairing, no barrier", k<8, 8, false, 3, true>, bpc, out);
    run("8a+8b swapped pairing, barrier", k<8, 8, true, 3, true>, bpc, out);
    run("8a+8b swapped + ILP order, barrier", k<8, 8, true, 3, true, true>, bpc, out);
    run("no LDS, swapped + ILP order", k<0, 0, false, 3, true, true>, bpc, out);
  }
  return 0;
}
