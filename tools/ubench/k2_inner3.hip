// k2_inner3.hip -- operand-read scheduling variants for the K2 inner loop (256 threads, 8x8 pairs/lane,
// 3 blocks/CU, 48 KiB ring, barrier per 16 planes):
//   A: shipped before: 8 a-reads then b one ahead            (b128)
//   B: b0 first, then a, b one ahead (shipped now)           (b128)
//   C: 2-plane steps, both operands double-buffered in registers (b64)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__device__ __forceinline__ unsigned or_xor(unsigned d, unsigned a, unsigned b) { return __builtin_amdgcn_bitop3_b32(d, a, b, 0xF6); }

template <int VAR>
__global__ __launch_bounds__(256, 3) void k(unsigned *out, int iters) {
  __shared__ __attribute__((aligned(16))) uint4 lds[3 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 3 * 1024; i += 256) lds[i] = make_uint4(i * 2654435761u, i ^ 0x1234567, i * 40503u, ~i);
  __syncthreads();
  const int tx = ((wave & 1) << 3) + (lane & 7), ty = ((wave >> 1) << 3) + (lane >> 3);
  const int base_a = ty * 4, base_b = 512 + tx * 4, xa = (ty >> 2) & 3, xb = (tx >> 2) & 3;
  unsigned d[8][8];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) d[r][c] = 0;
  for (int it = 0; it < iters; ++it) {
    const uint4 *S = lds + (it % 3) * 1024;
    __syncthreads();
    if (VAR <= 1) {
#pragma unroll 1
      for (int seg = 0; seg < 4; ++seg) {
        const uint4 *Sa = S + base_a + (seg ^ xa);
        const uint4 *Sb = S + base_b + (seg ^ xb);
        uint4 a[8], b;
        if (VAR == 1) b = Sb[0];
#pragma unroll
        for (int r = 0; r < 8; ++r) a[r] = Sa[r * 64];
        if (VAR == 0) b = Sb[0];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const uint4 bn = (c + 1 < 8) ? Sb[(c + 1) * 64] : b;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            unsigned v = or_xor(d[r][c], a[r].x, b.y);
            v = or_xor(v, a[r].y, b.x);
            v = or_xor(v, a[r].z, b.w);
            d[r][c] = or_xor(v, a[r].w, b.z);
          }
          b = bn;
        }
      }
    } else {
      // 8 half-segments of 2 planes; operands of step h+1 are read while step h computes
      const uint2 *S2 = reinterpret_cast<const uint2 *>(S);
      uint2 a0[8], a1[8], b0[8], b1[8];
      auto load = [&](int h, uint2 (&av)[8], uint2 (&bv)[8]) {
        const int seg = h >> 1, half = h & 1;
#pragma unroll
        for (int r = 0; r < 8; ++r) av[r] = S2[2 * (base_a + (seg ^ xa) + r * 64) + half];
#pragma unroll
        for (int c = 0; c < 8; ++c) bv[c] = S2[2 * (base_b + (seg ^ xb) + c * 64) + (half ^ 0)];
      };
      auto comp = [&](const uint2 (&av)[8], const uint2 (&bv)[8]) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int r = 0; r < 8; ++r) d[r][c] = or_xor(or_xor(d[r][c], av[r].x, bv[c].y), av[r].y, bv[c].x);
      };
      load(0, a0, b0);
#pragma unroll 1
      for (int h = 0; h < 8; h += 2) {
        load(h + 1, a1, b1);
        comp(a0, b0);
        if (h + 2 < 8) load(h + 2, a0, b0);
        comp(a1, b1);
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
void run(const char *name, K kern, unsigned *out) {
  const int iters = 2000, blocks = 256 * 3;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters / 4);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double ops = (double)blocks * 256 * iters * 4 * 256;
  printf("%-60s %8.2f ms  %7.1f T lane-bitop3/s\n", name, ms, ops / (ms * 1e-3) / 1e12);
}

int main() {
  unsigned *out; CHECK(hipMalloc(&out, 256 * 4 * 256 * sizeof(unsigned)));
  run("A: a-reads, then b one ahead (b128)", k<0>, out);
  run("B: b0 first, then a, b one ahead (b128)  [shipped]", k<1>, out);
  run("C: 2-plane steps, operands double-buffered (b64)", k<2>, out);
  run("A again", k<0>, out);
  return 0;
}
