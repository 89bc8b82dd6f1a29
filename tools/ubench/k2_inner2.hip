// k2_inner2.hip -- would a 512-thread workgroup (8 waves, 8x4 pairs per lane, 4 waves/SIMD at 2 blocks/CU)
// beat the shipped 256-thread / 8x8 / 3-blocks-per-CU structure?  Same LDS traffic pattern, no global traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__device__ __forceinline__ unsigned or_xor(unsigned d, unsigned a, unsigned b) { return __builtin_amdgcn_bitop3_b32(d, a, b, 0xF6); }

template <int THREADS, int NC, int MINW, bool BARRIER>
__global__ __launch_bounds__(THREADS, MINW) void k(unsigned *out, int iters) {
  __shared__ __attribute__((aligned(16))) uint4 lds[3 * 1024];   // 48 KiB ring
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 3 * 1024; i += THREADS) lds[i] = make_uint4(i * 2654435761u, i ^ 0x1234567, i * 40503u, ~i);
  __syncthreads();
  const int tx = (wave & 3) * 8 + (lane & 7), ty = (wave >> 2) * 8 + (lane >> 3);
  const int base_a = (ty & 15) * 4, base_b = 512 + (tx & 31) * 4, xa = (ty >> 2) & 3, xb = (tx >> 2) & 3;
  unsigned d[8][NC];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < NC; ++c) d[r][c] = 0;
  for (int it = 0; it < iters; ++it) {
    const uint4 *S = lds + (it % 3) * 1024;
    if (BARRIER) __syncthreads();
#pragma unroll 1
    for (int seg = 0; seg < 4; ++seg) {
      const uint4 *Sa = S + base_a + (seg ^ xa);
      const uint4 *Sb = S + base_b + (seg ^ xb);
      uint4 a[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) a[r] = Sa[r * 64];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const uint4 b = Sb[(c * 128) & 511];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          unsigned v = or_xor(d[r][c], a[r].x, b.y);
          v = or_xor(v, a[r].y, b.x);
          v = or_xor(v, a[r].z, b.w);
          d[r][c] = or_xor(v, a[r].w, b.z);
        }
      }
    }
  }
  unsigned acc = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc += d[r][c];
  out[blockIdx.x * THREADS + tid] = acc;
}

// 2-plane steps: operands read as 8 bytes (a: 16 registers instead of 32) so that 8x8 pairs per lane fit 128 VGPRs
// and 4 workgroups of 256 threads (4 waves per SIMD) are resident
template <int MINW>
__global__ __launch_bounds__(256, MINW) void k64(unsigned *out, int iters) {
  __shared__ __attribute__((aligned(16))) uint4 lds[3 * 768];   // 36 KiB ring (12-plane stages)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 3 * 768; i += 256) lds[i] = make_uint4(i * 2654435761u, i ^ 0x1234567, i * 40503u, ~i);
  __syncthreads();
  const int tx = (wave & 1) * 8 + (lane & 7), ty = (wave >> 1) * 8 + (lane >> 3);
  const int base_a = ty * 3, base_b = 384 + tx * 3;
  unsigned d[8][8];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) d[r][c] = 0;
  for (int it = 0; it < iters; ++it) {
    const uint2 *S = reinterpret_cast<const uint2 *>(lds + (it % 3) * 768);
    __syncthreads();
#pragma unroll 1
    for (int hs = 0; hs < 6; ++hs) {          // 6 half-segments of 2 planes = 12 planes
      const uint2 *Sa = S + 2 * base_a + hs;
      const uint2 *Sb = S + 2 * base_b + hs;
      uint2 a[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) a[r] = Sa[r * 96];
      uint2 b = Sb[0];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint2 bn = (c + 1 < 8) ? Sb[(c + 1) * 96] : b;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          unsigned v = or_xor(d[r][c], a[r].x, b.y);
          d[r][c] = or_xor(v, a[r].y, b.x);
        }
        b = bn;
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
void run(const char *name, K kern, int threads, int nc, int blocks_per_cu, unsigned *out) {
  const int iters = 2000, blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters / 4);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double ops = (double)blocks * threads * iters * (nc < 0 ? 6 * 8 * 8 * 2 : 4 * 8 * nc * 4);
  printf("%-52s blocks/CU=%d  %8.2f ms  %7.1f T lane-bitop3/s\n", name, blocks_per_cu, ms, ops / (ms * 1e-3) / 1e12);
}

int main() {
  unsigned *out; CHECK(hipMalloc(&out, 256 * 4 * 512 * sizeof(unsigned)));
  run("256 thr, 8x8/lane, barrier (shipped shape)", k<256, 8, 3, true>, 256, 8, 3, out);
  run("512 thr, 8x4/lane, barrier, 2 blocks/CU", k<512, 4, 4, true>, 512, 4, 2, out);
  run("512 thr, 8x4/lane, no barrier, 2 blocks/CU", k<512, 4, 4, false>, 512, 4, 2, out);
  run("256 thr, 8x4/lane, barrier, 4 blocks/CU", k<256, 4, 4, true>, 256, 4, 4, out);
  run("256 thr, 8x4/lane, barrier, 3 blocks/CU", k<256, 4, 4, true>, 256, 4, 3, out);
  run("512 thr, 8x8/lane (256 VGPR budget), 1 block/CU", k<512, 8, 2, true>, 512, 8, 1, out);
  run("256 thr, 8x16/lane (256 VGPR budget), 2 blocks/CU", k<256, 16, 2, true>, 256, 16, 2, out);
  run("256 thr, 8x12/lane (256 VGPR budget), 2 blocks/CU", k<256, 12, 2, true>, 256, 12, 2, out);
  run("256 thr, 8x8/lane, 2 blocks/CU (occupancy only)", k<256, 8, 2, true>, 256, 8, 2, out);
  run("256 thr, 8x8/lane, b64 operands, 12-plane stages, 4 blocks/CU", k64<4>, 256, -1, 4, out);
  run("256 thr, 8x8/lane, b64 operands, 12-plane stages, 3 blocks/CU", k64<3>, 256, -1, 3, out);
  run("256 thr, 8x8/lane, b64 operands, 12-plane stages, 2 blocks/CU", k64<2>, 256, -1, 2, out);
  return 0;
}
