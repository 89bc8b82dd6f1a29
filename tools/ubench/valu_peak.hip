// valu_peak.hip -- what does the integer VALU of gfx950 actually sustain for the
// "count equal u32 pairs" inner loop?  Register-only loops (no memory), several
// candidate instruction sequences, swept over waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_peak valu_peak.hip && ./valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef unsigned long long u64;

// variant 0: v_cmp_eq_u32 -> sgpr pair ; v_addc_co_u32 acc += carry      (2 VALU / compare)
// 4 independent chains so the VALU-writes-SGPR -> VALU-reads-SGPR hazard is covered
__global__ void k_cmp_addc(unsigned *out, int iters, unsigned seed) {
  unsigned a0 = threadIdx.x * 7 + seed, a1 = a0 ^ 0x55, a2 = a0 + 3, a3 = a0 * 5;
  unsigned b0 = blockIdx.x + seed, b1 = b0 + 1, b2 = b0 ^ 9, b3 = b0 * 3;
  unsigned c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      u64 m0, m1, m2, m3, d0, d1, d2, d3;
      asm volatile(
          "v_cmp_eq_u32_e64 %4, %12, %16\n\t"
          "v_cmp_eq_u32_e64 %5, %13, %17\n\t"
          "v_cmp_eq_u32_e64 %6, %14, %18\n\t"
          "v_cmp_eq_u32_e64 %7, %15, %19\n\t"
          "v_addc_co_u32_e64 %0, %8, 0, %0, %4\n\t"
          "v_addc_co_u32_e64 %1, %9, 0, %1, %5\n\t"
          "v_addc_co_u32_e64 %2, %10, 0, %2, %6\n\t"
          "v_addc_co_u32_e64 %3, %11, 0, %3, %7\n\t"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3),
            "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3)
          : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
    }
    a0 += c0; b1 += c1;  // keep the loop from being hoisted
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
}

// variant 1: what hipcc emits for acc += (a==b): cmp, cmp, cndmask, addc   (2 VALU / compare)
__global__ void k_cmp_cnd_addc(unsigned *out, int iters, unsigned seed) {
  unsigned a0 = threadIdx.x * 7 + seed, a1 = a0 ^ 0x55, a2 = a0 + 3, a3 = a0 * 5;
  unsigned b0 = blockIdx.x + seed, b1 = b0 + 1, b2 = b0 ^ 9, b3 = b0 * 3;
  unsigned c0 = 0, c1 = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      u64 m0, m1, m2, m3, d0, d1; unsigned t0, t1;
      asm volatile(
          "v_cmp_eq_u32_e64 %4, %10, %14\n\t"
          "v_cmp_eq_u32_e64 %5, %11, %15\n\t"
          "v_cmp_eq_u32_e64 %6, %12, %16\n\t"
          "v_cmp_eq_u32_e64 %7, %13, %17\n\t"
          "v_cndmask_b32_e64 %2, 0, 1, %4\n\t"
          "v_cndmask_b32_e64 %3, 0, 1, %5\n\t"
          "v_addc_co_u32_e64 %0, %8, %0, %2, %6\n\t"
          "v_addc_co_u32_e64 %1, %9, %1, %3, %7\n\t"
          : "+v"(c0), "+v"(c1), "=&v"(t0), "=&v"(t1), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(d0), "=&s"(d1)
          : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
    }
    a0 += c0; b1 += c1;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1;
}

// variant 2: 16-bit ids, two per dword: v_xor, v_pk_min_u16(x,1), v_pk_add_u16   (1.5 VALU / compare)
__global__ void k_pk16(unsigned *out, int iters, unsigned seed) {
  unsigned a0 = threadIdx.x * 7 + seed, a1 = a0 ^ 0x55, a2 = a0 + 3, a3 = a0 * 5;
  unsigned b0 = blockIdx.x + seed, b1 = b0 + 1, b2 = b0 ^ 9, b3 = b0 * 3;
  unsigned c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  const unsigned one = 0x00010001u;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      unsigned x0, x1, x2, x3;
      asm volatile(
          "v_xor_b32 %4, %8, %12\n\t"
          "v_xor_b32 %5, %9, %13\n\t"
          "v_xor_b32 %6, %10, %14\n\t"
          "v_xor_b32 %7, %11, %15\n\t"
          "v_pk_min_u16 %4, %4, %16\n\t"
          "v_pk_min_u16 %5, %5, %16\n\t"
          "v_pk_min_u16 %6, %6, %16\n\t"
          "v_pk_min_u16 %7, %7, %16\n\t"
          "v_pk_add_u16 %0, %0, %4\n\t"
          "v_pk_add_u16 %1, %1, %5\n\t"
          "v_pk_add_u16 %2, %2, %6\n\t"
          "v_pk_add_u16 %3, %3, %7\n\t"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
          : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(one));
    }
    a0 += c0; b1 += c1;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
}

// variant 3: lanes over hash functions: v_cmp -> s_bcnt1_i32_b64 -> s_add_u32   (1 VALU + 2 SALU / 64 compares)
__global__ void k_ballot(unsigned *out, int iters, unsigned seed) {
  unsigned a0 = threadIdx.x * 7 + seed, a1 = a0 ^ 0x55, a2 = a0 + 3, a3 = a0 * 5;
  unsigned b0 = blockIdx.x + seed, b1 = b0 + 1, b2 = b0 ^ 9, b3 = b0 * 3;
  unsigned s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      u64 m0, m1, m2, m3; unsigned p0, p1, p2, p3;
      asm volatile(
          "v_cmp_eq_u32_e64 %4, %12, %16\n\t"
          "v_cmp_eq_u32_e64 %5, %13, %17\n\t"
          "v_cmp_eq_u32_e64 %6, %14, %18\n\t"
          "v_cmp_eq_u32_e64 %7, %15, %19\n\t"
          "s_bcnt1_i32_b64 %8, %4\n\t"
          "s_bcnt1_i32_b64 %9, %5\n\t"
          "s_bcnt1_i32_b64 %10, %6\n\t"
          "s_bcnt1_i32_b64 %11, %7\n\t"
          "s_add_u32 %0, %0, %8\n\t"
          "s_add_u32 %1, %1, %9\n\t"
          "s_add_u32 %2, %2, %10\n\t"
          "s_add_u32 %3, %3, %11\n\t"
          : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3),
            "=&s"(p0), "=&s"(p1), "=&s"(p2), "=&s"(p3)
          : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3)
          : "scc");
    }
    a0 += s0; b1 += s1;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s0 + s1 + s2 + s3;
}

// variant 4: one operand from an SGPR (wave-uniform row), the other from a VGPR
__global__ void k_cmp_sgpr_addc(unsigned *out, int iters, unsigned seed) {
  unsigned a0 = threadIdx.x * 7 + seed, a1 = a0 ^ 0x55, a2 = a0 + 3, a3 = a0 * 5;
  unsigned b0 = __builtin_amdgcn_readfirstlane(blockIdx.x + seed), b1 = b0 + 1, b2 = b0 ^ 9, b3 = b0 * 3;
  unsigned c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      u64 m0, m1, m2, m3, d0, d1, d2, d3;
      asm volatile(
          "v_cmp_eq_u32_e64 %4, %16, %12\n\t"
          "v_cmp_eq_u32_e64 %5, %17, %13\n\t"
          "v_cmp_eq_u32_e64 %6, %18, %14\n\t"
          "v_cmp_eq_u32_e64 %7, %19, %15\n\t"
          "v_addc_co_u32_e64 %0, %8, 0, %0, %4\n\t"
          "v_addc_co_u32_e64 %1, %9, 0, %1, %5\n\t"
          "v_addc_co_u32_e64 %2, %10, 0, %2, %6\n\t"
          "v_addc_co_u32_e64 %3, %11, 0, %3, %7\n\t"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3),
            "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3)
          : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "s"(b0), "s"(b1), "s"(b2), "s"(b3));
    }
    a0 += c0; a1 += c1;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
}

// variant 5: plain v_add_u32 chains, the generic int VALU rate
__global__ void k_add(unsigned *out, int iters, unsigned seed) {
  unsigned c0 = threadIdx.x, c1 = seed, c2 = 3, c3 = 4, c4 = 5, c5 = 6, c6 = 7, c7 = 8;
  unsigned b0 = blockIdx.x + seed;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      asm volatile(
          "v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\t"
          "v_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\t"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(b0));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

template <typename K>
double run(K kern, int blocks, int threads, int iters, unsigned *out) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters / 8, 1u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters, 1u);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e-3;
}

int main() {
  unsigned *out; CHECK(hipMalloc(&out, 256 * 8 * 1024 * sizeof(unsigned)));
  const int iters = 4000;
  printf("waves/SIMD | cmp+addc Gcmp/s | cmp,cmp,cnd,addc | pk16 (2 ids/dword) | ballot+bcnt | sgpr-operand | v_add Gop/s (lane-ops)\n");
  for (int wps : {1, 2, 3, 4, 6, 8}) {
    // 256-thread blocks = 1 wave per SIMD per block; wps blocks per CU
    int blocks = 256 * wps;
    double t0 = run(k_cmp_addc, blocks, 256, iters, out);
    double t1 = run(k_cmp_cnd_addc, blocks, 256, iters, out);
    double t2 = run(k_pk16, blocks, 256, iters, out);
    double t3 = run(k_ballot, blocks, 256, iters, out);
    double t4 = run(k_cmp_sgpr_addc, blocks, 256, iters, out);
    double t5 = run(k_add, blocks, 256, iters, out);
    double lanes = (double)blocks * 256;
    double cmp_per_thread = (double)iters * 16 * 4;  // compares per thread in variants 0,1,3,4
    printf("%10d | %14.1f | %16.1f | %18.1f | %11.1f | %12.1f | %10.1f\n", wps,
           lanes * cmp_per_thread / t0 / 1e9, lanes * cmp_per_thread / t1 / 1e9,
           lanes * (double)iters * 16 * 8 / t2 / 1e9,  // 4 dwords x 2 ids
           lanes * cmp_per_thread / t3 / 1e9, lanes * cmp_per_thread / t4 / 1e9,
           lanes * (double)iters * 16 * 8 / t5 / 1e9);
  }
  return 0;
}
