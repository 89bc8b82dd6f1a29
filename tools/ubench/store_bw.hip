// store_bw.hip -- how fast can an MI355X WRITE a dense 80 GB float64 matrix, and does the pattern matter?  (The headline step of
// similarityMH is an index expansion that writes N x N doubles once; k_expand_stream reaches 5.5 - 6.1 TB/s, k_sp_tiles 6.7.)
// Patterns over a 100 000 x 100 000 double matrix (ld = 100 000), every element written once, 16-byte stores:
//   linear     grid-stride over the whole buffer, consecutive lanes on consecutive 16-byte units (fully coalesced, no structure)
//   rows       a 1024-thread workgroup owns whole output rows (800 KB each), 16 KiB per iteration -- k_expand_stream's pattern
//   rows4      the same, four rows at a time (a string with four copies: the same values into four rows)
//   tiles      a 256-thread workgroup writes a 128 x 128 tile (128 row pieces of 1 KiB) + its mirror image -- k_sp_tiles' pattern
//   memset     hipMemsetAsync of the buffer (the driver's fill kernel)
// each with nontemporal ("nt") and ordinary stores, and with the workgroup -> address map either interleaved or in per-XCD ranges.
//   hipcc --offload-arch=gfx950 -O3 -o store_bw store_bw.hip && ./store_bw [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ void st2(double *p, double a, double b) {
  d2 v = {a, b};
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(p));
  else *reinterpret_cast<d2 *>(p) = v;
}

template <bool NT> __global__ __launch_bounds__(256) void k_linear(double *out, int64_t units, double v) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < units; u += stride) st2<NT>(out + 2 * u, v, v + 1.0);
}

// XCD: 0 = rows dealt round-robin to the workgroups; 1 = workgroup b (XCD b & 7) takes rows from the b & 7 -th eighth of the matrix
template <bool NT, int COPIES, int XCD> __global__ __launch_bounds__(1024) void k_rows(double *out, int n, int64_t ld, double v) {
  const int groups = n / COPIES;                       // row groups of COPIES consecutive rows
  const int nb = gridDim.x;
  for (int g = blockIdx.x; g < groups; g += nb) {
    int gg = g;
    if (XCD == 1) { const int x = blockIdx.x & 7, per = groups / 8; gg = x * per + (g >> 3) % per; }
    if (XCD >= 2) {                                    // block-cyclic: row blocks of XCD rows, block k belongs to XCD k & 7
      const int x = blockIdx.x & 7, k = (g >> 3);      // k-th row group served by this XCD's workgroups
      gg = ((k / XCD) * 8 + x) * XCD + k % XCD;
      if (gg >= groups) continue;
    }
    double *r0 = out + (int64_t)gg * COPIES * ld;
    for (int j2 = threadIdx.x; j2 < (n >> 1); j2 += 1024) {
      const double a = v + j2, b = a + 1.0;
#pragma unroll
      for (int q = 0; q < COPIES; ++q) st2<NT>(r0 + q * ld + 2 * j2, a, b);
    }
  }
}

template <bool NT, int XCD> __global__ __launch_bounds__(256, 4) void k_tiles(double *out, int T, int64_t ld, double v) {
  // upper tiles incl. the diagonal, row-major ids; each writes the tile and (off the diagonal) its mirror image
  const int64_t ntiles = (int64_t)T * (T + 1) / 2;
  int64_t L = blockIdx.x;
  if (XCD) { const int64_t per = (ntiles + 7) / 8; L = (blockIdx.x & 7) * per + (blockIdx.x >> 3); }
  if (L >= ntiles) return;
  int ti = (int)((2.0 * T + 1.0 - sqrt((2.0 * T + 1.0) * (2.0 * T + 1.0) - 8.0 * (double)L)) / 2.0);
  while ((int64_t)ti * T - (int64_t)ti * (ti - 1) / 2 > L) --ti;
  while ((int64_t)(ti + 1) * T - (int64_t)(ti + 1) * ti / 2 <= L) ++ti;
  const int tj = ti + (int)(L - ((int64_t)ti * T - (int64_t)ti * (ti - 1) / 2));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int pass = 0; pass < (ti == tj ? 1 : 2); ++pass) {
    double *base = out + (int64_t)(pass ? tj : ti) * 128 * ld + (int64_t)(pass ? ti : tj) * 128;
    for (int r = wave; r < 128; r += 4) st2<NT>(base + (int64_t)r * ld + 2 * lane, v + r, v + lane);   // a wave = one 1 KiB row piece
  }
}

static float timed(void (*launch)(void), int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int i = 0; i < reps; ++i) {
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

static double *g_out; static int g_n; static int64_t g_ld;
#define L_(NAME, ...) static void NAME() { __VA_ARGS__; }
L_(l_lin_nt, hipLaunchKernelGGL(k_linear<true>, dim3(256 * 16), dim3(256), 0, 0, g_out, (int64_t)g_n * g_ld / 2, 1.0))
L_(l_lin_wb, hipLaunchKernelGGL(k_linear<false>, dim3(256 * 16), dim3(256), 0, 0, g_out, (int64_t)g_n * g_ld / 2, 1.0))
L_(l_lin_nt_big, hipLaunchKernelGGL(k_linear<true>, dim3(256 * 64), dim3(256), 0, 0, g_out, (int64_t)g_n * g_ld / 2, 1.0))
L_(l_rows_nt, hipLaunchKernelGGL((k_rows<true, 1, 0>), dim3(256), dim3(1024), 0, 0, g_out, g_n, g_ld, 1.0))
L_(l_rows_nt2, hipLaunchKernelGGL((k_rows<true, 1, 0>), dim3(512), dim3(1024), 0, 0, g_out, g_n, g_ld, 1.0))
L_(l_rows_wb, hipLaunchKernelGGL((k_rows<false, 1, 0>), dim3(256), dim3(1024), 0, 0, g_out, g_n, g_ld, 1.0))
L_(l_rows_nt_x, hipLaunchKernelGGL((k_rows<true, 1, 1>), dim3(256), dim3(1024), 0, 0, g_out, g_n, g_ld, 1.0))
L_(l_rows_c16, hipLaunchKernelGGL((k_rows<true, 1, 16>), dim3(256), dim3(1024), 0, 0, g_out, g_n, g_ld, 1.0))
L_(l_rows_c128, hipLaunchKernelGGL((k_rows<true, 1, 128>), dim3(256), dim3(1024), 0, 0, g_out, g_n, g_ld, 1.0))
L_(l_rows_c1024, hipLaunchKernelGGL((k_rows<true, 1, 1024>), dim3(256), dim3(1024), 0, 0, g_out, g_n, g_ld, 1.0))
L_(l_rows_c4096, hipLaunchKernelGGL((k_rows<true, 1, 4096>), dim3(256), dim3(1024), 0, 0, g_out, g_n, g_ld, 1.0))
L_(l_rows4_nt, hipLaunchKernelGGL((k_rows<true, 4, 0>), dim3(256), dim3(1024), 0, 0, g_out, g_n, g_ld, 1.0))
L_(l_rows4_nt_x, hipLaunchKernelGGL((k_rows<true, 4, 1>), dim3(256), dim3(1024), 0, 0, g_out, g_n, g_ld, 1.0))
static int g_T;
L_(l_tiles_nt, hipLaunchKernelGGL((k_tiles<true, 0>), dim3((unsigned)((int64_t)g_T * (g_T + 1) / 2)), dim3(256), 0, 0, g_out, g_T, g_ld, 1.0))
L_(l_tiles_nt_x, hipLaunchKernelGGL((k_tiles<true, 1>), dim3((unsigned)(((int64_t)g_T * (g_T + 1) / 2 + 7) / 8 * 8)), dim3(256), 0, 0, g_out, g_T, g_ld, 1.0))
L_(l_tiles_wb_x, hipLaunchKernelGGL((k_tiles<false, 1>), dim3((unsigned)(((int64_t)g_T * (g_T + 1) / 2 + 7) / 8 * 8)), dim3(256), 0, 0, g_out, g_T, g_ld, 1.0))
L_(l_memset, CHECK(hipMemsetAsync(g_out, 0, (size_t)g_n * g_ld * 8, 0)))

int main(int argc, char **argv) {
  g_n = argc > 1 ? atoi(argv[1]) : 100000;
  g_n = g_n / 1024 * 1024;                  // whole tiles, rows divisible by everything used here
  g_ld = g_n;
  g_T = g_n / 128;
  CHECK(hipMalloc(&g_out, (size_t)g_n * g_ld * 8));
  const double gb = (double)g_n * g_ld * 8 / 1e9;
  struct { const char *name; void (*f)(void); } list[] = {
      {"linear, nt stores, 4096 workgroups", l_lin_nt}, {"linear, nt stores, 16384 workgroups", l_lin_nt_big}, {"linear, ordinary stores", l_lin_wb},
      {"rows (1024 threads, 1 / CU), nt", l_rows_nt}, {"rows, 2 workgroups / CU, nt", l_rows_nt2}, {"rows, ordinary stores", l_rows_wb}, {"rows, nt, per-XCD row ranges", l_rows_nt_x},
      {"rows, nt, XCD-cyclic blocks of 16 rows", l_rows_c16}, {"rows, nt, XCD-cyclic blocks of 128 rows", l_rows_c128},
      {"rows, nt, XCD-cyclic blocks of 1024 rows", l_rows_c1024}, {"rows, nt, XCD-cyclic blocks of 4096 rows", l_rows_c4096},
      {"rows x 4 copies, nt", l_rows4_nt}, {"rows x 4 copies, nt, per-XCD ranges", l_rows4_nt_x},
      {"tiles + mirror, nt", l_tiles_nt}, {"tiles + mirror, nt, per-XCD id ranges", l_tiles_nt_x}, {"tiles + mirror, ordinary, per-XCD", l_tiles_wb_x},
      {"hipMemsetAsync", l_memset},
  };
  printf("writing a %d x %d float64 matrix (%.1f GB) once, best of 4\n", g_n, g_n, gb);
  for (auto &e : list) {
    const float ms = timed(e.f, 4);
    printf("%-44s %8.3f ms  %6.2f TB/s  (%.3f of 8 TB/s)\n", e.name, ms, gb / ms, gb / ms / 8.0);
  }
  return 0;
}
