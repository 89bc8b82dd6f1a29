// graph_kernels.hip -- the step right after the hot path in the reference's caller:
//
//   threshold <- quantile(pep.sim[upper.tri(pep.sim)], thresh_p)        R/clusterbreak.R:219
//   pep.sim[pep.sim < threshold] <- 0                                   R/clusterbreak.R:221
//   graph_from_adjacency_matrix(pep.sim, mode = "upper", weighted = TRUE)   R/clusterbreak.R:122-124
//
// MinHash similarities take only n_hash+1 distinct values (count / n_hash), so a histogram of the
// match counts of the strict upper triangle gives R's type-7 quantile EXACTLY, and only the
// surviving entries need to leave the GPU -- as an edge list instead of an 80 GB dense matrix
// (SURVEY 8(f)-1).  Both kernels stream the compact uint16 count matrix K2 produces (HBM-bound).
#include "da_common.hpp"

#include <algorithm>

#include <hipcub/hipcub.hpp>

namespace da {
namespace {

constexpr int G_TILE = 128;
constexpr int G_THREADS = 256;
constexpr int G_LDS_BINS = 8192;   // 32 KiB: MH counts (n_hash + 1) and NW codes of peptides up to 31 residues stay in LDS

// upper-triangular 128x128 tile id -> (ti, tj), row-major over the triangle
__device__ __forceinline__ void tri_tile(int64_t L, int T, int &ti, int &tj) {
  const double Td = (double)T;
  int64_t t = (int64_t)(Td + 0.5 - sqrt((Td + 0.5) * (Td + 0.5) - 2.0 * (double)L));
  if (t < 0) t = 0;
  if (t > T - 1) t = T - 1;
  auto start = [&](int64_t r) { return r * T - r * (r - 1) / 2; };
  while (t > 0 && start(t) > L) --t;
  while (t + 1 <= T - 1 && start(t + 1) <= L) ++t;
  ti = (int)t;
  tj = (int)(t + (L - start(t)));
}

// Where the count matrix lives: a dense n x n uint16 matrix (world == 0), or one rank's folded
// shard block (ShardGeom; rank p holds tile rows q*world + p right of the diagonal).
// rowmap != NULL (dense layout only): row i of the matrix is row rowmap[i] of `m` -- the column-gathered table of the UNIQUE strings
// (k_gather_columns: m[r][j] = count(unique r, sequence j)), i.e. the n x n count matrix without materialising its duplicate rows.
struct Layout { ShardGeom g; int rank; int sharded; const int32_t *rowmap; };

// block id -> upper tile (ti, tj) and the offsets that turn global (i, j) into an element index
__device__ __forceinline__ bool locate_tile(const Layout &lay, int64_t L, int T, int &ti, int &tj, int64_t &roff, int64_t &coff) {
  roff = 0; coff = 0;
  if (!lay.sharded) { tri_tile(L, T, ti, tj); return true; }
  const int q = (int)(L / T);
  ti = q * lay.g.world + lay.rank;
  tj = (int)(L % T);
  if (ti >= T || tj < ti) return false;
  const bool front = q <= lay.g.Q - 1 - q;
  roff = (int64_t)(front ? q : lay.g.Q - 1 - q) * G_TILE - (int64_t)ti * G_TILE;   // local row = i + roff
  coff = front ? -(int64_t)ti * G_TILE : shard_back(lay.g.W, lay.g.n);                           // local col = j + coff
  return true;
}

// One 128 x 128 tile per workgroup: thread t reads, in 8 passes, 8 consecutive counts (16 bytes) of row
// 16*pass + t/16 at columns 8*(t%16) -- a wave-wide load covers 4 rows x 256 B.  Falls back to 2-byte loads
// when the rows are not 16-byte aligned (odd n / ld) or the tile crosses the matrix edge.
struct TileRows { uint32_t w[8][4]; };   // [pass][4 dwords = 8 counts]; 0xFFFF marks "not an element"

__device__ __forceinline__ void load_tile(const uint16_t *__restrict__ m, int64_t ld, int64_t n, int64_t I0, int64_t J0,
                                          int64_t roff, int64_t coff, bool diag_tile, bool keep_diagonal, TileRows &t,
                                          const int32_t *__restrict__ rowmap) {
  auto row_of = [&](int64_t i) -> int64_t { return rowmap ? (int64_t)rowmap[i] : i + roff; };
  const int tr = threadIdx.x >> 4, tc = (threadIdx.x & 15) * 8;
  const int64_t j0 = J0 + tc;
  const bool inside = I0 + G_TILE <= n && J0 + G_TILE <= n;
  const bool aligned = ((ld & 7) == 0) && (((J0 + coff) & 7) == 0) && ((reinterpret_cast<uintptr_t>(m) & 15) == 0);
  if (inside && aligned && !diag_tile) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const uint4 v = *reinterpret_cast<const uint4 *>(m + row_of(I0 + 16 * p + tr) * ld + j0 + coff);
      t.w[p][0] = v.x; t.w[p][1] = v.y; t.w[p][2] = v.z; t.w[p][3] = v.w;
    }
    return;
  }
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int64_t i = I0 + 16 * p + tr;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      uint32_t lo = 0xFFFFu, hi = 0xFFFFu;
      const int64_t j = j0 + e;
      if (i < n && j < n && (j > i || (keep_diagonal && j == i))) lo = m[row_of(i) * ld + j + coff];
      if (i < n && j + 1 < n && (j + 1 > i || (keep_diagonal && j + 1 == i))) hi = m[row_of(i) * ld + j + 1 + coff];
      t.w[p][e >> 1] = lo | (hi << 16);
    }
  }
}
__device__ __forceinline__ uint32_t tile_elem(const TileRows &t, int p, int e) {
  return (t.w[p][e >> 1] >> ((e & 1) * 16)) & 0xFFFFu;
}

// hist[v] += number of pairs i < j with m[i][j] == v
__global__ __launch_bounds__(G_THREADS) void k_upper_histogram(const uint16_t *__restrict__ m, int64_t ld, int64_t n,
                                                               int nbins, unsigned long long *__restrict__ hist, int T,
                                                               Layout lay, int64_t ntiles) {
  __shared__ unsigned int lh[G_LDS_BINS];
  const bool use_lds = nbins <= G_LDS_BINS;
  if (use_lds)
    for (int b = threadIdx.x; b < nbins; b += G_THREADS) lh[b] = 0;
  __syncthreads();
  // The overwhelmingly common value is 0 (unrelated peptides share no k-mer): counted in a register,
  // not with 64 lanes hammering one LDS word (same-address atomics serialise).
  unsigned long long zeros = 0;
  // persistent workgroups: the LDS histogram is cleared and flushed once per workgroup, not once per tile
  for (int64_t L = blockIdx.x; L < ntiles; L += gridDim.x) {
    int ti, tj;
    int64_t roff, coff;
    if (!locate_tile(lay, L, T, ti, tj, roff, coff)) continue;             // block-uniform
    TileRows t;
    load_tile(m, ld, n, (int64_t)ti * G_TILE, (int64_t)tj * G_TILE, roff, coff, ti == tj, false, t, lay.rowmap);
    unsigned z = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned v = tile_elem(t, p, e);
        if (v == 0) ++z;
        else if (v < (unsigned)nbins) {             // 0xFFFF (not an element) never is: nbins <= 65535
          if (use_lds) atomicAdd(&lh[v], 1u);
          else atomicAdd(&hist[v], 1ull);
        }
      }
    zeros += z;
  }
  for (int o = 32; o > 0; o >>= 1) zeros += __shfl_down(zeros, o);          // wave sum
  if ((threadIdx.x & 63) == 0 && zeros) atomicAdd(&hist[0], zeros);
  __syncthreads();
  if (use_lds)
    for (int b = threadIdx.x; b < nbins; b += G_THREADS)
      if (lh[b]) atomicAdd(&hist[b], (unsigned long long)lh[b]);
}

// append (i, j, v) for every i <= j (diagonal optional) whose value v is flagged in keep[].
// One global atomic per 128x128 tile (a single counter word saturates near 90 atomics/us, so
// per-wave reservations -- 78 M of them at N = 100k -- would serialise the whole kernel):
// every thread flags its 64 elements, the workgroup scans the per-thread counts in LDS, one
// lane reserves the tile's run of slots, threads then write their own contiguous sub-runs.
__global__ __launch_bounds__(G_THREADS) void k_extract_edges(const uint16_t *__restrict__ m, int64_t ld, int64_t n,
                                                             const uint8_t *__restrict__ keep, int nbins,
                                                             int include_diagonal, int32_t *__restrict__ ei,
                                                             int32_t *__restrict__ ej, uint16_t *__restrict__ ev,
                                                             long long capacity, unsigned long long *__restrict__ count,
                                                             int T, Layout lay) {
  __shared__ unsigned int scan[G_THREADS];
  __shared__ unsigned long long tile_base;
  int ti, tj;
  int64_t roff, coff;
  if (!locate_tile(lay, blockIdx.x, T, ti, tj, roff, coff)) return;       // block-uniform
  const int64_t I0 = (int64_t)ti * G_TILE, J0 = (int64_t)tj * G_TILE;
  TileRows t;
  load_tile(m, ld, n, I0, J0, roff, coff, ti == tj, include_diagonal != 0, t, lay.rowmap);
  unsigned long long kept = 0;                               // bit 8*pass + e
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned v = tile_elem(t, p, e);
      if (v < (unsigned)nbins && keep[v] != 0) kept |= 1ull << (8 * p + e);
    }
  const unsigned mine = (unsigned)__popcll(kept);
  scan[threadIdx.x] = mine;
  __syncthreads();
  for (int off = 1; off < G_THREADS; off <<= 1) {            // inclusive Hillis-Steele scan (256 entries)
    const unsigned add = threadIdx.x >= (unsigned)off ? scan[threadIdx.x - off] : 0u;
    __syncthreads();
    scan[threadIdx.x] += add;
    __syncthreads();
  }
  const unsigned total = scan[G_THREADS - 1];
  if (total == 0) return;
  if (threadIdx.x == 0) tile_base = atomicAdd(count, (unsigned long long)total);
  __syncthreads();
  unsigned long long slot = tile_base + (scan[threadIdx.x] - mine);
  const int tr = threadIdx.x >> 4, tc = (threadIdx.x & 15) * 8;
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (kept & (1ull << (8 * p + e))) {
        if ((long long)slot < capacity) {
          ei[slot] = (int32_t)(I0 + 16 * p + tr);
          ej[slot] = (int32_t)(J0 + tc + e);
          ev[slot] = (uint16_t)tile_elem(t, p, e);
        }
        ++slot;
      }
    }
}

// ---- edge list -> symmetric CSR on the device (the graph the caller's Louvain step wants; da_louvain_csr) -------------------
// (i, j, code) with i <= j  ->  both directions of every off-diagonal edge as 64-bit keys row << 32 | col (diagonal entries go to
// loops[] and get the sentinel row n), one radix sort of the pairs, row pointers by binary search, columns = low key halves.
__global__ __launch_bounds__(256) void k_edges_to_keys(const int32_t *__restrict__ ei, const int32_t *__restrict__ ej,
                                                       const uint16_t *__restrict__ ev, int64_t m, int64_t n, uint64_t *__restrict__ keys,
                                                       uint16_t *__restrict__ vals, uint16_t *__restrict__ loops) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= m) return;
  const uint32_t i = (uint32_t)ei[e], j = (uint32_t)ej[e];
  const uint16_t v = ev[e];
  if (i == j) {
    loops[i] = v;
    keys[2 * e] = keys[2 * e + 1] = (uint64_t)n << 32;
    vals[2 * e] = vals[2 * e + 1] = 0;
  } else {
    keys[2 * e] = ((uint64_t)i << 32) | j;
    keys[2 * e + 1] = ((uint64_t)j << 32) | i;
    vals[2 * e] = vals[2 * e + 1] = v;
  }
}
__global__ __launch_bounds__(256) void k_csr_ptr(const uint64_t *__restrict__ keys, int64_t count, int64_t n, int64_t *__restrict__ ptr) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v > n) return;
  const uint64_t want = (uint64_t)v << 32;                    // first key of row v (v = n: the sentinel = number of real entries)
  int64_t lo = 0, hi = count;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < want) lo = mid + 1; else hi = mid;
  }
  ptr[v] = lo;
}
__global__ __launch_bounds__(256) void k_csr_cols(const uint64_t *__restrict__ keys, const int64_t *__restrict__ ptr, int64_t n,
                                                  int32_t *__restrict__ adj) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k < ptr[n]) adj[k] = (int32_t)(uint32_t)keys[k];
}

}  // namespace

static int key_bits(int64_t n) {
  int b = 0;
  for (int64_t t = n; t > 0; t >>= 1) ++b;
  return 32 + b;
}
// workspace: keys in / out (2 x 2m x 8 B), values in (2m x 2 B; the sorted values go straight to d_codes) + the sort's own scratch
size_t edges_to_csr_workspace_bytes(int64_t m, int64_t n) {
  if (m <= 0) return 256;
  size_t temp = 0;
  const int64_t N = 2 * m;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, temp, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint16_t *)nullptr,
                                           (uint16_t *)nullptr, N, 0, key_bits(n), nullptr);
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  return up((size_t)N * 8) * 2 + up((size_t)N * 2) + up(temp) + 256;
}
int launch_edges_to_csr(const int32_t *d_i, const int32_t *d_j, const uint16_t *d_v, int64_t m, int64_t n, void *d_work, size_t work_bytes,
                        int64_t *d_ptr, int32_t *d_adj, uint16_t *d_codes, uint16_t *d_loops, hipStream_t stream) {
  if (n <= 0) return DA_OK;
  if (n > 0x7ffffff0LL || m > (int64_t)1 << 40) return fail(DA_ERR_UNSUPPORTED, "edge list too large");
  DA_HIP_TRY(hipMemsetAsync(d_loops, 0xff, (size_t)n * 2, stream));
  if (m <= 0) { DA_HIP_TRY(hipMemsetAsync(d_ptr, 0, (size_t)(n + 1) * 8, stream)); return DA_OK; }
  if (work_bytes < edges_to_csr_workspace_bytes(m, n)) return fail(DA_ERR_BAD_ARG, "edges -> CSR: workspace too small");
  const int64_t N = 2 * m;
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  char *w = static_cast<char *>(d_work);
  uint64_t *keys_in = reinterpret_cast<uint64_t *>(w); w += up((size_t)N * 8);
  uint64_t *keys_out = reinterpret_cast<uint64_t *>(w); w += up((size_t)N * 8);
  uint16_t *vals_in = reinterpret_cast<uint16_t *>(w); w += up((size_t)N * 2);
  size_t temp = work_bytes - (size_t)(w - static_cast<char *>(d_work));
  hipLaunchKernelGGL(k_edges_to_keys, dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, stream, d_i, d_j, d_v, m, n, keys_in, vals_in, d_loops);
  DA_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(w, temp, keys_in, keys_out, vals_in, d_codes, N, 0, key_bits(n), stream));
  hipLaunchKernelGGL(k_csr_ptr, dim3((unsigned)ceil_div(n + 1, 256)), dim3(256), 0, stream, keys_out, N, n, d_ptr);
  hipLaunchKernelGGL(k_csr_cols, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, stream, keys_out, d_ptr, n, d_adj);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

static Layout make_layout(int64_t n, int rank, int world, const int32_t *rowmap = nullptr) {
  Layout lay;
  lay.g = shard_geom(n, world > 0 ? world : 1, G_TILE);
  lay.rank = rank;
  lay.sharded = world > 0 ? 1 : 0;
  lay.rowmap = world > 0 ? nullptr : rowmap;
  return lay;
}

// world == 0: d_m is a dense n x n matrix (ld >= n); world >= 1: d_m is rank's folded shard block (ld >= W)
int launch_upper_histogram(const uint16_t *d_m, int64_t ld, int64_t n, int nbins, unsigned long long *d_hist,
                           hipStream_t stream, int rank, int world, const int32_t *d_rowmap) {
  if (n <= 1) return DA_OK;
  if (nbins > 65535) return fail(DA_ERR_UNSUPPORTED, "histogram of uint16 counts: at most 65535 bins (value 65535 is reserved)");
  const int T = (int)ceil_div(n, G_TILE);
  const Layout lay = make_layout(n, rank, world, d_rowmap);
  const int64_t tiles = world > 0 ? (int64_t)lay.g.Q * T : (int64_t)T * (T + 1) / 2;
  if (tiles > 0x7fffffffLL) return fail(DA_ERR_UNSUPPORTED, "matrix too large for one launch");
  const unsigned grid = (unsigned)std::min<int64_t>(tiles, 256 * 16);
  hipLaunchKernelGGL(k_upper_histogram, dim3(grid), dim3(G_THREADS), 0, stream, d_m, ld, n, nbins, d_hist, T, lay, tiles);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

int launch_extract_edges(const uint16_t *d_m, int64_t ld, int64_t n, const uint8_t *d_keep, int nbins,
                         bool include_diagonal, int32_t *d_i, int32_t *d_j, uint16_t *d_v, int64_t capacity,
                         unsigned long long *d_count, hipStream_t stream, int rank, int world, const int32_t *d_rowmap) {
  if (n <= 0) return DA_OK;
  if (nbins > 65535) return fail(DA_ERR_UNSUPPORTED, "edge extraction from uint16 counts: at most 65535 bins (value 65535 is reserved)");
  const int T = (int)ceil_div(n, G_TILE);
  const Layout lay = make_layout(n, rank, world, d_rowmap);
  const int64_t tiles = world > 0 ? (int64_t)lay.g.Q * T : (int64_t)T * (T + 1) / 2;
  if (tiles > 0x7fffffffLL) return fail(DA_ERR_UNSUPPORTED, "matrix too large for one launch");
  hipLaunchKernelGGL(k_extract_edges, dim3((unsigned)tiles), dim3(G_THREADS), 0, stream, d_m, ld, n, d_keep, nbins,
                     include_diagonal ? 1 : 0, d_i, d_j, d_v, (long long)capacity, d_count, T, lay);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

}  // namespace da
