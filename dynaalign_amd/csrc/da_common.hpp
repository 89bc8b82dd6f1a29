// da_common.hpp -- shared host-side plumbing for libdynaalign_hip.so
// (error channel, HIP call checking, launch-geometry helpers).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/dynaalign.h"

namespace da {
// ---- run-time switches --------------------------------------------------------------------------------------------------------
// Every DYNAALIGN_* environment variable the library honours, parsed ONCE (first use) into this struct; INTEGRATION.md lists them.
// They select between routes / kernels that produce identical bits (tests and A / B timings use them) or size host-side resources;
// none changes a result.  da_config_reload() (a test hook, also called by the Python mirror when the process environment changed)
// parses the environment again; library code never calls getenv.
struct Config {
  // similarityMH routes (api.cpp mh_full_symmetric)
  bool mh_no_dedup = false, mh_no_sparse = false, mh_no_pipe = false, mh_pipe_one_stream = false;
  int64_t mh_dedup_min_n = -1, mh_dedup_max_pct = -1;      // -1: the built-in rule
  uint64_t mh_sparse_max_pairs = 400000000ull;
  bool mh_no_hybrid = false, mh_hybrid_dedup = false;                          // heavy / rare split of the column dictionaries (8 dense planes + incidence lists)
  int64_t mh_hybrid_min_n = -1;                            // -1: the built-in rule
  int mh_expand = 0;                                       // 0 default (first form whose shape test passes), 1 rows, 2 rowspipe, 3 pipe, 4 tiles
  int mh_pipe_step = 0, mh_pipe_wg = 0, mh_pipe_head = 0;  // 0: the built-in schedule
  int plane_bits = 0;                                      // lower bound on the code planes: 0 / 12 / 14 / 15 / 16, 32 = raw signature bits
  // compare kernels
  bool k2_no_asm = false, k2_persist = false;
  int k2_wg_per_cu = 0;
  // similarityNW
  bool nw_no_dedup = false, nw_int32 = false, nw_no_prefix = false;
  int64_t nw_dedup_min_n = -1;
  // host-pointer boundary
  bool no_host_widen = false, plain_d2h = false, no_buffer_cache = false;
  int d2h_threads = 0, buffer_cache_pct = 30;
  uint64_t block_bytes = 0;                                // 0: half of the free device memory
  // diagnostics / host clustering
  bool trace = false, louvain_debug = false;
  int louvain_threads = 0;
};
const Config &config();


// thread-local message behind da_last_error()
std::string &last_error_ref();
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define DA_HIP_TRY(expr)                                                              \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess)                                                             \
      return ::da::fail(_e == hipErrorNoDevice || _e == hipErrorInvalidDevice         \
                            ? DA_ERR_NO_DEVICE                                        \
                            : DA_ERR_HIP,                                             \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),        \
                        __FILE__, __LINE__);                                          \
  } while (0)

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Folded layout of one rank's shard (row-sharded multi-GPU path).  Rank p owns tile rows
// t = q*world + p (q = 0..Q-1) and only their part right of the diagonal is valid, so local
// tile rows q and Q-1-q share one stored tile row of width W = ceil8(n) + world*tile: the first
// left-aligned from its diagonal tile, the second starting at column shard_back(W, n) = W - ceil8(n)
// (so both start on a multiple of 8 columns: 16-byte loads of row pieces stay aligned).  That
// halves the bytes the all-gather has to move.
struct ShardGeom {
  int64_t n, W, rows;  // rows = local rows per rank
  int world, tile, T, Q, Qh;
};
__host__ __device__ inline int64_t shard_back(int64_t W, int64_t n) { return W - ((n + 7) / 8) * 8; }   // local column of global column 0 in a back-aligned row
inline ShardGeom shard_geom(int64_t n, int world, int tile) {
  ShardGeom g;
  g.n = n; g.world = world; g.tile = tile;
  g.T = (int)ceil_div(n, tile);
  g.Q = (int)ceil_div(g.T, world);
  g.Qh = (g.Q + 1) / 2;
  g.W = ceil_div(n, 8) * 8 + (int64_t)world * tile;
  g.rows = (int64_t)g.Qh * tile;
  return g;
}

// ---- layout of the bit-plane operand of k_mh_compare (written by dict_kernels.hip) -------------
// The compare kernel stages, per 128-row tile and per stage (SP planes = SEGS 16-byte units per row),
// 128 x SEGS units into LDS with wave-wide DMA instructions of 64 consecutive units.  The operand
// is therefore stored in exactly that order -- [block of 128 rows][stage][LDS slot][unit] -- so a
// DMA instruction reads 1 KiB of consecutive memory (8 full lines) instead of 16..21 separate row
// pieces.  Slot order and the XOR swizzle of 64-byte slots are the kernel's LDS image:
//   slot of tile row r: ((r>>5)*2 + (r&1))*16 + ((r&31)>>1)   (the 16 lanes of a ds_read_b128
//   group hit consecutive slots);  unit position of logical segment s: s ^ ((slot>>2)&3) if SEGS == 4.
// Two copies: rows (planes in order) and, `copy_words` further, columns (each plane pair swapped).
__host__ __device__ inline int k2_slot(int r) { return (((r >> 5) * 2 + (r & 1)) << 4) + ((r & 31) >> 1); }
__host__ __device__ inline int k2_row_of_slot(int s) { return ((s >> 5) << 5) + ((s & 15) << 1) + ((s >> 4) & 1); }
struct PlaneGeom {
  int pl;       // planes per group of 32 hash functions: 8, 12, 16 (dictionary codes) or 32 (raw)
  int sp;       // planes per stage (pl, or 16 when pl = 32)
  int segs;     // 16-byte units per row per stage
  int nst;      // stages: groups x stages per group
  int64_t blocks, copy_words;
};
__host__ __device__ inline PlaneGeom plane_geom(int64_t n, int n_hash, int pl) {
  PlaneGeom g;
  g.pl = pl; g.sp = pl == 32 ? 16 : pl; g.segs = g.sp / 4;
  g.nst = ((n_hash + 31) / 32) * (pl / g.sp);
  g.blocks = (n + 127) / 128;
  g.copy_words = g.blocks * g.nst * 128 * g.sp;
  return g;
}
// word offset (inside one copy) of the 16-byte unit holding logical segment `seg` of `row` in stage `st`
__host__ __device__ inline int64_t plane_unit_word(const PlaneGeom &g, int64_t row, int st, int seg) {
  const int rs = k2_slot((int)(row & 127));
  const int xs = g.segs == 4 ? (rs >> 2) & 3 : 0;
  return ((((row >> 7) * g.nst + st) * 128 + rs) * g.segs + (seg ^ xs)) * 4;
}
// The hand-scheduled 16-plane kernel (k_mh_compare_a16) reads a PADDED twin of the 16-plane operand, stored behind the two
// regular copies: same [block of 128 rows][stage][LDS slot] order, but 80-byte slots (16 plane words + 4 words of padding):
// its 8-byte operand reads are then bank-conflict-free at plain immediate offsets and a wave's share of a stage is five 1 KiB
// DMA pieces.  Row copy: plane p at word p; column copy: word p ^ 1 (pair-swapped like the regular column copy).
constexpr int K2_PAD16_SLOT_WORDS = 20;
__host__ __device__ inline int64_t pad16_copy_words(int64_t n, int n_hash) {
  return ((n + 127) / 128) * (int64_t)((n_hash + 31) / 32) * 128 * K2_PAD16_SLOT_WORDS;
}
__host__ __device__ inline int64_t pad16_base_words(int64_t n, int n_hash) { return 2 * plane_geom(n, n_hash, 16).copy_words; }
__host__ __device__ inline int64_t pad16_slot_word(int64_t n, int n_hash, int64_t row, int group) {   // first word of the row's slot
  const int64_t nst = (n_hash + 31) / 32;
  return (((row >> 7) * nst + group) * 128 + k2_slot((int)(row & 127))) * K2_PAD16_SLOT_WORDS;
}
// uint32 words an operand buffer must hold for any plane count (raw 32 planes, or 16 planes + their padded twin)
inline int64_t mh_planes_words(int64_t n, int n_hash) {
  if (n <= 0 || n_hash <= 0) return 0;
  const int64_t raw = 2 * plane_geom(n, n_hash, 32).copy_words;
  const int64_t p16 = pad16_base_words(n, n_hash) + 2 * pad16_copy_words(n, n_hash);
  return raw > p16 ? raw : p16;
}

// Kernel launchers implemented in the .hip translation units.  All are
// asynchronous on `stream`; argument checking is done by the C-ABI layer.
int launch_minhash_signatures(const uint8_t *d_res, const int64_t *d_off, int64_t n,
                              int k, int n_hash, const uint32_t *d_seeds, uint32_t *d_sig,
                              int64_t ld_sig, hipStream_t stream);
int launch_mh_compare(const uint32_t *d_planes, int64_t n, int n_hash,
                      int64_t row_begin, int64_t row_end, bool symmetric, int kind,
                      void *d_out, int64_t ld, hipStream_t stream, int plane_bits = 32, int tile_stride = 1,
                      bool upper_only = false, int fold_q = 0, int64_t fold_w = 0);
// the symmetric uint16 12-plane compare in pieces (pipelined duplicate route): tile-row bands of 8 x 128 rows, taken in order
int64_t mh_sym_bands(int64_t n);
int64_t mh_sym_band_prefix(int64_t n, int64_t band);
bool mh_compare_bands_ok(int64_t n, int n_hash, int plane_bits, const void *d_out, int64_t ld);
int launch_mh_compare_bands_u16(const uint32_t *d_planes, int64_t n, int n_hash, uint16_t *d_out, int64_t ld, int64_t band_begin,
                                int64_t band_end, int wg_per_cu, hipStream_t stream, int plane_bits = 12);
int launch_mh_compare_edges_u16(const uint32_t *d_planes, int64_t n, int n_hash, uint16_t *d_out, int64_t ld, hipStream_t stream, int plane_bits = 12);
// minhash_kernels.hip, SPARSE route of the symmetric float64 compare (inputs whose signatures rarely agree): see the kernels' header comment
size_t mh_sparse_pairs_limit();
int launch_mh_sparse_count(const uint16_t *d_idsT, int64_t ld_ids, int64_t n, int n_hash, int max_ids, unsigned long long *d_stats, hipStream_t stream);
size_t mh_sparse_scratch_words(int64_t n, int n_hash, int max_ids, int64_t ld_ids);
int launch_mh_sparse(const uint16_t *d_idsT, int64_t ld_ids, int64_t n, int n_hash, int max_ids, uint64_t pairs, uint32_t *d_scratch,
                     uint32_t *d_entries32, uint16_t *d_entries, double *d_out, int64_t ld, hipStream_t stream, hipEvent_t after_buckets = nullptr);
// ... its list phase alone (k_sp_classes .. k_sp_band: the matching incidences of the codes in d_idsT, bucketed per 128 x 128 tile on or above the
// diagonal), and the kernel that ADDS those incidences to a finished dense result (heavy / rare split: the dense compare saw the heavy values only)
int launch_mh_sparse_lists(const uint16_t *d_idsT, int64_t ld_ids, int64_t n, int n_hash, int max_ids, uint64_t pairs, uint32_t *d_scratch,
                           uint32_t *d_entries32, uint16_t *d_entries, hipStream_t stream);
int launch_mh_sparse_fixup(const uint32_t *d_scratch, const uint16_t *d_entries, int64_t n, int n_hash, int max_ids, int64_t ld_ids, int kind, void *d_out,
                           int64_t ld, int64_t tile_row_begin, int64_t tile_row_end, hipStream_t stream);
// dict_kernels.hip: where the codes of launch_mh_dictionary sit in its workspace ([n_hash][*ld_ids] uint16, 0xFFFF = value seen once)
const uint16_t *mh_dictionary_codes(const void *d_work, int64_t n, int n_hash, int64_t *ld_ids);
#ifdef DA_K2_EXPERIMENTS
size_t release_compare_scratch();   // (experiment library only: idle scratch of the role-split compare kernel)
#endif
// dict_kernels.hip: signatures -> compare operand.  Dictionary codes (8 / 12 / 16 planes per group,
// whatever the largest column dictionary needs) are exact for n <= DA_DICT_MAX_N; *d_status_out
// points at two ints in the workspace ({error, largest id count}), valid once the stream has drained.
constexpr int64_t DA_DICT_MAX_N = 131068;
size_t mh_planes_workspace_bytes(int64_t n, int n_hash);
int launch_mh_dictionary(const uint32_t *d_sig, int64_t ld_sig, int64_t n, int n_hash, void *d_work,
                         int **d_status_out, hipStream_t stream);
int mh_plane_bits_for(int max_ids);
int launch_ids_to_planes(const void *d_work, int64_t n, int n_hash, int plane_bits, uint32_t *d_planes,
                         hipStream_t stream, const uint16_t *d_codes = nullptr);
// heavy / rare split of the column dictionaries (dict_kernels.hip k_hy_split): dense codes for 8 planes + sparse codes for the incidence lists
int launch_mh_heavy_split(void *d_work, int64_t n, int n_hash, int max_ids, int keep, const uint16_t **d_dense_out, const uint16_t **d_sparse_out,
                          unsigned long long *d_stats, hipStream_t stream);
int launch_sig_to_planes(const uint32_t *d_sig, int64_t ld_sig, int64_t n, int n_hash, uint32_t *d_planes,
                         hipStream_t stream);
int launch_finalize_sharded(const uint16_t *d_g, int64_t ld_g, const ShardGeom &geom, bool is_nw, int n_hash,
                            double *d_out, int64_t ld, hipStream_t stream);   // interior tiles: k_finalize_rows (16-byte stores)
int64_t shard_packed_bytes(const ShardGeom &g, int value_bits);
int launch_shards_to_table(const void *d_g, int64_t ld_g, const ShardGeom &geom, int value_bits, uint16_t *d_table, int64_t ld,
                           hipStream_t stream);
int launch_pack_shard(const uint16_t *d_local, int64_t ld, const ShardGeom &geom, int value_bits, uint8_t *d_packed,
                      hipStream_t stream);
int launch_finalize_packed(const uint8_t *d_g, const ShardGeom &geom, int value_bits, int n_hash, double *d_out, int64_t ld,
                           hipStream_t stream);
int launch_nw_encode(const uint8_t *d_res, int64_t total, uint8_t *d_codes, int32_t *d_bad,
                     hipStream_t stream);
int launch_nw(const uint8_t *d_codes, const int64_t *d_off, int64_t n, int64_t max_len,
              int matrix_id, int gap_open, int gap_ext, int64_t row_begin, int64_t row_end,
              bool symmetric, int kind, void *d_out, int64_t ld, int32_t *d_score,
              int64_t ld_score, hipStream_t stream, int shard_rank = 0, int shard_world = 0,
              const int32_t *ord_first = nullptr, const int32_t *ord_minfirst = nullptr, const int32_t *ord_maxlast = nullptr,
              const int32_t *ord_perm = nullptr, const uint8_t *ord_lcp = nullptr);   // ord_perm / ord_lcp: launch_nw_sort_unique (prefix sharing)
// nw_kernels.hip: lexicographic order of (unique) sequences of <= 24 residues + the common prefix of sorted neighbours, for the ordered DP
size_t nw_sort_unique_workspace_bytes(int64_t n);
int launch_nw_sort_unique(const uint8_t *d_codes, const int64_t *d_off, int64_t n, void *d_work, size_t work_bytes, const int32_t **perm_out,
                          const uint8_t **lcp_out, hipStream_t stream);
// nw_kernels.hip: collapse byte-identical sequences before the N x N sweep (exact; see the kernels' header comment)
struct NwDedupPlan {
  uint32_t *table; uint32_t table_size;
  int32_t *rep, *mult, *last, *fm, *fs, *pm, *ps, *uid_of, *uidx, *ufirst, *ulast, *ulen, *minfirst, *maxlast;
  int64_t *uoff;
  uint8_t *ucodes;
  size_t bytes;
};
NwDedupPlan nw_dedup_layout(void *work, int64_t n, int64_t total);
size_t nw_dedup_workspace_bytes(int64_t n, int64_t total);
int launch_nw_dedup_count(const uint8_t *d_codes, const int64_t *d_off, int64_t n, const NwDedupPlan &p, hipStream_t stream);
int launch_nw_dedup_build(const uint8_t *d_codes, const int64_t *d_off, int64_t n, int64_t U, const NwDedupPlan &p, hipStream_t stream,
                          bool first_order = false);   // true: unique ids purely by first occurrence (default: multi-copy strings first)
// minhash_kernels.hip: out[i][j] = value(D[uidx[min(i,j)]][uidx[max(i,j)]]) for the dense symmetric n x n result
int launch_expand_unique(const uint16_t *d_D, int64_t ld_d, const int32_t *d_uidx, int64_t n, int kind, bool is_nw, int n_hash,
                         void *d_out, int64_t ld, hipStream_t stream, int nw_max_len = 0, uint16_t *d_F = nullptr,
                         const int32_t *d_ufirst = nullptr, int64_t U = 0, hipEvent_t after_gather = nullptr, hipEvent_t after_rows = nullptr,
                         int table_world = 1, int64_t table_rows_local = 0,    // table_world > 1: d_D is all-gathered row blocks of cyclic 128-row units
                         bool only_leftover = false);                          // true: the caller ran launch_gather_columns / launch_expand_rows itself
// ROW expansion (MinHash: symmetric table): every output row written once from its table row held in LDS; no gathered copy
bool expand_stream_ok(int64_t n, int64_t U, int n_hash, const void *d_out, int64_t ld);
size_t expand_stream_scratch_bytes(int64_t n, int64_t U);
bool expand_stream_packed(int n_hash);   // the LDS row is packed to 9 bits per count: two K2 rings fit beside it
int launch_expand_stream(const uint16_t *d_D, int64_t ld_d, const int32_t *d_uidx, int64_t n, int64_t U, int n_hash, double *d_out, int64_t ld,
                         void *d_scratch, hipStream_t stream, hipEvent_t after_lists = nullptr);
int launch_expand_stream_lists(const int32_t *d_uidx, int64_t n, int64_t U, void *d_scratch, hipStream_t stream);
int launch_expand_stream_rows(const uint16_t *d_D, int64_t ld_d, const int32_t *d_uidx, int64_t n, int64_t U, int n_hash, double *d_out, int64_t ld,
                              void *d_scratch, int64_t row_begin, int64_t row_end, hipStream_t stream, int launch_no);   // launch_no: 0 .. 127, distinct per launch of a call
int launch_expand_rows(const uint16_t *d_F, int64_t ld_f, const int32_t *d_uidx, int64_t n, bool is_nw, int n_hash, int nw_max_len,
                       double *d_out, int64_t ld, int64_t band_begin, int64_t band_end, hipStream_t stream);
// bytes of the column-gathered table (d_F) that switches launch_expand_unique to its two streaming passes; 0 = shape not covered
size_t expand_rows_workspace_bytes(int64_t n, int64_t U, int kind, bool is_nw, int n_hash, int nw_max_len);
int launch_upper_histogram(const uint16_t *d_m, int64_t ld, int64_t n, int nbins, unsigned long long *d_hist,
                           hipStream_t stream, int rank = 0, int world = 0, const int32_t *d_rowmap = nullptr);
int launch_extract_edges(const uint16_t *d_m, int64_t ld, int64_t n, const uint8_t *d_keep, int nbins,
                         bool include_diagonal, int32_t *d_i, int32_t *d_j, uint16_t *d_v, int64_t capacity,
                         unsigned long long *d_count, hipStream_t stream, int rank = 0, int world = 0,
                         const int32_t *d_rowmap = nullptr);
// minhash_kernels.hip: F[r][j] = D[r][uidx[j]] for the columns right of (from_first_tile: from) the 128-tile of r's first occurrence
int launch_gather_columns(const uint16_t *d_D, int64_t ld_d, const int32_t *d_uidx, const int32_t *d_ufirst, int64_t n, int64_t U,
                          uint16_t *d_F, int64_t ld_f, bool from_first_tile, hipStream_t stream, int table_world = 1,
                          int64_t table_rows_local = 0, int64_t row_begin = 0, int64_t row_end = -1);
// graph_kernels.hip: (i <= j, code) edge list -> symmetric CSR sorted by (row, column); diagonal entries -> loops[] (0xFFFF = none)
size_t edges_to_csr_workspace_bytes(int64_t m, int64_t n);
int launch_edges_to_csr(const int32_t *d_i, const int32_t *d_j, const uint16_t *d_v, int64_t m, int64_t n, void *d_work, size_t work_bytes,
                        int64_t *d_ptr, int32_t *d_adj, uint16_t *d_codes, uint16_t *d_loops, hipStream_t stream);
int launch_symmetrize(void *d_mat, int64_t n, int64_t ld, int kind, hipStream_t stream);
int launch_acc_counts(uint32_t *d_acc, const uint16_t *d_cnt, int64_t count, bool first, hipStream_t stream);
int launch_counts32_to_f64(const uint32_t *d_acc, double *d_out, int64_t count, int n_hash, hipStream_t stream);
int launch_widen(const uint16_t *d_in, double *d_out, int64_t count, bool is_nw, int n_hash,
                 hipStream_t stream);

}  // namespace da
