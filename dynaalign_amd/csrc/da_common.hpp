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
// tile rows q and Q-1-q share one stored tile row of width W = n + world*tile (rounded up to
// 8): the first left-aligned from its diagonal tile, the second right-aligned.  That halves
// the bytes the all-gather has to move.
struct ShardGeom {
  int64_t n, W, rows;  // rows = local rows per rank
  int world, tile, T, Q, Qh;
};
inline ShardGeom shard_geom(int64_t n, int world, int tile) {
  ShardGeom g;
  g.n = n; g.world = world; g.tile = tile;
  g.T = (int)ceil_div(n, tile);
  g.Q = (int)ceil_div(g.T, world);
  g.Qh = (g.Q + 1) / 2;
  g.W = ceil_div(n + (int64_t)world * tile, 8) * 8;
  g.rows = (int64_t)g.Qh * tile;
  return g;
}

// Kernel launchers implemented in the .hip translation units.  All are
// asynchronous on `stream`; argument checking is done by the C-ABI layer.
int launch_minhash_signatures(const uint8_t *d_res, const int64_t *d_off, int64_t n,
                              int k, int n_hash, const uint32_t *d_seeds, uint32_t *d_sig,
                              int64_t ld_sig, uint32_t *d_planes, int64_t ld_planes,
                              hipStream_t stream);
int launch_mh_compare(const uint32_t *d_planes, int64_t ld_planes, int64_t n, int n_hash,
                      int64_t row_begin, int64_t row_end, bool symmetric, int kind,
                      void *d_out, int64_t ld, hipStream_t stream, int plane_bits = 32, int tile_stride = 1,
                      bool upper_only = false, int fold_q = 0, int64_t fold_w = 0);
// dict_kernels.hip: signatures -> compare operand.  Dictionary codes (16 planes per group) are exact
// for n <= DA_DICT_MAX_N; *d_status_out points into the workspace (0 = ok) and is valid once the
// stream has drained.
constexpr int64_t DA_DICT_MAX_N = 131068;
size_t mh_planes_workspace_bytes(int64_t n, int n_hash);
int launch_mh_dictionary_planes(const uint32_t *d_sig, int64_t ld_sig, int64_t n, int n_hash, void *d_work,
                                uint32_t *d_planes, int64_t ld_planes, int **d_status_out, hipStream_t stream);
int launch_sig_to_planes(const uint32_t *d_sig, int64_t ld_sig, int64_t n, int n_hash, uint32_t *d_planes,
                         int64_t ld_planes, hipStream_t stream);
int launch_finalize_sharded(const uint16_t *d_g, int64_t ld_g, const ShardGeom &geom, bool is_nw, int n_hash,
                            double *d_out, int64_t ld, hipStream_t stream);
int launch_nw_encode(const uint8_t *d_res, int64_t total, uint8_t *d_codes, int32_t *d_bad,
                     hipStream_t stream);
int launch_nw(const uint8_t *d_codes, const int64_t *d_off, int64_t n, int64_t max_len,
              int matrix_id, int gap_open, int gap_ext, int64_t row_begin, int64_t row_end,
              bool symmetric, int kind, void *d_out, int64_t ld, int32_t *d_score,
              int64_t ld_score, hipStream_t stream, int shard_rank = 0, int shard_world = 0);
int launch_upper_histogram(const uint16_t *d_m, int64_t ld, int64_t n, int nbins, unsigned long long *d_hist,
                           hipStream_t stream, int rank = 0, int world = 0);
int launch_extract_edges(const uint16_t *d_m, int64_t ld, int64_t n, const uint8_t *d_keep, int nbins,
                         bool include_diagonal, int32_t *d_i, int32_t *d_j, uint16_t *d_v, int64_t capacity,
                         unsigned long long *d_count, hipStream_t stream, int rank = 0, int world = 0);
int launch_symmetrize(void *d_mat, int64_t n, int64_t ld, int kind, hipStream_t stream);
int launch_widen(const uint16_t *d_in, double *d_out, int64_t count, bool is_nw, int n_hash,
                 hipStream_t stream);

}  // namespace da
