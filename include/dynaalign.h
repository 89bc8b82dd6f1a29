/*
 * dynaalign.h -- C ABI of libdynaalign_hip.so (MI355X / gfx950 HIP kernels).
 *
 * This is the drop-in boundary for DynaAlign's all-pairs similarity hot path.
 * The reference crosses from R into native code through two Rcpp-generated
 * .Call entry points (reference src/RcppExports.cpp:15-25 and :28-39) that
 * convert SEXPs and call
 *     NumericMatrix similarityMH(CharacterVector, int k, int n_hash)
 *                                        reference src/minHash.cpp:119-188
 *     NumericMatrix similarityNW(CharacterVector, std::string, int, int)
 *                                        reference src/pairwiseSeqAlign.cpp:331-365
 * The functions below replace the BODIES of those two functions.  The R-level
 * signatures (reference R/RcppExports.R:15-17, :34-36) and the .Call symbols
 * stay exactly as they are; the Rcpp glue that binds them to this ABI is in
 * r_glue/ and described in INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; no exceptions, no library-owned memory crosses.
 *   - sequences are passed packed: `residues` holds the raw bytes of all
 *     sequences back to back, `offsets[i]..offsets[i+1]` delimits sequence i
 *     (n+1 entries).  This is what the glue builds once from the STRSXP
 *     (the reference instead copies per element: src/minHash.cpp:147,
 *     src/pairwiseSeqAlign.cpp:341-343).
 *   - every function returns DA_OK (0) or a DA_ERR_* code; da_last_error()
 *     returns the message for the calling thread.  For the reference's own
 *     error conditions the message text is the reference's, verbatim.
 *   - N x N results are symmetric, so row- and column-major coincide: `out`
 *     may be R's REAL() pointer (reference allocates NumericMatrix(n,n) at
 *     src/minHash.cpp:134 / src/pairwiseSeqAlign.cpp:335).
 *   - "host" functions take host pointers and do their own H2D/D2H; "dev"
 *     functions take DEVICE pointers (hipMalloc'ed by the caller, e.g. a
 *     torch tensor's data_ptr()) plus a hipStream_t passed as void*, launch
 *     asynchronously on that stream and never synchronise.
 *   - there is NO CPU fallback: without a usable HIP device every compute
 *     entry point fails with DA_ERR_NO_DEVICE.
 */
#ifndef DYNAALIGN_H
#define DYNAALIGN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* (round 4 added entry points only -- da_config_reload, da_debug_comm_cache_state, da_mh_last_route_split: the version stays)
 * 2: the folded shard layout changed (da_shard_ld = ceil8(n) + world * 128, back-aligned rows start at column world * 128) and the
 *    duplicate-route / multi-device entry points were added; every round-1 entry point keeps its signature */
#define DA_ABI_VERSION 2

enum da_status {
  DA_OK = 0,
  DA_ERR_EMPTY_INPUT = 1,      /* "Input sequences vector cannot be empty"     src/minHash.cpp:121-123 */
  DA_ERR_BAD_K = 2,            /* "'k' must be a positive integer"             src/minHash.cpp:125-127 */
  DA_ERR_BAD_NHASH = 3,        /* "Number of hash functions must be positive"  src/minHash.cpp:129-131 */
  DA_ERR_BAD_MATRIX = 4,       /* "Invalid substitution matrix name: %s"       src/pairwiseSeqAlign.cpp:204 */
  DA_ERR_BAD_RESIDUE_SEQ1 = 5, /* "Invalid amino acid in sequence1: %c"        src/pairwiseSeqAlign.cpp:241-243 */
  DA_ERR_BAD_RESIDUE_SEQ2 = 6, /* "Invalid amino acid in sequence2: %c"        src/pairwiseSeqAlign.cpp:248-250 */
  DA_ERR_NOMEM = 7,
  DA_ERR_NO_DEVICE = 8,        /* no HIP device / runtime: the product has no CPU path */
  DA_ERR_HIP = 9,              /* a HIP call failed; message carries hipGetErrorString */
  DA_ERR_UNSUPPORTED = 10,     /* outside what the gfx950 kernels implement (message says what) */
  DA_ERR_BAD_ARG = 11
};

/* element type of a device-side similarity block */
enum da_out_kind {
  DA_OUT_F64 = 0,    /* double: MH matches/n_hash (src/minHash.cpp:174), NW matches/len (src/pairwiseSeqAlign.cpp:311) */
  DA_OUT_COMPACT = 1, /* uint16: MH match count; NW (matches << 8 | len) -- valid for len <= 255 */
  DA_OUT_PACK32 = 2   /* uint32, NW only: (matches << 16 | len) -- any supported length */
};

const char *da_last_error(void);
const char *da_status_message(int status); /* static text for a code ("" if none) */
int da_abi_version(void);
int da_device_count(void); /* 0 when no HIP device is usable */
/* The library parks its large device buffers (>= 1 MiB; per device at most 16 buffers totalling at most 30 % of the device's
 * memory (DYNAALIGN_BUFFER_CACHE_PCT changes the share) -- the 80 GB result buffer costs seconds to allocate, and one call of the duplicate-collapsing routes uses four
 * large buffers) for the next call instead of returning them to the driver.  Parked memory is invisible to other allocators
 * in the process (e.g. PyTorch's): this call hands everything back.  An allocation of the library's own that would
 * otherwise fail releases the parked buffers first.  Returns the bytes freed. */
size_t da_release_device_memory(void);
/* Run-time switches.  The library reads its DYNAALIGN_* environment variables (INTEGRATION.md lists them: route / kernel selection
 * between forms that produce identical bits, sizes of host-side resources, tracing) ONCE, at its first use, into one struct.
 * da_config_reload parses the environment again -- a hook for tests and A / B timing scripts that flip a switch inside one process
 * (the Python mirror calls it by itself when the process's DYNAALIGN_* environment changed since its last call). */
void da_config_reload(void);
/* Test hook for the communicator cache of DA_EXCHANGE_ALLGATHER (the communicators of a device list are kept between calls; a call in which
 * some rank failed destroys its list's entry, since a communicator that saw an abandoned collective may hang the next one):
 * out2 = {cached device lists, entries destroyed after a failed call}; force_fail != 0 makes rank 0 of the next ALLGATHER call fail
 * before the collective. */
int da_debug_comm_cache_state(int force_fail, size_t *out2);

/* ---- HashFamily (reference src/minHash.cpp:67-89) ------------------------ */

/* seeds_out[h] = h-th raw output of std::mt19937(seed)  (src/minHash.cpp:75-80). */
int da_hash_family_seeds(uint32_t seed, int n_hash, uint32_t *seeds_out);
/* what the reference's default argument does: std::random_device{}() (src/minHash.cpp:73). */
uint32_t da_random_seed(void);

/* ---- host-pointer entry points (what the Rcpp glue calls) ---------------- */

/* similarityMH body (src/minHash.cpp:119-188).  `seeds` = n_hash hash seeds
 * (da_hash_family_seeds); validation order and messages as :121-131.
 * out: n*n doubles, diagonal exactly 1.0 (:161). */
int da_similarity_mh(const uint8_t *residues, const int64_t *offsets, int64_t n,
                     int k, int n_hash, const uint32_t *seeds, double *out);

/* ---- the same two bodies on several GPUs of the node, ONE process (SURVEY 8(b) "da_opts", 8(e)) ----
 * opts == NULL behaves like da_similarity_mh / da_similarity_nw on the current device.  With a device list, a host
 * thread per device runs the pipeline on its GPU and copies ITS rows of the result into `out` (P PCIe links instead
 * of one).  `exchange` selects how the pair space is split and reassembled:
 *   DA_EXCHANGE_ROWS       device p computes the contiguous row block it copies out; no device-to-device traffic
 *                          (default; the same device may be listed more than once);
 *   DA_EXCHANGE_ALLGATHER  cyclic upper-triangle shards (da_dev_*_shard), ONE ncclAllGather over xGMI (RCCL,
 *                          ncclCommInitAll on the list), mirror + widen to the full matrix on every device;
 *   DA_EXCHANGE_PEERCOPY   the same shards, exchanged by concurrent hipMemcpyPeerAsync reads of every peer's block
 *                          (xGMI is point-to-point: all links at once; no RCCL needed).
 * NW with an exchange needs sequences of <= 64 residues (uint16 shard codes); ROWS has the limits of da_similarity_nw.
 * phase_ms (optional, DA_PHASE_COUNT doubles): per phase the maximum over devices, in ms --
 *   [0] upload + signatures/codes (+ ncclCommInitAll on the first call with a device list: the communicators are cached until
 *   da_release_device_memory())  [1] compare / NW  [2] exchange  [3] finalize  [4] device-to-host  [5] whole call, from entry.
 * The R glue builds this struct from options(DynaAlign.devices = , DynaAlign.exchange = ) (r_glue/). */
enum da_exchange { DA_EXCHANGE_ROWS = 0, DA_EXCHANGE_ALLGATHER = 1, DA_EXCHANGE_PEERCOPY = 2 };
#define DA_PHASE_COUNT 6
typedef struct da_opts {
  uint32_t struct_size;      /* sizeof(da_opts) -- lets the struct grow without breaking callers */
  int32_t n_devices;         /* 0: the current device */
  const int32_t *devices;    /* HIP device ordinals */
  int32_t exchange;          /* enum da_exchange */
  uint32_t reserved;         /* 0 */
  double *phase_ms;          /* NULL or DA_PHASE_COUNT doubles (out) */
} da_opts;
int da_similarity_mh_opts(const uint8_t *residues, const int64_t *offsets, int64_t n,
                          int k, int n_hash, const uint32_t *seeds, double *out, const da_opts *opts);
int da_similarity_nw_opts(const uint8_t *residues, const int64_t *offsets, int64_t n,
                          const char *matrix_name, int gap_open, int gap_ext, double *out, const da_opts *opts);
/* 1 when librccl can be bound in this process (DA_EXCHANGE_ALLGATHER usable), else 0. */
int da_rccl_available(void);

/* The signature matrix alone (src/minHash.cpp:140-157): sig_out[n][n_hash]. */
int da_minhash_signatures(const uint8_t *residues, const int64_t *offsets, int64_t n,
                          int k, int n_hash, const uint32_t *seeds, uint32_t *sig_out);

/* Integer match counts (the numerator at src/minHash.cpp:168-174) for rows
 * [row_begin,row_end) x all n columns; counts_out[rows][n]; diagonal = n_hash.
 * Requires n_hash <= 65535. */
int da_mh_counts(const uint8_t *residues, const int64_t *offsets, int64_t n,
                 int k, int n_hash, const uint32_t *seeds,
                 int64_t row_begin, int64_t row_end, uint16_t *counts_out);

/* similarityNW body (src/pairwiseSeqAlign.cpp:331-365): BLOSUM name lookup
 * (:190-206), pair (i,j), i<=j, evaluated as calc(seq[i], seq[j]) (:340-346),
 * mirrored (:349-350), diagonal computed.  n == 0 returns DA_OK and writes
 * nothing (the reference returns a 0x0 matrix).  Residue errors reproduce the
 * reference's first-raised message (lazy row-major validation, :238-250). */
int da_similarity_nw(const uint8_t *residues, const int64_t *offsets, int64_t n,
                     const char *matrix_name, int gap_open, int gap_ext, double *out);

/* The integers behind similarityNW for rows [row_begin,row_end) x all n
 * columns, each entry for calc(seq[min(i,j)], seq[max(i,j)]):
 *   matches_out / len_out : the two operands of the divide at :311
 *   score_out             : M[m][n] after the fill (:268-278), never exposed
 *                           by the reference -- auxiliary, may be NULL.
 * Arrays are [rows][n] int32; any may be NULL. */
int da_nw_pairs(const uint8_t *residues, const int64_t *offsets, int64_t n,
                const char *matrix_name, int gap_open, int gap_ext,
                int64_t row_begin, int64_t row_end,
                int32_t *matches_out, int32_t *len_out, int32_t *score_out);

/* ---- device-pointer entry points (bench / multi-GPU sharding) ------------ */

/* Leading dimension (in uint32 elements) the library uses for signature
 * matrices: n_hash rounded up to a multiple of 32. */
int64_t da_sig_ld(int n_hash);

/* K1: signature build.
 *   d_sig : n rows of ld_sig (>= n_hash) uint32 -- the signatures themselves
 *           (src/minHash.cpp:140-157); columns [n_hash, ld_sig) not written. */
int da_dev_minhash_signatures(const uint8_t *d_residues, const int64_t *d_offsets, int64_t n,
                              int64_t total_residues, int64_t max_len,
                              int k, int n_hash, const uint32_t *d_seeds,
                              uint32_t *d_sig, int64_t ld_sig, void *stream);

/* K1b: signatures -> operand of the compare kernel.  The compare only asks "equal or not" per
 * hash function (src/minHash.cpp:168-173), so each column of d_sig is re-coded exactly: values
 * occurring >= 2 times get dense ids, values occurring once get codes that never match (one code
 * on the row side, another on the column side; the compare kernel forces the diagonal).  The
 * codes are then bit-transposed per group of 32 hash functions with as many bit planes as the
 * largest column dictionary needs -- 8, 12 or 16 instead of the 32 of the raw values -- and the
 * bit-sliced compare does that fraction of the work.  Exact for every input with n <= 131068
 * (at most n/2 repeated values per column fit 16 bits); larger n -- or the never-observed
 * overflow of the dictionary's LDS table -- produce the raw 32-plane operand instead.
 *   min_plane_bits: 0 = as few planes as the data needs; 12 / 14 / 15 / 16 = at least that many code
 *                   bits; 32 = raw signature bits (no dictionary).  (14 / 15 = the 16-plane operand with
 *                   its top planes zero: the hand-scheduled kernel skips their step / half step.)
 *   d_planes      : 16-byte aligned buffer of planes_words >= da_mh_planes_words(n, n_hash)
 *                   uint32; opaque (blocked in the order the compare kernel stages it).
 *   d_work        : da_mh_planes_workspace_bytes(n, n_hash) bytes of scratch, 256-byte aligned
 *   plane_bits_out: 8, 12, 14, 15, 16 or 32 -- pass it to da_dev_mh_compare[_shard].
 * Synchronises `stream` once (reads the dictionary's status words) on the dictionary route.
 * The environment variable DYNAALIGN_PLANE_BITS (12, 16, 32) raises min_plane_bits (debugging aid;
 * it also reaches the host-pointer entry points). */
int64_t da_mh_planes_words(int64_t n, int n_hash);
size_t da_mh_planes_workspace_bytes(int64_t n, int n_hash);
int da_dev_mh_planes(const uint32_t *d_sig, int64_t ld_sig, int64_t n, int n_hash, int min_plane_bits,
                     void *d_work, size_t work_bytes, uint32_t *d_planes, int64_t planes_words,
                     int *plane_bits_out, void *stream);

/* K2: all-pairs signature compare (bit-sliced: OR over planes of a XOR b, then
 * popcount; matches = n_hash - mismatches).
 * Computes rows [row_begin,row_end) of the n x n result into d_out, which
 * holds (row_end-row_begin) rows of leading dimension ld (>= n) elements.
 *   symmetric != 0 : requires row_begin == 0, row_end == n; only tiles on or
 *                    above the diagonal are compared and each is stored twice
 *                    (direct + mirrored), like src/minHash.cpp:175-176.
 *   symmetric == 0 : every (i,j) of the row block is compared (row-sharding).
 * kind selects double or uint16 counts.  Diagonal = 1.0 / n_hash.
 * d_planes, plane_bits: from da_dev_mh_planes for the same n and n_hash. */
int da_dev_mh_compare(const uint32_t *d_planes, int plane_bits, int64_t n, int n_hash,
                      int64_t row_begin, int64_t row_end, int symmetric,
                      int kind, void *d_out, int64_t ld, void *stream);

/* K1 + K1b + K2 as ONE call: similarityMH (src/minHash.cpp:119-188) from packed residues in HBM to the dense float64
 * n x n matrix in HBM (d_out, leading dimension ld >= n doubles; 16-byte aligned and even ld for the wide-store kernels).
 * Byte-identical sequences have identical signatures, so -- like da_dev_nw -- the call first collapses them: signatures,
 * dictionary codes and the compare run on the U unique strings (K2's work shrinks by (U/n)^2) and the n x n matrix is an
 * index expansion of the U x U count table.  Exact; taken when U <= 0.68 n (0.6 n when only the tile expansion applies; below that K2's saving
 * outweighs the expansion), n >= 2048, U <= 65536, n_hash <= 2047; otherwise the three kernels run on all n rows (what uniform peptides
 * get).  Allocates its intermediates itself (parked between calls, da_release_device_memory frees them) and synchronises
 * the stream.  DYNAALIGN_MH_NO_DEDUP=1 disables the route.  da_mh_last_route reports what the calling thread's last such
 * call did: n, unique strings, whether the route was taken, the plane count K1b chose, and the times in ms of
 * {plan, K1 + K1b, K2, column gather, k_expand_rows, diagonal / border tiles} (direct route: {plan, K1 + K1b, K2, 0, 0, 0}).
 * A second exact route serves inputs without duplicates whose signatures rarely agree (uniform random peptides: 1.4e8 matching
 * (pair, hash function) incidences at n = 100k against 2.5e12 compared ones): the dictionary codes of K1b say which sequences
 * share a value in a column; those incidences are enumerated, bucketed per 128 x 128 output tile, added up in LDS and every tile
 * is written once.  Taken when the input has few duplicates (>= 90 % unique), the incidences number <= n_hash / 5000 per pair on average and
 * <= DYNAALIGN_MH_SPARSE_MAX_PAIRS (default 4e8) in all, no value occurs more than 4096 times in a column, <= 32768 repeated values per column, n >= 2048, n_hash <= 2047; DYNAALIGN_MH_NO_SPARSE=1 disables it.
 * Then *dedup_taken_out = 2, *unique_out = the number of incidences and the times are {plan, K1 + K1b, buckets, 0, tile pass, 0}.
 * The duplicate route's expansion has two kernel families, each with a form PIPELINED with K2 (DYNAALIGN_MH_EXPAND = rowspipe | rows | pipe |
 * tiles picks one; default: the first whose shape test passes; DYNAALIGN_MH_NO_PIPE=1 takes the pipelined forms out):
 *   rows (*dedup_taken_out = 4)   K2 on the table, then ONE pass that holds a table row in LDS and writes the output rows of its copies
 *                                 (16-byte stores: needs an even ld and a 16-byte aligned d_out); times {plan, K1 + K1b, K2, copy lists, that pass, 0}
 *   tiles (= 1)                   K2, column gather, tile expansion + diagonal / border tiles (the times listed above)
 *   rowspipe (= 5), pipe (= 3)    the same kernels with the table compared in bands of 1024 rows by a persistent kernel on a side stream (needs 12
 *                                 code planes and n_hash > 32) while the finished table rows are expanded; unique strings are numbered by first
 *                                 occurrence.  The times are then {plan, K1 + K1b, the compare's span, copy lists (rows) / sum of the gather launches
 *                                 (tiles), time some expansion launch was running, diagonal / border tiles} -- the middle three overlap;
 *                                 da_mh_last_route_chunks gives the number of chunks and of expansion launches.
 * All forms write the same bits.
 * Heavy / rare split (round 4; the direct route; the duplicate route's table compare when its dictionaries need more than 12 planes -- the banded kernel of the
 * pipelined forms exists for 12 and 8 planes -- or with DYNAALIGN_MH_HYBRID_DEDUP=1): when the column
 * dictionaries need 12 - 16 code planes but nearly all matching incidences sit on each column's 254 most frequent values (clustered data: 7e6 of
 * 2.2e10 on the h3n2-like 100k set), the bit-sliced compare runs on EIGHT planes of dense codes for those values -- every other value reads as
 * "never equal" -- and the incidences of the remaining repeated values are enumerated by the sparse route's list kernels and added to the result
 * (count' = count + m, the float64 element recomputed as (count + m) / n_hash: same division, same bits).  Exact.  Taken when n >= 16384
 * (DYNAALIGN_MH_HYBRID_MIN_N), 32 < n_hash <= 2047, <= 32768 repeated values per column, the rare incidences number <= n_hash / 25000 per pair on average;
 * DYNAALIGN_MH_NO_HYBRID=1 disables it.  Then *plane_bits_out = 8 and da_mh_last_route_split gives the number of rare incidences (-1: not taken)
 * and the plane count the dictionaries would have needed.
 * Any pointer may be NULL. */
int da_dev_similarity_mh(const uint8_t *d_residues, const int64_t *d_offsets, int64_t n, int64_t total_residues,
                         int k, int n_hash, const uint32_t *d_seeds, double *d_out, int64_t ld, void *stream);
int da_mh_last_route(int64_t *n_out, int64_t *unique_out, int *dedup_taken_out, int *plane_bits_out, double *ms6_out);
int da_mh_last_route_chunks(int *chunks_out, int *expand_launches_out);
int da_mh_last_route_split(int64_t *rare_incidences_out, int *plane_bits_without_out);

/* ---- the pieces of the duplicate-collapsing routes, for callers that orchestrate the steps themselves (the one-process-per-GPU
 * sharded drivers: every rank builds the same plan, computes ITS shard of the unique table with the *_shard / *_unique_rows calls
 * on the plan's strings, all-gathers the shards -- (U/n)^2 of the bytes -- and expands locally).
 * da_dev_unique_plan: byte-identical strings of (d_bytes, d_offsets) collapsed (exact: byte-for-byte compares).  Fills *plan with
 * device pointers INTO d_work (valid while d_work lives): unique id of every input row, first / last occurrence of every unique
 * string, the unique strings back to back + their offsets, and per 64-row block of the unique table the smallest first / largest
 * last occurrence (the ordered NW sweep skips what no original pair i < j needs).  Multi-copy strings are numbered before single-copy
 * ones, each group in input order.  Synchronises `stream` once (the unique count). */
typedef struct da_unique_plan {
  uint32_t struct_size;            /* sizeof(da_unique_plan), set by the caller */
  int32_t reserved;
  int64_t n, unique;
  const int32_t *d_uidx;           /* [n] */
  const int32_t *d_ufirst;         /* [unique] */
  const int32_t *d_ulast;          /* [unique] */
  const uint8_t *d_ubytes;         /* the unique strings */
  const int64_t *d_uoffsets;       /* [unique + 1] */
  const int32_t *d_minfirst;       /* [ceil(unique / 64)] */
  const int32_t *d_maxlast;        /* [ceil(unique / 64)] */
} da_unique_plan;
size_t da_dev_unique_plan_bytes(int64_t n, int64_t total_bytes);
int da_dev_unique_plan(const uint8_t *d_bytes, const int64_t *d_offsets, int64_t n, int64_t total_bytes, void *d_work, size_t work_bytes,
                       da_unique_plan *plan, void *stream);
/* gathered MinHash shards of an n-row problem (da_dev_mh_compare_shard blocks in rank order: value_bits = 0 and ld_g = da_shard_ld, or
 * da_dev_pack_shard blocks: value_bits = their bit count) -> the symmetric uint16 count table [n][ld_table]. */
int da_dev_shards_to_table(const void *d_gathered, int64_t ld_g, int64_t n, int world, int value_bits, uint16_t *d_table, int64_t ld_table,
                           void *stream);
/* rank `rank` of `world`'s part of the ORDERED unique table of similarityNW, calc(U_p, U_q) with U_p as sequence1
 * (src/pairwiseSeqAlign.cpp:340-346 evaluates calc(seq[i], seq[j]) for i < j and the function is not symmetric): the cyclic 128-row
 * units rank, rank + world, ... stored back to back in d_out (ceil(ceil(unique / 128) / world) * 128 rows of ld >= unique uint16
 * matches<<8|length codes; world = 1: the whole table in natural row order); entries no original pair needs stay unwritten.  The plan
 * must have been built on the ENCODED residues (da_dev_nw_encode).  Sequences of 1..64 residues, penalties >= 0. */
int da_dev_nw_unique_rows(const da_unique_plan *plan, int64_t max_len, int matrix_id, int gap_open, int gap_ext, int rank, int world,
                          uint16_t *d_out, int64_t ld, void *stream);
/* the n x n uint16 matrix WITHOUT its duplicate rows: d_rows[r][j] = table[r][u(j)] for every unique string r and every sequence j from
 * the 128-column tile of r's first occurrence on ([unique][ceil8(n)] uint16, da_dev_unique_rows_bytes; natural-row-order table of at most
 * 65536 strings).  Row i of the full matrix is row plan->d_uidx[i] of it for every j >= i: exactly what the *_rows variants of the
 * histogram / edge-extraction calls below read (the threshold + edge list of the duplicate route, with no n x n matrix at all). */
size_t da_dev_unique_rows_bytes(int64_t n, int64_t unique);
int da_dev_unique_rows(const uint16_t *d_table, int64_t ld_table, const da_unique_plan *plan, uint16_t *d_rows, void *stream);
int da_dev_upper_histogram_rows(const uint16_t *d_rows, int64_t ld, const int32_t *d_rowmap, int64_t n, int nbins, uint64_t *d_hist,
                                void *stream);
int da_dev_extract_edges_rows(const uint16_t *d_rows, int64_t ld, const int32_t *d_rowmap, int64_t n, const uint8_t *d_keep, int nbins,
                              int include_diagonal, int32_t *d_i, int32_t *d_j, uint16_t *d_v, int64_t capacity, uint64_t *d_count,
                              void *stream);
/* dense float64 n x n result from the table of the unique strings: out[i][j] = value(table[u(min(i,j))][u(max(i,j))]), value = count / n_hash
 * (is_nw = 0) or matches / length (is_nw = 1, nw_max_len = longest sequence).  table_world = 1: row r of the table is row r; > 1: the
 * table is the all-gathered row blocks of cyclic 128-row units (rank p computed units p, p + world, ...; every block holds
 * ceil(ceil(unique / 128) / world) * 128 rows).  d_work: da_dev_expand_workspace_bytes bytes (the column-gathered twin of the
 * table for the two streaming passes; with less, or NULL, the one-kernel expansion runs).  MinHash tables in one block (is_nw = 0, table_world = 1;
 * the table must be the full symmetric square) with an even ld, a 16-byte aligned d_out, unique <= 65536 and n_hash <= 2047 take the ROW expansion
 * instead (every output row written once from its table row held in LDS; d_work then only holds a few MB of copy lists; DYNAALIGN_EXPAND_NO_STREAM=1
 * keeps the tile passes). */
size_t da_dev_expand_workspace_bytes(int64_t n, int64_t unique, int is_nw, int n_hash, int nw_max_len);
int da_dev_expand_unique(const uint16_t *d_table, int64_t ld_table, int table_world, const da_unique_plan *plan, int is_nw, int n_hash,
                         int nw_max_len, void *d_work, size_t work_bytes, double *d_out, int64_t ld, void *stream);

/* K0: validate + encode residues to BLOSUM row indices 0..23
 * (src/pairwiseSeqAlign.cpp:15-21).  d_codes[total]; *d_bad (int32, caller
 * zeroes it) becomes INT32_MAX - (smallest offending byte position) if any
 * byte is not one of ARNDCQEGHILKMFPSTWYVBZX*, and stays 0 otherwise. */
int da_dev_nw_encode(const uint8_t *d_residues, int64_t total_residues,
                     uint8_t *d_codes, int32_t *d_bad, void *stream);

/* Sequence lengths: <= 64 residues run one lane per pair (k_nw_short), 65..1024 one wavefront per pair (k_nw_long),
 * 1025..32767 the same sweep in column blocks of 1024 with the block boundary spilled to HBM (k_nw_xlong; float64 / 32-bit
 * packed output; synchronises the stream); longer ones fail with DA_ERR_UNSUPPORTED (16-bit alignment length). */
/* K3: all-pairs NW identity.  Same row-block / symmetric / ld conventions as
 * da_dev_mh_compare.  d_codes from da_dev_nw_encode.  kind: double ratio or
 * uint16 (matches<<8|len).  d_score (int32, same shape, ld_score) may be NULL.
 * max_len = longest sequence (host knows it from offsets). */
int da_dev_nw(const uint8_t *d_codes, const int64_t *d_offsets, int64_t n, int64_t max_len,
              int matrix_id, int gap_open, int gap_ext,
              int64_t row_begin, int64_t row_end, int symmetric,
              int kind, void *d_out, int64_t ld, int32_t *d_score, int64_t ld_score,
              void *stream);

/* With symmetric != 0, kind F64 / COMPACT and no score output, da_dev_nw (and da_similarity_nw[_edges]) first collapse
 * byte-identical sequences: the DP runs on the table of unique strings -- as an ORDERED square, calculate_similarity is
 * not symmetric (src/pairwiseSeqAlign.cpp:340-346 evaluates calc(seq[i], seq[j]), i < j) -- and the n x n result is an index
 * expansion of it.  Exact; taken when >= 15 % of the sequences are duplicates, n >= 2048, sequences <= 64 residues,
 * penalties >= 0; synchronises the stream.  DYNAALIGN_NW_NO_DEDUP=1 disables it.  da_nw_last_route reports what the calling
 * thread's last such call did: n, unique strings, whether the route was taken, and {plan, DP kernel, expansion} times in ms
 * (direct route: {0, DP kernel, 0}).  Any pointer may be NULL. */
int da_nw_last_route(int64_t *n_out, int64_t *unique_out, int *dedup_taken_out, double *ms3_out);

/* ---- row-sharding of the pair space over the GPUs of a node ---
 * Rank p of `world` owns the tile rows p, p+world, p+2*world, ... of the pair
 * space (tile = 128 rows for both kinds; cyclic so the upper-triangular work
 * is balanced) and computes only the tiles on or right of the diagonal.  Its
 * result is a compact uint16 block of da_shard_rows() rows x da_shard_ld()
 * columns in a FOLDED layout: local tile rows q and Q-1-q share one stored tile
 * row (the first left-aligned from its diagonal tile, the second right-aligned),
 * which halves the bytes to exchange.  One all-gather of those equally sized
 * blocks (RCCL; rank order) followed by da_dev_finalize_shards on every rank
 * reassembles the full float64 matrix.  The reference has no counterpart (it is
 * single-process); the math per pair is unchanged. */
int64_t da_shard_rows(int64_t n, int world, int is_nw);
int64_t da_shard_ld(int64_t n, int world, int is_nw);
int da_dev_mh_compare_shard(const uint32_t *d_planes, int plane_bits, int64_t n, int n_hash,
                            int rank, int world, uint16_t *d_local, int64_t ld, void *stream);
int da_dev_nw_shard(const uint8_t *d_codes, const int64_t *d_offsets, int64_t n, int64_t max_len,
                    int matrix_id, int gap_open, int gap_ext, int rank, int world,
                    uint16_t *d_local, int64_t ld, void *stream);
/* d_gathered: world * da_shard_rows() rows of ld_g >= da_shard_ld() uint16 (the all-gather output).
 * Writes out[i][j] for all i,j < n: MH count/n_hash, NW (v>>8)/(v&255), mirrored. */
int da_dev_finalize_shards(const uint16_t *d_gathered, int64_t ld_g, int64_t n, int world,
                           int is_nw, int n_hash, double *d_out, int64_t ld_out, void *stream);

/* The MH exchange in value_bits = bits(n_hash) <= 16 bits per count instead of 16 (9 at n_hash = 500;
 * the all-gather is what the multi-GPU MH step waits for): da_dev_pack_shard turns a rank's uint16
 * block into a byte plane (low 8 bits) + value_bits - 8 bit planes, da_shard_packed_bytes() bytes in
 * all, 8-byte aligned; one all-gather of those; da_dev_finalize_shards_packed expands the gathered
 * world * da_shard_packed_bytes() bytes to the dense float64 matrix like da_dev_finalize_shards. */
int64_t da_shard_packed_bytes(int64_t n, int world, int value_bits);
int da_dev_pack_shard(const uint16_t *d_local, int64_t ld, int64_t n, int world, int value_bits,
                      uint8_t *d_packed, void *stream);
int da_dev_finalize_shards_packed(const uint8_t *d_gathered, int64_t n, int world, int value_bits,
                                  int n_hash, double *d_out, int64_t ld_out, void *stream);

/* ---- threshold + sparsify: what clusterbreak does right after sim_fn ---------
 * reference R/clusterbreak.R:219-221 (+ netcluster's graph_from_adjacency_matrix
 * mode = "upper", :122-124):
 *     threshold <- quantile(pep.sim[upper.tri(pep.sim)], thresh_p)      (R type 7)
 *     pep.sim[pep.sim < threshold] <- 0
 * MinHash similarities take n_hash+1 distinct values, so a device-side histogram
 * of the match counts yields that quantile exactly and only the surviving
 * upper-triangle entries (i <= j, diagonal = 1.0 included; zero weights are no
 * edges) leave the GPU, as a (i, j, weight) list sorted by (i, j), 0-based.
 * Call with ei = ej = ew = NULL to obtain threshold and edge count first. */
int da_similarity_mh_edges(const uint8_t *residues, const int64_t *offsets, int64_t n,
                           int k, int n_hash, const uint32_t *seeds, double thresh_p,
                           double *threshold_out, int64_t *n_edges_out,
                           int64_t capacity, int32_t *ei, int32_t *ej, double *ew);
/* The same for similarityNW: matches / length takes few distinct values too (the uint16 code
 * matches << 8 | length), so threshold and edge list come from a histogram of the codes.
 * Sequences up to 127 residues; an empty sequence is refused (its similarities are NaN and
 * R's quantile() stops on NaN).  Errors of da_similarity_nw apply unchanged. */
int da_similarity_nw_edges(const uint8_t *residues, const int64_t *offsets, int64_t n,
                           const char *matrix_name, int gap_open, int gap_ext, double thresh_p,
                           double *threshold_out, int64_t *n_edges_out,
                           int64_t capacity, int32_t *ei, int32_t *ej, double *ew);
/* One-pass form of the two functions above: *_begin runs the pipeline ONCE and parks the sorted edge list in a
 * handle; the caller sizes its vectors from n_edges_out, copies with da_edges_fetch and releases the handle with
 * da_edges_free (the size-query-then-fill convention above runs the whole pipeline twice).  The handle is the one
 * library-owned object of this ABI; it holds host memory only. */
typedef struct da_edges da_edges;
int da_similarity_mh_edges_begin(const uint8_t *residues, const int64_t *offsets, int64_t n,
                                 int k, int n_hash, const uint32_t *seeds, double thresh_p,
                                 da_edges **handle_out, double *threshold_out, int64_t *n_edges_out);
int da_similarity_nw_edges_begin(const uint8_t *residues, const int64_t *offsets, int64_t n,
                                 const char *matrix_name, int gap_open, int gap_ext, double thresh_p,
                                 da_edges **handle_out, double *threshold_out, int64_t *n_edges_out);
int da_edges_fetch(const da_edges *handle, int64_t capacity, int32_t *ei, int32_t *ej, double *ew);
void da_edges_free(da_edges *handle);
/* R's quantile(x, p, type = 7) of the multiset {values[b] repeated hist[b] times},
 * values ascending (host arithmetic, no device needed). */
int da_quantile_type7(const uint64_t *hist, const double *values, int nbins, double p, double *q_out);
/* device pieces: histogram of the strict upper triangle of an n x n uint16 matrix
 * (caller zeroes d_hist[nbins]); append of entries flagged in d_keep[nbins]
 * (caller zeroes *d_count; entries beyond `capacity` are counted, not stored). */
int da_dev_upper_histogram(const uint16_t *d_compact, int64_t ld, int64_t n, int nbins,
                           uint64_t *d_hist, void *stream);
int da_dev_extract_edges(const uint16_t *d_compact, int64_t ld, int64_t n, const uint8_t *d_keep,
                         int nbins, int include_diagonal, int32_t *d_i, int32_t *d_j, uint16_t *d_v,
                         int64_t capacity, uint64_t *d_count, void *stream);

/* The same two steps on one rank's folded shard block (da_dev_mh_compare_shard output).  Every
 * unordered pair lives on exactly one rank: all-reduce the n_hash+1 histogram words, derive the
 * threshold, and each rank extracts its own (disjoint) edges -- no N x N exchange. */
int da_dev_shard_histogram(const uint16_t *d_local, int64_t ld, int64_t n, int rank, int world,
                           int nbins, uint64_t *d_hist, void *stream);
int da_dev_shard_extract_edges(const uint16_t *d_local, int64_t ld, int64_t n, int rank, int world,
                               const uint8_t *d_keep, int nbins, int include_diagonal,
                               int32_t *d_i, int32_t *d_j, uint16_t *d_v, int64_t capacity,
                               uint64_t *d_count, void *stream);

/* ---- the caller's clustering step (reference R/clusterbreak.R:112-136, netcluster) ------------
 * igraph::cluster_louvain(graph_from_adjacency_matrix(S, mode = "upper", weighted = TRUE),
 *                         weights = E(g)$weight, resolution = 1.05)$membership
 * on an edge list (i, j, weight), 0-based, i == j = self-loop (the thresholded matrix keeps its 1.0
 * diagonal, R/clusterbreak.R:221), (i, j) and (j, i) the same undirected edge.  HOST code, like igraph in
 * the reference: multilevel modularity optimisation with igraph's conventions (loops count twice in a
 * vertex's strength; the last level's membership, renumbered 1.. by first appearance).  Deterministic in
 * (graph, seed) whatever the order of the edge arrays; igraph's own result depends on R's RNG stream and
 * is not reproducible outside R.  membership_out: n_vertices ids starting at 1; modularity_out /
 * levels_out may be NULL. */
int da_louvain(int64_t n_vertices, int64_t n_edges, const int32_t *ei, const int32_t *ej, const double *ew,
               double resolution, uint32_t seed, int32_t *membership_out, double *modularity_out,
               int32_t *levels_out);
/* The same clustering from a graph that is already canonical: symmetric CSR (both directions of every edge, columns ascending and
 * distinct, no diagonal entries), weights as codes into values[n_values], self-loops per vertex as codes (0xFFFF = none; NULL = none).
 * Identical result to da_louvain on the corresponding edge list.  da_dev_edges_to_csr builds exactly this on the device from the
 * (i <= j, code) list of da_dev_extract_edges[_rows]: d_work = da_dev_edges_to_csr_bytes(n_edges, n) bytes (256-byte aligned),
 * d_ptr[n + 1], d_adj / d_codes with room for 2 n_edges entries, d_loops[n]; d_ptr[n] = number of entries written. */
int da_louvain_csr(int64_t n_vertices, const int64_t *ptr, const int32_t *adj, const uint16_t *codes, const uint16_t *loop_codes,
                   const double *values, int32_t n_values, double resolution, uint32_t seed, int32_t *membership_out,
                   double *modularity_out, int32_t *levels_out);
size_t da_dev_edges_to_csr_bytes(int64_t n_edges, int64_t n);
int da_dev_edges_to_csr(const int32_t *d_i, const int32_t *d_j, const uint16_t *d_v, int64_t n_edges, int64_t n, void *d_work,
                        size_t work_bytes, int64_t *d_ptr, int32_t *d_adj, uint16_t *d_codes, uint16_t *d_loops, void *stream);

/* name -> id for da_dev_nw; -1 + DA_ERR_BAD_MATRIX message when unknown. */
int da_matrix_id(const char *matrix_name);

/* Fill the strict lower triangle of an n x n device matrix from its upper
 * triangle (after a row-sharded gather of upper-triangular row blocks). */
int da_dev_symmetrize(void *d_mat, int64_t n, int64_t ld, int kind, void *stream);

/* Widen a compact uint16 block to double on the device:
 *   MH: count / n_hash      NW: (v >> 8) / (v & 255)
 * `is_nw` selects the rule; n_hash ignored for NW. */
int da_dev_widen(const uint16_t *d_in, double *d_out, int64_t count, int is_nw, int n_hash, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DYNAALIGN_H */
