/*
 * dynaalign_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's all-pairs similarity hot path
 * (similarityMH / similarityNW).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (dynaalign_amd/, include/dynaalign.h) never links, imports or calls it.
 *
 * PARITY PINNING STATUS -- read this:
 *   The reference has no tests, golden vectors or fixtures for this path
 *   (SURVEY.md section 4 / 8c), and it cannot be built in this container:
 *   both translation units include <Rcpp.h>, R/Rcpp are absent, and writing
 *   a stand-in header is not allowed.  By the letter of the contract this
 *   oracle is therefore "parity unpinned" against a reference run made here.
 *   What it IS pinned to (tests/test_oracle_golden.py):
 *     - the known-answer values SURVEY.md A.3 records from the reference
 *       (murmur3, HashFamily signatures, generate_kmers counts, the 4x4 NW
 *       matrix, the order-asymmetric pair, empty/invalid edge cases, the
 *       641-probe evp_peparray NW checksum),
 *     - published vectors of the two standard algorithms the reference
 *       embeds (MurmurHash3_x86_32; std::mt19937, 10000th output 4123659995),
 *     - a second, independently written traceback-free NW in numpy-free
 *       Python (tests/nw_model.py) used as a cross-check.
 *
 * Each function cites the reference lines it restates (paths relative to
 * /root/reference).
 */
#ifndef DYNAALIGN_ORACLE_H
#define DYNAALIGN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* same numeric values as include/dynaalign.h so tests can compare codes */
enum {
  ORC_OK = 0,
  ORC_ERR_EMPTY_INPUT = 1,     /* src/minHash.cpp:121-123 */
  ORC_ERR_BAD_K = 2,           /* src/minHash.cpp:125-127 */
  ORC_ERR_BAD_NHASH = 3,       /* src/minHash.cpp:129-131 */
  ORC_ERR_BAD_MATRIX = 4,      /* src/pairwiseSeqAlign.cpp:204 */
  ORC_ERR_BAD_RESIDUE_SEQ1 = 5,/* src/pairwiseSeqAlign.cpp:241-243 */
  ORC_ERR_BAD_RESIDUE_SEQ2 = 6,/* src/pairwiseSeqAlign.cpp:248-250 */
  ORC_ERR_NOMEM = 7
};

/* src/minHash.cpp:21-64 */
uint32_t orc_murmur3_32(const uint8_t *key, size_t len, uint32_t seed);

/* src/minHash.cpp:73-81: seeds[h] = h-th raw output of std::mt19937(seed). */
void orc_mt19937_seeds(uint32_t seed, int n, uint32_t *out);

/* src/minHash.cpp:92-105: number of k-mers generate_kmers() would return. */
int64_t orc_num_kmers(int64_t len, int k);

/* src/minHash.cpp:140-157.  sig is [n][n_hash], row-major. */
int orc_minhash_signatures(const uint8_t *residues, const int64_t *offsets,
                           int64_t n, int k, int n_hash, const uint32_t *seeds,
                           uint32_t *sig);

/* src/minHash.cpp:160-178, rows [row_begin,row_end) x all n columns, match
 * COUNTS (the integer before the divide at :174); diagonal = n_hash. */
void orc_mh_counts_rows(const uint32_t *sig, int64_t n, int n_hash,
                        int64_t row_begin, int64_t row_end, uint16_t *counts);

/* src/minHash.cpp:119-188 with the seed made explicit (HashFamily(n_hash,
 * seed), :73).  out is n*n doubles (symmetric, so row/column-major agree).
 * Loop structure and the two OpenMP sites follow the reference. */
int orc_similarity_mh(const uint8_t *residues, const int64_t *offsets, int64_t n,
                      int k, int n_hash, const uint32_t *seeds, double *out);

/* orc_similarity_mh with the reference's data structures (row-pointer signature storage, copied k-mers, column-major
 * element stores): the CPU baseline SURVEY.md 8(d) specifies.  Same result bits. */
int orc_similarity_mh_rowptr(const uint8_t *residues, const int64_t *offsets, int64_t n,
                             int k, int n_hash, const uint32_t *seeds, double *out);

/* -1 if name is not one of the six tables (src/pairwiseSeqAlign.cpp:190-206) */
int orc_matrix_id(const char *name);
const signed char *orc_matrix_table(int id); /* 576 scores, row-major */
/* src/pairwiseSeqAlign.cpp:15-21: 0..23, or -1 for any other byte */
int orc_aa_index(uint8_t c);

/* src/pairwiseSeqAlign.cpp:209-313, full matrices + traceback as written.
 * Returns ORC_OK or ORC_ERR_BAD_RESIDUE_SEQ1/2 (bad_char set).  matches and
 * alen are the two integers divided at :311; score is M[m][n] (never exposed
 * by the reference; auxiliary). */
int orc_nw_pair(const uint8_t *s1, int64_t m, const uint8_t *s2, int64_t n,
                const signed char *table, int gap_open, int gap_ext,
                int32_t *matches, int32_t *alen, int32_t *score,
                uint8_t *bad_char);

/* src/pairwiseSeqAlign.cpp:331-365 (serial i, j from i, mirror store).
 * errbuf receives the reference's message text on error. */
int orc_similarity_nw(const uint8_t *residues, const int64_t *offsets, int64_t n,
                      const char *matrix_name, int gap_open, int gap_ext,
                      double *out, char *errbuf, size_t errlen);

/* rows [row_begin,row_end) x columns [0,n): (matches, alen, score) of
 * calc(seq[min(i,j)], seq[max(i,j)]) -- SURVEY fact 3.  Any of the three
 * output pointers may be NULL.  Arrays are [rows][n]. */
int orc_nw_rows(const uint8_t *residues, const int64_t *offsets, int64_t n,
                int64_t row_begin, int64_t row_end, const char *matrix_name,
                int gap_open, int gap_ext, int32_t *matches, int32_t *alen,
                int32_t *score, char *errbuf, size_t errlen);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
