/*
 * dynaalign_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See dynaalign_oracle.h for the pinning status ("parity unpinned" against a
 * reference build made here; pinned to SURVEY.md A.3 known answers and the
 * published MurmurHash3 / mt19937 vectors).
 *
 * Plain C99 (+ OpenMP at the reference's two sites).  Written from the
 * algorithm description; nothing here is copied from the reference.
 */
#include "dynaalign_oracle.h"

#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "blosum_data.inc"

/* ------------------------------------------------------------------ hash */

static inline uint32_t rol32(uint32_t v, int r) { return (v << r) | (v >> (32 - r)); }

/* MurmurHash3_x86_32 over `len` bytes.  Reference: src/minHash.cpp:21-64
 * (constants :22-27, 4-byte little-endian body :34-41, tail :43-54,
 * length mix + avalanche :56-61). */
uint32_t orc_murmur3_32(const uint8_t *key, size_t len, uint32_t seed) {
  const uint32_t C1 = 0xcc9e2d51u, C2 = 0x1b873593u;
  uint32_t h = seed;
  size_t nblk = len >> 2;
  for (size_t b = 0; b < nblk; ++b) {
    const uint8_t *p = key + 4 * b;
    uint32_t w = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) |
                 ((uint32_t)p[3] << 24);
    w *= C1;
    w = rol32(w, 15);
    w *= C2;
    h ^= w;
    h = rol32(h, 13) * 5u + 0xe6546b64u;
  }
  size_t rem = len & 3;
  if (rem) {
    const uint8_t *t = key + 4 * nblk;
    uint32_t w = 0;
    if (rem == 3) w ^= (uint32_t)t[2] << 16;
    if (rem >= 2) w ^= (uint32_t)t[1] << 8;
    w ^= (uint32_t)t[0];
    w *= C1;
    w = rol32(w, 15);
    w *= C2;
    h ^= w;
  }
  h ^= (uint32_t)len;
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}

/* std::mt19937 (ISO C++ [rand.predef]: w=32 n=624 m=397 r=31 a=0x9908b0df
 * u=11 d=0xffffffff s=7 b=0x9d2c5680 t=15 c=0xefc60000 l=18 f=1812433253).
 * The reference draws through uniform_int_distribution<uint32_t> with the full
 * range (src/minHash.cpp:75-80), which on libstdc++ returns the raw draw
 * ("Third-party arithmetic dependencies"). */
void orc_mt19937_seeds(uint32_t seed, int n, uint32_t *out) {
  uint32_t mt[624];
  mt[0] = seed;
  for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
  int idx = 624;
  for (int o = 0; o < n; ++o) {
    if (idx == 624) {
      for (int i = 0; i < 624; ++i) {
        uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        uint32_t v = mt[(i + 397) % 624] ^ (y >> 1);
        if (y & 1u) v ^= 0x9908b0dfu;
        mt[i] = v;
      }
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    out[o] = y;
  }
}

/* src/minHash.cpp:92-105: len-k+1 windows when len >= k (k>0), else none. */
int64_t orc_num_kmers(int64_t len, int k) {
  if (k <= 0 || len < (int64_t)k) return 0;
  return len - k + 1;
}

/* ------------------------------------------------------------- MinHash */

/* src/minHash.cpp:140-157: sig[i][h] = min over windows of murmur3(window,
 * seeds[h]); identity UINT32_MAX.  OpenMP over sequences as at :143-146. */
int orc_minhash_signatures(const uint8_t *residues, const int64_t *offsets,
                           int64_t n, int k, int n_hash, const uint32_t *seeds,
                           uint32_t *sig) {
  if (n <= 0) return ORC_ERR_EMPTY_INPUT;
  if (k <= 0) return ORC_ERR_BAD_K;
  if (n_hash <= 0) return ORC_ERR_BAD_NHASH;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int64_t i = 0; i < n; ++i) {
    uint32_t *row = sig + (size_t)i * (size_t)n_hash;
    for (int h = 0; h < n_hash; ++h) row[h] = UINT32_MAX;
    const uint8_t *s = residues + offsets[i];
    int64_t nk = orc_num_kmers(offsets[i + 1] - offsets[i], k);
    for (int64_t p = 0; p < nk; ++p)
      for (int h = 0; h < n_hash; ++h) {
        uint32_t v = orc_murmur3_32(s + p, (size_t)k, seeds[h]);
        if (v < row[h]) row[h] = v;
      }
  }
  return ORC_OK;
}

/* src/minHash.cpp:168-173 for one ordered pair */
static inline int mh_matches(const uint32_t *a, const uint32_t *b, int n_hash) {
  int c = 0;
  for (int h = 0; h < n_hash; ++h) c += (a[h] == b[h]);
  return c;
}

void orc_mh_counts_rows(const uint32_t *sig, int64_t n, int n_hash,
                        int64_t row_begin, int64_t row_end, uint16_t *counts) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int64_t i = row_begin; i < row_end; ++i) {
    uint16_t *o = counts + (size_t)(i - row_begin) * (size_t)n;
    const uint32_t *a = sig + (size_t)i * (size_t)n_hash;
    for (int64_t j = 0; j < n; ++j)
      o[j] = (uint16_t)(i == j ? n_hash : mh_matches(a, sig + (size_t)j * (size_t)n_hash, n_hash));
  }
}

/* src/minHash.cpp:119-188 with an explicit seed vector.  Same validation order
 * (:121-131), same serial-i / parallel-j compare nest (:160-178), same
 * (double)matches / n_hash (:174), diagonal forced to 1.0 (:161). */
int orc_similarity_mh(const uint8_t *residues, const int64_t *offsets, int64_t n,
                      int k, int n_hash, const uint32_t *seeds, double *out) {
  if (n <= 0) return ORC_ERR_EMPTY_INPUT;
  if (k <= 0) return ORC_ERR_BAD_K;
  if (n_hash <= 0) return ORC_ERR_BAD_NHASH;
  uint32_t *sig = (uint32_t *)malloc((size_t)n * (size_t)n_hash * sizeof(uint32_t));
  if (!sig) return ORC_ERR_NOMEM;
  orc_minhash_signatures(residues, offsets, n, k, n_hash, seeds, sig);
  for (int64_t i = 0; i < n; ++i) {
    out[(size_t)i * n + i] = 1.0;
    const uint32_t *a = sig + (size_t)i * (size_t)n_hash;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t j = i + 1; j < n; ++j) {
      int c = mh_matches(a, sig + (size_t)j * (size_t)n_hash, n_hash);
      double s = (double)c / n_hash;
      out[(size_t)i * n + j] = s;
      out[(size_t)j * n + i] = s;
    }
  }
  free(sig);
  return ORC_OK;
}

/* The same call with the reference's DATA STRUCTURES (what SURVEY.md 8(d) names as the CPU baseline; bench.py times it):
 *   - signatures[n][n_hash] as n separately allocated rows behind a row-pointer array (std::vector<std::vector<uint32_t>>,
 *     src/minHash.cpp:140),
 *   - per sequence, every k-mer COPIED out of the sequence first (generate_kmers builds a std::vector<std::string>,
 *     src/minHash.cpp:92-105), then the k-mer-outer / hash-inner loop of :151-156 with the bounds-checked HashFamily::hash
 *     call (:83-88) -- here a function call through the seed table,
 *   - the result written through a column-major accessor M(i,j) = out[i + j*n]: one of the two stores per pair is
 *     stride-n (:175-176).
 * Same two OpenMP sites, same arithmetic, same result bits as orc_similarity_mh. */
static uint32_t hashfamily_hash(const uint32_t *seeds, int n_hash, const uint8_t *key, size_t len, int idx, int *oob) {
  if (idx < 0 || idx >= n_hash) { *oob = 1; return 0; }                 /* src/minHash.cpp:84-86 (Rcpp::stop there) */
  return orc_murmur3_32(key, len, seeds[idx]);
}
int orc_similarity_mh_rowptr(const uint8_t *residues, const int64_t *offsets, int64_t n,
                             int k, int n_hash, const uint32_t *seeds, double *out) {
  if (n <= 0) return ORC_ERR_EMPTY_INPUT;
  if (k <= 0) return ORC_ERR_BAD_K;
  if (n_hash <= 0) return ORC_ERR_BAD_NHASH;
  uint32_t **sig = (uint32_t **)calloc((size_t)n, sizeof(uint32_t *));
  if (!sig) return ORC_ERR_NOMEM;
  int rc = ORC_OK;
  for (int64_t i = 0; i < n; ++i) {                                       /* :140 */
    sig[i] = (uint32_t *)malloc((size_t)n_hash * sizeof(uint32_t));
    if (!sig[i]) { rc = ORC_ERR_NOMEM; break; }
    for (int h = 0; h < n_hash; ++h) sig[i][h] = UINT32_MAX;
  }
  if (rc == ORC_OK) {
    int oob_any = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < n; ++i) {                                     /* :143-157 */
      const int64_t len = offsets[i + 1] - offsets[i];
      const int64_t nk = orc_num_kmers(len, k);
      uint8_t **kmers = (uint8_t **)malloc((size_t)(nk > 0 ? nk : 1) * sizeof(uint8_t *));
      for (int64_t p = 0; p < nk; ++p) {                                  /* :99-103: one string per k-mer */
        kmers[p] = (uint8_t *)malloc((size_t)k + 1);
        memcpy(kmers[p], residues + offsets[i] + p, (size_t)k);
        kmers[p][k] = 0;
      }
      for (int64_t p = 0; p < nk; ++p)
        for (int h = 0; h < n_hash; ++h) {
          int oob = 0;
          uint32_t v = hashfamily_hash(seeds, n_hash, kmers[p], (size_t)k, h, &oob);
          if (oob) oob_any = 1;
          if (v < sig[i][h]) sig[i][h] = v;                               /* :154 */
        }
      for (int64_t p = 0; p < nk; ++p) free(kmers[p]);
      free(kmers);
    }
    (void)oob_any;
    for (int64_t i = 0; i < n; ++i) {                                     /* :160-178 */
      out[(size_t)i + (size_t)i * (size_t)n] = 1.0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
      for (int64_t j = i + 1; j < n; ++j) {
        int matches = 0;
        for (int h = 0; h < n_hash; ++h)
          if (sig[i][h] == sig[j][h]) ++matches;
        double s = (double)matches / n_hash;
        out[(size_t)i + (size_t)j * (size_t)n] = s;                       /* M(i, j), column-major */
        out[(size_t)j + (size_t)i * (size_t)n] = s;                       /* M(j, i) */
      }
    }
  }
  for (int64_t i = 0; i < n; ++i) free(sig[i]);
  free(sig);
  return rc;
}

/* ------------------------------------------------------------------ NW */

int orc_matrix_id(const char *name) {
  if (!name) return -1;
  for (int i = 0; i < ORC_NUM_MATRICES; ++i)
    if (strcmp(name, orc_matrix_names[i]) == 0) return i;
  return -1;
}

const signed char *orc_matrix_table(int id) {
  return (id >= 0 && id < ORC_NUM_MATRICES) ? orc_matrix_data[id] : NULL;
}

/* src/pairwiseSeqAlign.cpp:15-21 */
int orc_aa_index(uint8_t c) {
  static const char order[] = "ARNDCQEGHILKMFPSTWYVBZX*";
  if (c == 0) return -1;
  const char *p = strchr(order, (int)c);
  return p ? (int)(p - order) : -1;
}

/* int32 arithmetic with two's-complement wrap (what the reference's int
 * expressions do on every value it can reach without UB). */
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static inline int32_t max2(int32_t a, int32_t b) { return a > b ? a : b; }

/* src/pairwiseSeqAlign.cpp:209-313.  Three score planes + a move plane, all
 * (m+1)x(n+1), allocated per pair like the reference (:216-219); boundary
 * :222-235; row-major fill with lazy residue validation :238-281; walk back
 * from (m,n) :284-308. */
int orc_nw_pair(const uint8_t *s1, int64_t m, const uint8_t *s2, int64_t n,
                const signed char *table, int gap_open, int gap_ext,
                int32_t *matches, int32_t *alen, int32_t *score, uint8_t *bad_char) {
  const int32_t NEG = INT_MIN / 2;
  const size_t W = (size_t)n + 1, cells = ((size_t)m + 1) * W;
  int32_t *M = (int32_t *)malloc(cells * sizeof(int32_t));
  int32_t *X = (int32_t *)malloc(cells * sizeof(int32_t));
  int32_t *Y = (int32_t *)malloc(cells * sizeof(int32_t));
  char *mv = (char *)malloc(cells);
  if (!M || !X || !Y || !mv) { free(M); free(X); free(Y); free(mv); return ORC_ERR_NOMEM; }
  for (size_t c = 0; c < cells; ++c) { M[c] = X[c] = Y[c] = NEG; mv[c] = '0'; }
  M[0] = 0;
  for (int64_t i = 1; i <= m; ++i) {            /* first column */
    X[(size_t)i * W] = wsub(-gap_open, wmul((int32_t)(i - 1), gap_ext));
    mv[(size_t)i * W] = 'U';
  }
  for (int64_t j = 1; j <= n; ++j) {            /* first row */
    Y[j] = wsub(-gap_open, wmul((int32_t)(j - 1), gap_ext));
    mv[j] = 'L';
  }
  const int32_t open_ext = wadd(gap_open, gap_ext);
  int rc = ORC_OK;
  for (int64_t i = 1; i <= m && rc == ORC_OK; ++i) {
    int ia = orc_aa_index(s1[i - 1]);
    if (ia < 0) { rc = ORC_ERR_BAD_RESIDUE_SEQ1; if (bad_char) *bad_char = s1[i - 1]; break; }
    for (int64_t j = 1; j <= n; ++j) {
      int ib = orc_aa_index(s2[j - 1]);
      if (ib < 0) { rc = ORC_ERR_BAD_RESIDUE_SEQ2; if (bad_char) *bad_char = s2[j - 1]; break; }
      const int32_t sub = table[ia * 24 + ib];
      const size_t c = (size_t)i * W + (size_t)j, up = c - W, lf = c - 1, dg = up - 1;
      const int32_t x = max2(wsub(M[up], open_ext), wsub(X[up], gap_ext));
      const int32_t y = max2(wsub(M[lf], open_ext), wsub(Y[lf], gap_ext));
      const int32_t d = max2(max2(wadd(M[dg], sub), wadd(X[dg], sub)), wadd(Y[dg], sub));
      X[c] = x;
      Y[c] = y;
      if (d >= x && d >= y) { M[c] = d; mv[c] = 'D'; }
      else if (x >= y)      { M[c] = x; mv[c] = 'U'; }
      else                  { M[c] = y; mv[c] = 'L'; }
    }
  }
  if (rc == ORC_OK) {
    int32_t nm = 0, len = 0;
    int64_t i = m, j = n;
    while (i > 0 || j > 0) {
      char t = mv[(size_t)i * W + (size_t)j];
      if (t == 'D') { nm += (s1[i - 1] == s2[j - 1]); --i; --j; }
      else if (t == 'U') --i;
      else --j;
      ++len;
    }
    if (matches) *matches = nm;
    if (alen) *alen = len;
    if (score) *score = M[cells - 1];
  }
  free(M); free(X); free(Y); free(mv);
  return rc;
}

static void nw_errmsg(int rc, uint8_t bad, const char *name, char *buf, size_t len) {
  if (!buf || !len) return;
  if (rc == ORC_ERR_BAD_MATRIX) snprintf(buf, len, "Invalid substitution matrix name: %s", name ? name : "");
  else if (rc == ORC_ERR_BAD_RESIDUE_SEQ1) snprintf(buf, len, "Invalid amino acid in sequence1: %c", (char)bad);
  else if (rc == ORC_ERR_BAD_RESIDUE_SEQ2) snprintf(buf, len, "Invalid amino acid in sequence2: %c", (char)bad);
  else buf[0] = 0;
}

/* src/pairwiseSeqAlign.cpp:331-365 */
int orc_similarity_nw(const uint8_t *residues, const int64_t *offsets, int64_t n,
                      const char *matrix_name, int gap_open, int gap_ext,
                      double *out, char *errbuf, size_t errlen) {
  int id = orc_matrix_id(matrix_name);
  if (id < 0) { nw_errmsg(ORC_ERR_BAD_MATRIX, 0, matrix_name, errbuf, errlen); return ORC_ERR_BAD_MATRIX; }
  const signed char *tab = orc_matrix_data[id];
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = i; j < n; ++j) {
      int32_t nm, len; uint8_t bad = 0;
      int rc = orc_nw_pair(residues + offsets[i], offsets[i + 1] - offsets[i],
                           residues + offsets[j], offsets[j + 1] - offsets[j],
                           tab, gap_open, gap_ext, &nm, &len, NULL, &bad);
      if (rc != ORC_OK) { nw_errmsg(rc, bad, matrix_name, errbuf, errlen); return rc; }
      double s = (double)nm / len;
      out[(size_t)i * n + j] = s;
      out[(size_t)j * n + i] = s;
    }
  return ORC_OK;
}

int orc_nw_rows(const uint8_t *residues, const int64_t *offsets, int64_t n,
                int64_t row_begin, int64_t row_end, const char *matrix_name,
                int gap_open, int gap_ext, int32_t *matches, int32_t *alen,
                int32_t *score, char *errbuf, size_t errlen) {
  int id = orc_matrix_id(matrix_name);
  if (id < 0) { nw_errmsg(ORC_ERR_BAD_MATRIX, 0, matrix_name, errbuf, errlen); return ORC_ERR_BAD_MATRIX; }
  const signed char *tab = orc_matrix_data[id];
  int rc_all = ORC_OK;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
  for (int64_t i = row_begin; i < row_end; ++i)
    for (int64_t j = 0; j < n; ++j) {
      int64_t a = i < j ? i : j, b = i < j ? j : i;   /* lower index is sequence1 */
      int32_t nm = 0, len = 0, sc = 0; uint8_t bad = 0;
      int rc = orc_nw_pair(residues + offsets[a], offsets[a + 1] - offsets[a],
                           residues + offsets[b], offsets[b + 1] - offsets[b],
                           tab, gap_open, gap_ext, &nm, &len, &sc, &bad);
      if (rc != ORC_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
        { if (rc_all == ORC_OK) { rc_all = rc; nw_errmsg(rc, bad, matrix_name, errbuf, errlen); } }
        continue;
      }
      size_t o = (size_t)(i - row_begin) * (size_t)n + (size_t)j;
      if (matches) matches[o] = nm;
      if (alen) alen[o] = len;
      if (score) score[o] = sc;
    }
  return rc_all;
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
