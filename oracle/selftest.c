/* selftest.c -- runs the oracle's known answers under AddressSanitizer + UBSan on the CPU
 * (`make -C oracle selftest_asan && oracle/selftest_asan`).  Test infrastructure only. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dynaalign_oracle.h"

static int fails = 0;
#define CHECK(cond) do { if (!(cond)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++fails; } } while (0)

static void pack(const char **seqs, int n, uint8_t *res, int64_t *off) {
  off[0] = 0;
  for (int i = 0; i < n; ++i) {
    size_t l = strlen(seqs[i]);
    memcpy(res + off[i], seqs[i], l);
    off[i + 1] = off[i] + (int64_t)l;
  }
}

int main(void) {
  /* SURVEY A.3 / published vectors */
  CHECK(orc_murmur3_32((const uint8_t *)"ACDE", 4, 0) == 2543633677u);
  CHECK(orc_murmur3_32((const uint8_t *)"AC", 2, 0) == 635767089u);
  CHECK(orc_murmur3_32((const uint8_t *)"", 0, 0) == 0u);
  CHECK(orc_murmur3_32((const uint8_t *)"ACDEF", 5, 42) == 3294844255u);
  CHECK(orc_murmur3_32((const uint8_t *)"Hello, world!", 13, 0) == 0xc0363e43u);
  uint32_t *mt = (uint32_t *)malloc(10000 * sizeof(uint32_t));
  orc_mt19937_seeds(5489, 10000, mt);
  CHECK(mt[9999] == 4123659995u);
  orc_mt19937_seeds(1, 8, mt);
  CHECK(mt[0] == 1791095845u && mt[3] == 4005303368u);

  /* signatures KAT */
  const char *one[] = {"STSIPALTAVET"};
  uint8_t res[4096];
  int64_t off[64];
  pack(one, 1, res, off);
  uint32_t sig[8];
  static const uint32_t want4[8] = {465993298u, 462870073u, 777677u, 627250232u, 572031937u, 234667797u, 38437948u, 409760030u};
  CHECK(orc_minhash_signatures(res, off, 1, 4, 8, mt, sig) == ORC_OK);
  CHECK(memcmp(sig, want4, sizeof sig) == 0);

  /* NW 4x4 KAT, asymmetry, edges */
  const char *four[] = {"RRAVELQTVAFP", "PPPSYETVMAAA", "TPPPSYETVMAA", "TPPASYHTVMAA"};
  pack(four, 4, res, off);
  double W[16];
  char msg[128];
  CHECK(orc_similarity_nw(res, off, 4, "BLOSUM62", 10, 4, W, msg, sizeof msg) == ORC_OK);
  CHECK(W[1 * 4 + 2] == 11.0 / 13.0 && W[1 * 4 + 3] == 9.0 / 13.0 && W[2 * 4 + 3] == 10.0 / 12.0 && W[0] == 1.0);
  int32_t nm, ln, sc;
  uint8_t bad;
  const signed char *b62 = orc_matrix_table(orc_matrix_id("BLOSUM62"));
  CHECK(orc_nw_pair((const uint8_t *)"YDYIHIYADKQDRIGWLGNT", 20, (const uint8_t *)"MYCEMNVEIQYMATKNMWNT", 20, b62, 10, 4, &nm, &ln, &sc, &bad) == ORC_OK && nm == 3 && ln == 21);
  CHECK(orc_nw_pair((const uint8_t *)"MYCEMNVEIQYMATKNMWNT", 20, (const uint8_t *)"YDYIHIYADKQDRIGWLGNT", 20, b62, 10, 4, &nm, &ln, &sc, &bad) == ORC_OK && nm == 4 && ln == 21);
  CHECK(orc_nw_pair((const uint8_t *)"", 0, (const uint8_t *)"", 0, b62, 10, 4, &nm, &ln, &sc, &bad) == ORC_OK && ln == 0);
  CHECK(orc_nw_pair((const uint8_t *)"AJ", 2, (const uint8_t *)"AA", 2, b62, 10, 4, &nm, &ln, &sc, &bad) == ORC_ERR_BAD_RESIDUE_SEQ1 && bad == 'J');
  CHECK(orc_matrix_id("PAM250") == -1);

  /* ragged fuzz through every entry point (memory errors are the point here) */
  srand(7);
  enum { N = 40 };
  const char *alpha = "ARNDCQEGHILKMFPSTWYVBZX*";
  static char store[N][40];
  const char *ptrs[N];
  for (int i = 0; i < N; ++i) {
    int L = rand() % 33;
    for (int c = 0; c < L; ++c) store[i][c] = alpha[rand() % 24];
    store[i][L] = 0;
    ptrs[i] = store[i];
  }
  pack(ptrs, N, res, off);
  orc_mt19937_seeds(12345, 64, mt);
  double *M = (double *)malloc(N * N * sizeof(double));
  CHECK(orc_similarity_mh(res, off, N, 3, 64, mt, M) == ORC_OK);
  for (int i = 0; i < N; ++i) CHECK(M[i * N + i] == 1.0);
  uint32_t *S = (uint32_t *)malloc(N * 64 * sizeof(uint32_t));
  uint16_t *C = (uint16_t *)malloc(N * N * sizeof(uint16_t));
  CHECK(orc_minhash_signatures(res, off, N, 3, 64, mt, S) == ORC_OK);
  orc_mh_counts_rows(S, N, 64, 0, N, C);
  for (int i = 0; i < N * N; ++i) CHECK((double)C[i] / 64 == M[i]);
  for (int id = 0; id < 6; ++id) {
    static const char *names[] = {"BLOSUM62", "BLOSUM45", "BLOSUM50", "BLOSUM80", "BLOSUM90", "BLOSUM100"};
    int32_t *a = (int32_t *)malloc(N * N * 4), *b = (int32_t *)malloc(N * N * 4), *c = (int32_t *)malloc(N * N * 4);
    CHECK(orc_nw_rows(res, off, N, 0, N, names[id], id * 3, id, a, b, c, msg, sizeof msg) == ORC_OK);
    CHECK(orc_similarity_nw(res, off, N, names[id], id * 3, id, M, msg, sizeof msg) == ORC_OK);
    for (int i = 0; i < N * N; ++i)
      CHECK(b[i] == 0 ? isnan(M[i]) : M[i] == (double)a[i] / b[i]);
    free(a); free(b); free(c);
  }
  free(mt); free(M); free(S); free(C);
  printf(fails ? "oracle selftest: %d FAILED\n" : "oracle selftest: ok\n", fails);
  return fails != 0;
}
