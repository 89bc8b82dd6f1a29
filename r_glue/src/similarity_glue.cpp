// similarity_glue.cpp -- Rcpp glue that replaces the BODIES of the reference's
// src/minHash.cpp (similarityMH, :119-188) and src/pairwiseSeqAlign.cpp
// (similarityNW, :331-365) with calls into libdynaalign_hip.so (include/dynaalign.h).
//
// Drop-in recipe: delete those two files from the reference's src/,
// add this file and the Makevars next to it, run Rcpp::compileAttributes() (or keep the
// reference's generated src/RcppExports.cpp and R/RcppExports.R unchanged -- the exported
// signatures below are identical, so the generated code is byte-identical too).
//
// NOT compiled in the build container (R and Rcpp are absent there); it is the binding a
// maintainer adds.  Everything it calls is exercised through the same C ABI by
// dynaalign_amd/similarity.py and the test-suite.
#include <Rcpp.h>

#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "dynaalign.h"

using namespace Rcpp;

namespace {

// One pass over the STRSXP on the calling (main R) thread: bytes of every element back
// to back + offsets.  (The reference copies each element per use -- src/minHash.cpp:147 --
// from OpenMP worker threads, and src/pairwiseSeqAlign.cpp:341-343 once per pair.)
struct Packed {
  std::vector<uint8_t> residues;
  std::vector<int64_t> offsets;
  explicit Packed(const CharacterVector &x) : offsets(x.length() + 1, 0) {
    const R_xlen_t n = x.length();
    size_t total = 0;
    for (R_xlen_t i = 0; i < n; ++i) total += std::string(as<std::string>(x[i])).size();
    residues.reserve(total ? total : 1);
    for (R_xlen_t i = 0; i < n; ++i) {
      const std::string s = as<std::string>(x[i]);  // same conversion the reference uses
      residues.insert(residues.end(), s.begin(), s.end());
      offsets[i + 1] = (int64_t)residues.size();
    }
    if (residues.empty()) residues.push_back(0);
  }
};

void check(int rc) {
  if (rc != DA_OK) Rcpp::stop("%s", da_last_error());  // reference message texts come from the library
}

void set_dimnames(NumericMatrix &m) {  // src/minHash.cpp:181-185, src/pairwiseSeqAlign.cpp:356-362
  const R_xlen_t n = m.nrow();
  CharacterVector labels(n);
  for (R_xlen_t i = 0; i < n; ++i) labels[i] = std::to_string(i + 1);
  m.attr("dimnames") = List::create(labels, labels);
}

// Seed for HashFamily: the reference draws std::random_device{}() (src/minHash.cpp:73) and
// gives no way to fix it.  options(DynaAlign.seed = <int>) or DYNAALIGN_SEED=<int> make a
// run reproducible without touching the R-level signature.
uint32_t hash_seed() {
  Environment base = Environment::base_env();
  Function getOption = base["getOption"];
  SEXP opt = getOption("DynaAlign.seed");
  if (!Rf_isNull(opt)) return (uint32_t)as<double>(opt);
  if (const char *e = std::getenv("DYNAALIGN_SEED")) return (uint32_t)std::strtoul(e, nullptr, 10);
  return da_random_seed();
}

}  // namespace

//' @name similarityMH
//' @title Compute MinHash Similarity Matrix
//' @param sequences A character vector of input sequences
//' @param k The length of k-mers to use (default: 4)
//' @param n_hash Number of hash functions to use (default: 50)
//' @return A numeric matrix of pairwise similarities
//' @export
// [[Rcpp::export]]
NumericMatrix similarityMH(CharacterVector sequences, int k = 4, int n_hash = 50) {
  const Packed in(sequences);
  const int64_t n = sequences.length();
  // validation order and messages (src/minHash.cpp:121-131) are enforced by the library;
  // call it first with no output so that errors surface before the n*n allocation
  std::vector<uint32_t> seeds(n_hash > 0 ? n_hash : 1);
  if (n_hash > 0) check(da_hash_family_seeds(hash_seed(), n_hash, seeds.data()));
  if (n <= 0 || k <= 0 || n_hash <= 0) {
    check(da_similarity_mh(in.residues.data(), in.offsets.data(), n, k, n_hash, seeds.data(), nullptr));
  }
  NumericMatrix out(n, n);  // column-major; the result is symmetric, so layout does not matter
  check(da_similarity_mh(in.residues.data(), in.offsets.data(), n, k, n_hash, seeds.data(), REAL(out)));
  set_dimnames(out);
  return out;
}

//' @name similarityNW
//' @title Sequence Alignment using Needleman-Wunsch Algorithm
//' @param sequences A character vector of input sequences
//' @param matrixName A substitution matrix for scoring alignments
//' @param gapOpen penalty for opening a gap
//' @param gapExt penalty for extending a gap
//' @return A numeric matrix of pairwise similarities
//' @export
// [[Rcpp::export]]
NumericMatrix similarityNW(CharacterVector sequences, std::string matrixName = "BLOSUM62",
                           int gapOpen = 10, int gapExt = 4) {
  const Packed in(sequences);
  const int64_t n = sequences.length();
  NumericMatrix out(n, n);
  check(da_similarity_nw(in.residues.data(), in.offsets.data(), n, matrixName.c_str(), gapOpen, gapExt,
                         n > 0 ? REAL(out) : nullptr));
  set_dimnames(out);
  return out;
}

//' @name similarityMH_edges
//' @title MinHash similarity + clusterbreak's quantile threshold, as an edge list
//' @description Non-breaking addition.  Equivalent to
//'   S <- similarityMH(sequences, k, n_hash); thr <- quantile(S[upper.tri(S)], thresh_p); S[S < thr] <- 0
//' (R/clusterbreak.R:217-221) but returns only the surviving upper-triangle entries, so a 100k-peptide set
//' never materialises its 80 GB matrix:  igraph::graph_from_data_frame(res$edges, directed = FALSE).
//' @export
// [[Rcpp::export]]
List similarityMH_edges(CharacterVector sequences, int k = 4, int n_hash = 50, double thresh_p = 0.8) {
  const Packed in(sequences);
  const int64_t n = sequences.length();
  std::vector<uint32_t> seeds(n_hash > 0 ? n_hash : 1);
  if (n_hash > 0) check(da_hash_family_seeds(hash_seed(), n_hash, seeds.data()));
  double thr = 0;
  int64_t m = 0;
  check(da_similarity_mh_edges(in.residues.data(), in.offsets.data(), n, k, n_hash, seeds.data(), thresh_p, &thr, &m, 0,
                               nullptr, nullptr, nullptr));
  IntegerVector from(m), to(m);
  NumericVector weight(m);
  check(da_similarity_mh_edges(in.residues.data(), in.offsets.data(), n, k, n_hash, seeds.data(), thresh_p, &thr, &m, m,
                               INTEGER(from), INTEGER(to), REAL(weight)));
  for (int64_t e = 0; e < m; ++e) { from[e] += 1; to[e] += 1; }   // R is 1-based
  return List::create(_["threshold"] = thr,
                      _["edges"] = DataFrame::create(_["from"] = from, _["to"] = to, _["weight"] = weight));
}

//' @name similarityNW_edges
//' @title Needleman-Wunsch identity + clusterbreak's quantile threshold, as an edge list
//' @description The similarityNW counterpart of similarityMH_edges (sequences up to 127 residues, none empty).
//' @export
// [[Rcpp::export]]
List similarityNW_edges(CharacterVector sequences, std::string matrixName = "BLOSUM62", int gapOpen = 10,
                        int gapExt = 4, double thresh_p = 0.8) {
  const Packed in(sequences);
  const int64_t n = sequences.length();
  double thr = 0;
  int64_t m = 0;
  check(da_similarity_nw_edges(in.residues.data(), in.offsets.data(), n, matrixName.c_str(), gapOpen, gapExt, thresh_p,
                               &thr, &m, 0, nullptr, nullptr, nullptr));
  IntegerVector from(m), to(m);
  NumericVector weight(m);
  check(da_similarity_nw_edges(in.residues.data(), in.offsets.data(), n, matrixName.c_str(), gapOpen, gapExt, thresh_p,
                               &thr, &m, m, INTEGER(from), INTEGER(to), REAL(weight)));
  for (int64_t e = 0; e < m; ++e) { from[e] += 1; to[e] += 1; }   // R is 1-based
  return List::create(_["threshold"] = thr,
                      _["edges"] = DataFrame::create(_["from"] = from, _["to"] = to, _["weight"] = weight));
}
