// similarity_glue.cpp -- Rcpp glue that replaces the BODIES of the reference's
// src/minHash.cpp (similarityMH, :119-188) and src/pairwiseSeqAlign.cpp
// (similarityNW, :331-365) with calls into libdynaalign_hip.so (include/dynaalign.h).
//
// Drop-in recipe: delete those two files from the reference's src/, add this
// file and the Makevars next to it and re-run Rcpp::compileAttributes().  The two exported
// signatures the reference already has (similarityMH, similarityNW) are kept verbatim, so their
// generated shims come out byte-identical; the two ADDED exports (similarityMH_edges,
// similarityNW_edges) get new shims in src/RcppExports.cpp + R/RcppExports.R and need
// `export(similarityMH_edges)` / `export(similarityNW_edges)` in NAMESPACE (roxygen writes them
// from the @export tags below).  Without compileAttributes() only the two original functions
// are callable.
//
// NOT compiled in the build container (R and Rcpp are absent there); it is the binding a
// maintainer adds.  Everything it calls is exercised through the same C ABI by
// dynaalign_amd/similarity.py and the test-suite.
#include <Rcpp.h>

#include <cmath>
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
  if (!Rf_isNull(opt)) {
    // reduce modulo 2^32 through a 64-bit integer: a direct double -> uint32_t cast of a negative or
    // > 2^32 value is undefined behaviour
    const double d = std::floor(as<double>(opt));
    if (!(std::fabs(d) < 9.0e15)) Rcpp::stop("options(DynaAlign.seed) must be a finite integer");
    return (uint32_t)((uint64_t)(int64_t)d & 0xffffffffu);
  }
  if (const char *e = std::getenv("DYNAALIGN_SEED")) return (uint32_t)(std::strtoull(e, nullptr, 10) & 0xffffffffu);
  return da_random_seed();
}

// options(DynaAlign.devices = c(0L, 1L, ...)) -- HIP device ordinals this ONE R process drives (a host thread per
// device inside the library) -- and options(DynaAlign.exchange = "rows" | "allgather" | "peercopy").  Unset: the
// current device, exactly as before.  DYNAALIGN_DEVICES=0,1,.. is the environment form.
struct DeviceOpts {
  std::vector<int32_t> devices;
  da_opts opts;
  bool set = false;
  DeviceOpts() {
    Environment base = Environment::base_env();
    Function getOption = base["getOption"];
    SEXP dv = getOption("DynaAlign.devices");
    if (!Rf_isNull(dv)) {
      IntegerVector v = as<IntegerVector>(dv);
      devices.assign(v.begin(), v.end());
      set = true;
    } else if (const char *e = std::getenv("DYNAALIGN_DEVICES")) {
      for (const char *p = e; *p;) {
        char *end = nullptr;
        const long d = std::strtol(p, &end, 10);
        if (end == p) break;
        devices.push_back((int32_t)d);
        p = (*end == ',') ? end + 1 : end;
      }
      set = !devices.empty();
    }
    int exchange = DA_EXCHANGE_ROWS;
    SEXP ex = getOption("DynaAlign.exchange");
    if (!Rf_isNull(ex)) {
      const std::string x = as<std::string>(ex);
      if (x == "rows") exchange = DA_EXCHANGE_ROWS;
      else if (x == "allgather") exchange = DA_EXCHANGE_ALLGATHER;
      else if (x == "peercopy") exchange = DA_EXCHANGE_PEERCOPY;
      else Rcpp::stop("options(DynaAlign.exchange) must be \"rows\", \"allgather\" or \"peercopy\"");
    }
    opts.struct_size = (uint32_t)sizeof(da_opts);
    opts.n_devices = (int32_t)devices.size();
    opts.devices = devices.empty() ? nullptr : devices.data();
    opts.exchange = exchange;
    opts.reserved = 0;
    opts.phase_ms = nullptr;
  }
  const da_opts *get() const { return set ? &opts : nullptr; }   // NULL = da_similarity_mh / _nw on the current device
};

struct EdgesHandle {   // da_edges_free on scope exit
  da_edges *h;
  explicit EdgesHandle(da_edges *p) : h(p) {}
  ~EdgesHandle() { da_edges_free(h); }
  EdgesHandle(const EdgesHandle &) = delete;
  EdgesHandle &operator=(const EdgesHandle &) = delete;
};

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
    check(da_similarity_mh(in.residues.data(), in.offsets.data(), n, k, n_hash, seeds.data(), nullptr));   // always an error here
  }
  NumericMatrix out(n, n);  // column-major; the result is symmetric, so layout does not matter
  const DeviceOpts dev;
  check(da_similarity_mh_opts(in.residues.data(), in.offsets.data(), n, k, n_hash, seeds.data(), REAL(out), dev.get()));
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
  const DeviceOpts dev;
  check(da_similarity_nw_opts(in.residues.data(), in.offsets.data(), n, matrixName.c_str(), gapOpen, gapExt,
                              n > 0 ? REAL(out) : nullptr, dev.get()));
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
  da_edges *h = nullptr;                  // one pass: the pipeline runs once, the edges wait in the handle
  check(da_similarity_mh_edges_begin(in.residues.data(), in.offsets.data(), n, k, n_hash, seeds.data(), thresh_p, &h, &thr, &m));
  const EdgesHandle guard(h);             // released even if the allocations below throw
  IntegerVector from(m), to(m);
  NumericVector weight(m);
  check(da_edges_fetch(h, m, INTEGER(from), INTEGER(to), REAL(weight)));
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
  da_edges *h = nullptr;
  check(da_similarity_nw_edges_begin(in.residues.data(), in.offsets.data(), n, matrixName.c_str(), gapOpen, gapExt, thresh_p,
                                     &h, &thr, &m));
  const EdgesHandle guard(h);
  IntegerVector from(m), to(m);
  NumericVector weight(m);
  check(da_edges_fetch(h, m, INTEGER(from), INTEGER(to), REAL(weight)));
  for (int64_t e = 0; e < m; ++e) { from[e] += 1; to[e] += 1; }   // R is 1-based
  return List::create(_["threshold"] = thr,
                      _["edges"] = DataFrame::create(_["from"] = from, _["to"] = to, _["weight"] = weight));
}
