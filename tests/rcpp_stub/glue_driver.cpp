// TEST-ONLY driver of r_glue/src/similarity_glue.cpp (compiled against tests/rcpp_stub/Rcpp.h, linked with
// libdynaalign_hip.so): plays R's side of the four exported functions so that tests/test_r_glue.py can compare what the glue
// hands back with the ctypes path.  Protocol: argv = <function> <args...>, sequences one per line on stdin, result as raw
// little-endian bytes on stdout, an Rcpp::exception as "ERROR: <message>" on stderr with exit code 3.
#include <Rcpp.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

using namespace Rcpp;

NumericMatrix similarityMH(CharacterVector sequences, int k, int n_hash);
NumericMatrix similarityNW(CharacterVector sequences, std::string matrixName, int gapOpen, int gapExt);
List similarityMH_edges(CharacterVector sequences, int k, int n_hash, double thresh_p);
List similarityNW_edges(CharacterVector sequences, std::string matrixName, int gapOpen, int gapExt, double thresh_p);

static void put(const void *p, size_t bytes) { std::cout.write(static_cast<const char *>(p), (std::streamsize)bytes); }

static void put_matrix(NumericMatrix m) {
  SEXP s = m;
  const int64_t n = m.nrow();
  put(&n, 8);
  put(s->real.data(), s->real.size() * 8);
  // dimnames = list("1".."n", "1".."n")  (src/minHash.cpp:181-185)
  SEXP dn = s->attrs.at("dimnames");
  int64_t ok = dn->list.size() == 2;
  for (size_t a = 0; ok && a < 2; ++a) {
    ok = (int64_t)dn->list[a]->str.size() == n;
    for (int64_t i = 0; ok && i < n; ++i) ok = dn->list[a]->str[(size_t)i] == std::to_string(i + 1);
  }
  put(&ok, 8);
}

static void put_edges(List l) {
  SEXP s = l;
  SEXP thr = s->list.at(0), df = s->list.at(1);
  SEXP from = df->list.at(0), to = df->list.at(1), w = df->list.at(2);
  const int64_t m = (int64_t)from->integer.size();
  put(thr->real.data(), 8);
  put(&m, 8);
  put(from->integer.data(), (size_t)m * 4);
  put(to->integer.data(), (size_t)m * 4);
  put(w->real.data(), (size_t)m * 8);
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  std::vector<std::string> lines;
  for (std::string l; std::getline(std::cin, l);) lines.push_back(l);
  CharacterVector seqs((R_xlen_t)lines.size());
  for (size_t i = 0; i < lines.size(); ++i) seqs[(R_xlen_t)i] = lines[i];
  if (const char *e = std::getenv("GLUE_OPTION_SEED")) {       // options(DynaAlign.seed = <double>)
    SEXP s = stub::make(SEXPREC::REAL);
    s->real.push_back(std::strtod(e, nullptr));
    stub::options()["DynaAlign.seed"] = s;
  }
  if (const char *e = std::getenv("GLUE_OPTION_DEVICES")) {    // options(DynaAlign.devices = c(..))
    SEXP s = stub::make(SEXPREC::INT);
    for (const char *p = e; *p;) {
      char *end = nullptr;
      const long d = std::strtol(p, &end, 10);
      if (end == p) break;
      s->integer.push_back((int)d);
      p = (*end == ',') ? end + 1 : end;
    }
    stub::options()["DynaAlign.devices"] = s;
  }
  if (const char *e = std::getenv("GLUE_OPTION_EXCHANGE")) {
    SEXP s = stub::make(SEXPREC::STR);
    s->str.push_back(e);
    stub::options()["DynaAlign.exchange"] = s;
  }
  const std::string f = argv[1];
  try {
    if (f == "mh" && argc == 4) put_matrix(similarityMH(seqs, std::atoi(argv[2]), std::atoi(argv[3])));
    else if (f == "nw" && argc == 5) put_matrix(similarityNW(seqs, argv[2], std::atoi(argv[3]), std::atoi(argv[4])));
    else if (f == "mh_edges" && argc == 5) put_edges(similarityMH_edges(seqs, std::atoi(argv[2]), std::atoi(argv[3]), std::atof(argv[4])));
    else if (f == "nw_edges" && argc == 6)
      put_edges(similarityNW_edges(seqs, argv[2], std::atoi(argv[3]), std::atoi(argv[4]), std::atof(argv[5])));
    else return 2;
  } catch (const Rcpp::exception &e) {
    std::cerr << "ERROR: " << e.what() << std::endl;
    return 3;
  }
  std::cout.flush();
  return 0;
}
