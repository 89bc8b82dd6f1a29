// TEST-ONLY stand-in for <Rcpp.h>: exactly the surface r_glue/src/similarity_glue.cpp uses, small enough to read in a minute.
//
// Purpose: give the glue a COMPILER PASS (and a run against libdynaalign_hip.so) in an image without R -- a typo / ABI-drift
// guard for argument order and types against include/dynaalign.h.  It is NOT Rcpp, it pins nothing about the reference,
// and it is never used to build the reference's sources (those stay unbuildable here).  Only
// tests/test_r_glue.py includes it.  Semantics kept deliberately minimal: values live in never-freed heap records, a
// "SEXP" is a pointer to one, options() are a process-global map the driver fills.
#pragma once

#include <cstddef>
#include <cstdio>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

typedef std::ptrdiff_t R_xlen_t;

struct SEXPREC {
  enum Kind { NIL, REAL, INT, STR, LIST } kind = NIL;
  std::vector<double> real;
  std::vector<int> integer;
  std::vector<std::string> str;
  std::vector<SEXPREC *> list;
  std::vector<std::string> names;
  std::map<std::string, SEXPREC *> attrs;
  R_xlen_t nrow = 0, ncol = 0;
};
typedef SEXPREC *SEXP;

inline bool Rf_isNull(SEXP s) { return s == nullptr || s->kind == SEXPREC::NIL; }
inline double *REAL(SEXP s) { return s->real.data(); }
inline int *INTEGER(SEXP s) { return s->integer.data(); }

namespace Rcpp {

class exception : public std::runtime_error {
 public:
  explicit exception(const std::string &m) : std::runtime_error(m) {}
};

// Rcpp::stop(fmt, ...): printf-style here (the glue only uses "%s" with a C string, or no arguments)
template <typename... A>
[[noreturn]] inline void stop(const char *fmt, A... a) {
  char buf[1024];
  std::snprintf(buf, sizeof buf, fmt, a...);
  throw exception(buf);
}
[[noreturn]] inline void stop(const char *msg) { throw exception(msg); }

namespace stub {
inline std::map<std::string, SEXP> &options() {
  static std::map<std::string, SEXP> o;
  return o;
}
inline SEXP make(SEXPREC::Kind k) {
  SEXP s = new SEXPREC;
  s->kind = k;
  return s;
}
struct string_proxy {
  SEXP v;
  R_xlen_t i;
  string_proxy &operator=(const std::string &s) {
    v->str[(size_t)i] = s;
    return *this;
  }
  const std::string &get() const { return v->str[(size_t)i]; }
};
}  // namespace stub

class CharacterVector {
  SEXP v;

 public:
  CharacterVector() : v(stub::make(SEXPREC::STR)) {}
  explicit CharacterVector(R_xlen_t n) : v(stub::make(SEXPREC::STR)) { v->str.resize((size_t)n); }
  CharacterVector(SEXP s) : v(s) {}
  R_xlen_t length() const { return (R_xlen_t)v->str.size(); }
  stub::string_proxy operator[](R_xlen_t i) const { return stub::string_proxy{v, i}; }
  operator SEXP() const { return v; }
};

class IntegerVector {
  SEXP v;

 public:
  explicit IntegerVector(R_xlen_t n) : v(stub::make(SEXPREC::INT)) { v->integer.assign((size_t)n, 0); }
  IntegerVector(SEXP s) : v(s) {}
  int *begin() const { return v->integer.data(); }
  int *end() const { return v->integer.data() + v->integer.size(); }
  int &operator[](R_xlen_t i) const { return v->integer[(size_t)i]; }
  operator SEXP() const { return v; }
};

class NumericVector {
  SEXP v;

 public:
  explicit NumericVector(R_xlen_t n) : v(stub::make(SEXPREC::REAL)) { v->real.assign((size_t)n, 0.0); }
  double &operator[](R_xlen_t i) const { return v->real[(size_t)i]; }
  operator SEXP() const { return v; }
};

struct attr_proxy {
  SEXP v;
  std::string name;
  attr_proxy &operator=(SEXP x) {
    v->attrs[name] = x;
    return *this;
  }
};

class NumericMatrix {
  SEXP v;

 public:
  NumericMatrix(R_xlen_t r, R_xlen_t c) : v(stub::make(SEXPREC::REAL)) {   // zero-initialised like R's allocMatrix + fill
    v->real.assign((size_t)r * (size_t)c, 0.0);
    v->nrow = r;
    v->ncol = c;
  }
  R_xlen_t nrow() const { return v->nrow; }
  R_xlen_t ncol() const { return v->ncol; }
  attr_proxy attr(const char *name) { return attr_proxy{v, name}; }
  operator SEXP() const { return v; }
};

// _["name"] = value
struct named_object {
  std::string name;
  SEXP value;
};
struct name_tag {
  std::string name;
  named_object operator=(SEXP x) const { return named_object{name, x}; }
  named_object operator=(double x) const {
    SEXP s = stub::make(SEXPREC::REAL);
    s->real.push_back(x);
    return named_object{name, s};
  }
};
struct name_placeholder {
  name_tag operator[](const char *n) const { return name_tag{n}; }
};
static const name_placeholder _{};

class List {
  SEXP v;
  static void push(SEXP l, const named_object &o) {
    l->list.push_back(o.value);
    l->names.push_back(o.name);
  }
  static void push(SEXP l, SEXP x) {
    l->list.push_back(x);
    l->names.push_back("");
  }

 public:
  explicit List(SEXP s) : v(s) {}
  template <typename... A>
  static List create(const A &...a) {
    SEXP l = stub::make(SEXPREC::LIST);
    (void)std::initializer_list<int>{(push(l, a), 0)...};
    return List(l);
  }
  operator SEXP() const { return v; }
};

class DataFrame {
  SEXP v;

 public:
  explicit DataFrame(SEXP s) : v(s) {}
  template <typename... A>
  static DataFrame create(const A &...a) {
    return DataFrame((SEXP)List::create(a...));
  }
  operator SEXP() const { return v; }
};

class Function {
  std::string name;

 public:
  explicit Function(std::string n) : name(std::move(n)) {}
  SEXP operator()(const char *arg) const {   // only getOption(<name>) is ever called
    if (name != "getOption") stop("stub: only getOption is callable");
    auto it = stub::options().find(arg);
    return it == stub::options().end() ? nullptr : it->second;
  }
};

struct function_proxy {
  std::string name;
  operator Function() const { return Function(name); }
};

class Environment {
 public:
  static Environment base_env() { return Environment(); }
  function_proxy operator[](const char *n) const { return function_proxy{n}; }
};

template <typename T>
T as(SEXP s);
template <>
inline double as<double>(SEXP s) {
  if (s->kind == SEXPREC::INT) return (double)s->integer.at(0);
  return s->real.at(0);
}
template <>
inline std::string as<std::string>(SEXP s) { return s->str.at(0); }
template <>
inline IntegerVector as<IntegerVector>(SEXP s) { return IntegerVector(s); }
template <typename T>
inline T as(const stub::string_proxy &p) { return T(p.get()); }

}  // namespace Rcpp
