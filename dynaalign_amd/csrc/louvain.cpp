// louvain.cpp -- host-side multilevel (Louvain) community detection on a weighted edge list.
//
// Why it is here: the caller of the hot path, clusterbreak (reference R/clusterbreak.R:203-259), hands every
// thresholded similarity matrix to netcluster (:112-136), whose default cluster_func is
// igraph::cluster_louvain(resolution = 1.05) on graph_from_adjacency_matrix(mode = "upper", weighted = TRUE).
// igraph is host C code in the reference too; this file is its stand-in for the recursion driver
// (dynaalign_amd/clusterbreak.py), sized for the edge lists the fused threshold step emits at N = 100k
// (~1e8 edges; python graph libraries do not get there).  It is NOT part of the similarity hot path and
// runs on the host by design -- exactly where the reference runs it.
//
// Algorithm = the structure of igraph_community_multilevel (Blondel et al. 2008 with a resolution
// parameter): per level, visit the vertices in a shuffled order, move each to the neighbouring community
// with the largest modularity gain, repeat passes while something moves and modularity improves, then
// aggregate communities into vertices; stop when a level moves nothing.  Conventions kept from igraph:
// an undirected self-loop of weight w adds 2w to its vertex's strength and to its community's internal
// weight; gain(i -> C) = w(i, C) - resolution * tot(C) * k_i / (2m); ties keep the earlier candidate;
// the membership returned is the one of the last level (highest modularity), renumbered 1.. in order of
// first appearance by vertex.
//
// igraph's result depends on R's RNG stream, which cannot be reproduced here; what this implementation
// guarantees instead is DETERMINISM: the adjacency is canonicalised (symmetric CSR, neighbours sorted,
// duplicate entries summed in sorted order), the shuffle is an explicit Fisher-Yates on mt19937(seed),
// and all floating-point sums run in that canonical order -- the same graph and seed give the same
// membership whatever order the edges arrive in (the device appends edges in arrival order).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <thread>
#include <vector>

#include "da_common.hpp"

namespace da {
namespace {

struct Mt19937 {   // ISO C++ [rand.predef] mt19937, restated (std::shuffle / uniform_int_distribution are implementation-defined)
  uint32_t s[624];
  int pos = 624;
  explicit Mt19937(uint32_t seed) {
    s[0] = seed;
    for (uint32_t i = 1; i < 624; ++i) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + i;
  }
  uint32_t next() {
    if (pos == 624) {
      for (int i = 0; i < 624; ++i) {
        const uint32_t y = (s[i] & 0x80000000u) | (s[(i + 1) % 624] & 0x7fffffffu);
        s[i] = s[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      pos = 0;
    }
    uint32_t y = s[pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
};

// symmetric CSR without self-loops; self-loop weights kept per vertex
struct Graph {
  int32_t n = 0;
  std::vector<int64_t> ptr;     // n + 1
  std::vector<int32_t> adj;     // neighbour ids, ascending per vertex
  std::vector<double> w;        // edge weights
  std::vector<double> loop;     // self-loop weight per vertex (0 if none)
  double total = 0.0;           // 2m = sum of strengths
};

// (i, j, w) list -> canonical Graph.  Entries with i == j are loops; (i, j) and (j, i) are the same edge;
// repeated entries are summed.
int build_graph(int64_t n, int64_t m, const int32_t *ei, const int32_t *ej, const double *ew, Graph &g) {
  g.n = (int32_t)n;
  g.ptr.assign((size_t)n + 1, 0);
  g.loop.assign((size_t)n, 0.0);
  for (int64_t e = 0; e < m; ++e) {
    const int32_t a = ei[e], b = ej[e];
    if (a < 0 || b < 0 || a >= n || b >= n) return fail(DA_ERR_BAD_ARG, "edge %lld: vertex out of range", (long long)e);
    if (!(ew[e] == ew[e])) return fail(DA_ERR_BAD_ARG, "edge %lld: weight is NaN", (long long)e);
    if (a != b) { ++g.ptr[(size_t)a + 1]; ++g.ptr[(size_t)b + 1]; }
  }
  for (int64_t v = 0; v < n; ++v) g.ptr[(size_t)v + 1] += g.ptr[(size_t)v];
  g.adj.resize((size_t)g.ptr[(size_t)n]);
  g.w.resize((size_t)g.ptr[(size_t)n]);
  std::vector<int64_t> fill(g.ptr.begin(), g.ptr.end() - 1);
  // loops: summed in ascending order of their position in a canonical (sorted) view -- a vertex's loops are
  // gathered first, sorted, then added, so the sum does not depend on arrival order
  std::vector<std::pair<int32_t, double>> loops;
  for (int64_t e = 0; e < m; ++e) {
    const int32_t a = ei[e], b = ej[e];
    if (a == b) { loops.emplace_back(a, ew[e]); continue; }
    g.adj[(size_t)fill[(size_t)a]] = b; g.w[(size_t)fill[(size_t)a]++] = ew[e];
    g.adj[(size_t)fill[(size_t)b]] = a; g.w[(size_t)fill[(size_t)b]++] = ew[e];
  }
  std::sort(loops.begin(), loops.end());
  for (const auto &l : loops) g.loop[(size_t)l.first] += l.second;
  // sort each neighbour list by (id, weight) -- vertices are independent, so a few host threads share them (the result does
  // not depend on the thread count) -- then merge duplicate entries, sequentially and only if there are any
  {
    int nthreads = (int)std::min<int64_t>(16, std::max<int64_t>(1, (int64_t)g.adj.size() / 2000000));
    if (const char *e = getenv("DYNAALIGN_LOUVAIN_THREADS")) nthreads = std::max(1, atoi(e));
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw > 0 && (unsigned)nthreads > hw) nthreads = (int)hw;
    std::atomic<int> any_dup(0);
    auto work = [&](int t) {
      std::vector<std::pair<int32_t, double>> tmp;
      bool dup = false;
      // contiguous ranges of roughly equal adjacency volume
      const int64_t total_adj = (int64_t)g.adj.size();
      const int64_t lo = total_adj * t / nthreads, hi = total_adj * (t + 1) / nthreads;
      int64_t v0 = std::lower_bound(g.ptr.begin(), g.ptr.end() - 1, lo) - g.ptr.begin();
      int64_t v1 = (t + 1 == nthreads) ? n : std::lower_bound(g.ptr.begin(), g.ptr.end() - 1, hi) - g.ptr.begin();
      for (int64_t v = v0; v < v1; ++v) {
        const int64_t b = g.ptr[(size_t)v], e = g.ptr[(size_t)v + 1];
        tmp.clear();
        for (int64_t k = b; k < e; ++k) tmp.emplace_back(g.adj[(size_t)k], g.w[(size_t)k]);
        std::sort(tmp.begin(), tmp.end());
        for (int64_t k = b; k < e; ++k) {
          g.adj[(size_t)k] = tmp[(size_t)(k - b)].first;
          g.w[(size_t)k] = tmp[(size_t)(k - b)].second;
          if (k > b && g.adj[(size_t)k] == g.adj[(size_t)k - 1]) dup = true;
        }
      }
      if (dup) any_dup.store(1);
    };
    if (nthreads <= 1) work(0);
    else {
      std::vector<std::thread> th;
      for (int t = 0; t < nthreads; ++t) th.emplace_back(work, t);
      for (auto &x : th) x.join();
    }
    if (any_dup.load()) {
      int64_t out = 0;
      std::vector<int64_t> nptr((size_t)n + 1, 0);
      for (int64_t v = 0; v < n; ++v) {
        const int64_t b = g.ptr[(size_t)v], e = g.ptr[(size_t)v + 1];
        nptr[(size_t)v] = out;
        for (int64_t k = b; k < e; ++k) {
          if (out > nptr[(size_t)v] && g.adj[(size_t)out - 1] == g.adj[(size_t)k]) g.w[(size_t)out - 1] += g.w[(size_t)k];
          else { g.adj[(size_t)out] = g.adj[(size_t)k]; g.w[(size_t)out] = g.w[(size_t)k]; ++out; }
        }
      }
      nptr[(size_t)n] = out;
      g.ptr.swap(nptr);
      g.adj.resize((size_t)out);
      g.w.resize((size_t)out);
    }
  }
  g.total = 0.0;
  for (int64_t v = 0; v < n; ++v) {
    double k = 2.0 * g.loop[(size_t)v];
    for (int64_t q = g.ptr[(size_t)v]; q < g.ptr[(size_t)v + 1]; ++q) k += g.w[(size_t)q];
    g.total += k;
  }
  return DA_OK;
}

double modularity_of(const Graph &g, const std::vector<int32_t> &comm, double resolution) {
  if (g.total <= 0.0) return 0.0;
  int32_t nc = 0;
  for (int32_t c : comm) nc = std::max(nc, c + 1);
  std::vector<double> in((size_t)nc, 0.0), tot((size_t)nc, 0.0);
  for (int32_t v = 0; v < g.n; ++v) {
    const int32_t c = comm[(size_t)v];
    double k = 2.0 * g.loop[(size_t)v];
    in[(size_t)c] += 2.0 * g.loop[(size_t)v];
    for (int64_t q = g.ptr[(size_t)v]; q < g.ptr[(size_t)v + 1]; ++q) {
      k += g.w[(size_t)q];
      if (comm[(size_t)g.adj[(size_t)q]] == c) in[(size_t)c] += g.w[(size_t)q];
    }
    tot[(size_t)c] += k;
  }
  double Q = 0.0;
  for (int32_t c = 0; c < nc; ++c) Q += in[(size_t)c] / g.total - resolution * (tot[(size_t)c] / g.total) * (tot[(size_t)c] / g.total);
  return Q;
}

// One level: local moving on g.  comm (out): community per vertex, renumbered 0..nc-1 in order of first
// appearance by vertex.  Returns the number of communities; *moved says whether anything changed.
int32_t one_level(const Graph &g, double resolution, Mt19937 &rng, std::vector<int32_t> &comm, bool *moved) {
  const int32_t n = g.n;
  comm.resize((size_t)n);
  std::iota(comm.begin(), comm.end(), 0);
  // per community: tot = sum of member strengths, in = internal weight (every internal edge twice, loops twice) -- both
  // kept up to date move by move, so the modularity after a pass costs O(n) instead of another sweep over all edges
  std::vector<double> k((size_t)n), tot((size_t)n), in((size_t)n);
  for (int32_t v = 0; v < n; ++v) {
    double s = 2.0 * g.loop[(size_t)v];
    for (int64_t q = g.ptr[(size_t)v]; q < g.ptr[(size_t)v + 1]; ++q) s += g.w[(size_t)q];
    k[(size_t)v] = tot[(size_t)v] = s;
    in[(size_t)v] = 2.0 * g.loop[(size_t)v];
  }
  auto modularity_now = [&]() {
    double Q = 0.0;
    for (int32_t c = 0; c < n; ++c)
      if (tot[(size_t)c] != 0.0 || in[(size_t)c] != 0.0)
        Q += in[(size_t)c] / g.total - resolution * (tot[(size_t)c] / g.total) * (tot[(size_t)c] / g.total);
    return Q;
  };
  std::vector<int32_t> order((size_t)n);
  std::iota(order.begin(), order.end(), 0);
  for (int32_t i = 0; i + 1 < n; ++i) {                       // Fisher-Yates on raw mt19937 draws
    const int32_t j = i + (int32_t)(rng.next() % (uint32_t)(n - i));
    std::swap(order[(size_t)i], order[(size_t)j]);
  }
  std::vector<double> wto((size_t)n, 0.0);                    // weight from the current vertex to community c
  std::vector<char> seen((size_t)n, 0);
  std::vector<int32_t> touched;                               // neighbouring communities in order of first appearance
  *moved = false;
  const double m2 = g.total;
  if (m2 <= 0.0) return n;
  double q_prev = modularity_now();
  for (;;) {
    int64_t changed = 0;
    for (int32_t idx = 0; idx < n; ++idx) {
      const int32_t v = order[(size_t)idx];
      const int32_t old = comm[(size_t)v];
      touched.clear();
      for (int64_t q = g.ptr[(size_t)v]; q < g.ptr[(size_t)v + 1]; ++q) {
        const int32_t c = comm[(size_t)g.adj[(size_t)q]];
        if (!seen[(size_t)c]) { seen[(size_t)c] = 1; touched.push_back(c); }
        wto[(size_t)c] += g.w[(size_t)q];
      }
      tot[(size_t)old] -= k[(size_t)v];                         // take v out of its community
      const double kv = k[(size_t)v];
      int32_t best = old;
      double best_gain = wto[(size_t)old] - resolution * tot[(size_t)old] * kv / m2;   // staying put
      for (int32_t c : touched) {
        if (c == old) continue;
        const double gain = wto[(size_t)c] - resolution * tot[(size_t)c] * kv / m2;
        if (gain > best_gain) { best_gain = gain; best = c; }
      }
      tot[(size_t)best] += kv;
      comm[(size_t)v] = best;
      if (best != old) {
        ++changed;
        const double lv = 2.0 * g.loop[(size_t)v];
        in[(size_t)old] -= 2.0 * wto[(size_t)old] + lv;
        in[(size_t)best] += 2.0 * wto[(size_t)best] + lv;
      }
      for (int32_t c : touched) { wto[(size_t)c] = 0.0; seen[(size_t)c] = 0; }
    }
    if (getenv("DYNAALIGN_LOUVAIN_DEBUG")) fprintf(stderr, "[louvain] n=%d pass: %lld moved\n", n, (long long)changed);
    if (changed == 0) break;
    *moved = true;
    const double q_now = modularity_now();
    if (!(q_now > q_prev)) break;                               // igraph: keep passing only while modularity improves
    q_prev = q_now;
  }
  // renumber in order of first appearance by vertex
  std::vector<int32_t> newid((size_t)n, -1);
  int32_t nc = 0;
  for (int32_t v = 0; v < n; ++v) {
    int32_t &id = newid[(size_t)comm[(size_t)v]];
    if (id < 0) id = nc++;
    comm[(size_t)v] = id;
  }
  return nc;
}

// communities -> vertices of the next level; intra-community weight becomes a self-loop
void aggregate(const Graph &g, const std::vector<int32_t> &comm, int32_t nc, Graph &out) {
  out.n = nc;
  out.loop.assign((size_t)nc, 0.0);
  out.ptr.assign((size_t)nc + 1, 0);
  out.adj.clear();
  out.w.clear();
  // members of each community, ascending vertex id
  std::vector<int64_t> mptr((size_t)nc + 1, 0);
  for (int32_t v = 0; v < g.n; ++v) ++mptr[(size_t)comm[(size_t)v] + 1];
  for (int32_t c = 0; c < nc; ++c) mptr[(size_t)c + 1] += mptr[(size_t)c];
  std::vector<int32_t> members((size_t)g.n);
  {
    std::vector<int64_t> f(mptr.begin(), mptr.end() - 1);
    for (int32_t v = 0; v < g.n; ++v) members[(size_t)f[(size_t)comm[(size_t)v]]++] = v;
  }
  std::vector<double> acc((size_t)nc, 0.0);
  std::vector<char> seen((size_t)nc, 0);
  std::vector<int32_t> touched;
  for (int32_t c = 0; c < nc; ++c) {
    touched.clear();
    double inner = 0.0;                                        // sum over ordered pairs inside c (each edge twice)
    for (int64_t t = mptr[(size_t)c]; t < mptr[(size_t)c + 1]; ++t) {
      const int32_t v = members[(size_t)t];
      out.loop[(size_t)c] += g.loop[(size_t)v];
      for (int64_t q = g.ptr[(size_t)v]; q < g.ptr[(size_t)v + 1]; ++q) {
        const int32_t d = comm[(size_t)g.adj[(size_t)q]];
        if (d == c) { inner += g.w[(size_t)q]; continue; }
        if (!seen[(size_t)d]) { seen[(size_t)d] = 1; touched.push_back(d); }
        acc[(size_t)d] += g.w[(size_t)q];
      }
    }
    out.loop[(size_t)c] += 0.5 * inner;                        // an undirected edge inside c = a loop of that weight
    std::sort(touched.begin(), touched.end());
    for (int32_t d : touched) {
      out.adj.push_back(d);
      out.w.push_back(acc[(size_t)d]);
      acc[(size_t)d] = 0.0;
      seen[(size_t)d] = 0;
    }
    out.ptr[(size_t)c + 1] = (int64_t)out.adj.size();
  }
  out.total = 0.0;
  for (int32_t v = 0; v < nc; ++v) {
    double k = 2.0 * out.loop[(size_t)v];
    for (int64_t q = out.ptr[(size_t)v]; q < out.ptr[(size_t)v + 1]; ++q) k += out.w[(size_t)q];
    out.total += k;
  }
}

}  // namespace
}  // namespace da

using namespace da;

extern "C" int da_louvain(int64_t n_vertices, int64_t n_edges, const int32_t *ei, const int32_t *ej, const double *ew,
                          double resolution, uint32_t seed, int32_t *membership_out, double *modularity_out,
                          int32_t *levels_out) {
  if (n_vertices < 0 || n_edges < 0 || n_vertices > 0x7fffffffLL) return fail(DA_ERR_BAD_ARG, "bad vertex / edge count");
  if (n_vertices > 0 && !membership_out) return fail(DA_ERR_BAD_ARG, "NULL membership buffer");
  if (n_edges > 0 && (!ei || !ej || !ew)) return fail(DA_ERR_BAD_ARG, "NULL edge arrays");
  if (!(resolution >= 0.0)) return fail(DA_ERR_BAD_ARG, "resolution must be >= 0");
  if (modularity_out) *modularity_out = 0.0;
  if (levels_out) *levels_out = 0;
  if (n_vertices == 0) return DA_OK;
  Graph g0;
  const auto t_start = std::chrono::steady_clock::now();
  int rc = build_graph(n_vertices, n_edges, ei, ej, ew, g0);
  if (rc != DA_OK) return rc;
  if (getenv("DYNAALIGN_LOUVAIN_DEBUG"))
    fprintf(stderr, "[louvain] graph built in %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
  Mt19937 rng(seed);
  std::vector<int32_t> member((size_t)n_vertices);             // community of every ORIGINAL vertex so far
  std::iota(member.begin(), member.end(), 0);
  Graph cur, next;
  const Graph *g = &g0;
  std::vector<int32_t> comm;
  int32_t levels = 0;
  for (;;) {
    bool moved = false;
    const int32_t nc = one_level(*g, resolution, rng, comm, &moved);
    if (!moved) break;
    ++levels;
    for (auto &c : member) c = comm[(size_t)c];
    if (nc == g->n) break;
    aggregate(*g, comm, nc, next);
    cur = std::move(next);
    g = &cur;
    next = Graph();
  }
  // 1-based ids in order of first appearance by vertex (igraph reindexes the same way; R adds 1)
  std::vector<int32_t> newid((size_t)n_vertices, -1);
  int32_t nc = 0;
  for (int64_t v = 0; v < n_vertices; ++v) {
    int32_t &id = newid[(size_t)member[(size_t)v]];
    if (id < 0) id = nc++;
    member[(size_t)v] = id;
  }
  if (modularity_out) *modularity_out = modularity_of(g0, member, resolution);
  if (levels_out) *levels_out = levels;
  for (int64_t v = 0; v < n_vertices; ++v) membership_out[v] = member[(size_t)v] + 1;
  return DA_OK;
}
