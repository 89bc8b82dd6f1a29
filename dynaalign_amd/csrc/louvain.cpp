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
#include <condition_variable>
#include <cstdint>
#include <mutex>
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

// Host threads for the data-parallel parts (CSR build, strengths, aggregation) and for the speculative neighbourhood scans of
// the local-moving passes.  Nothing the function returns depends on the count (DYNAALIGN_LOUVAIN_THREADS overrides it; 1 = serial).
int host_threads(int64_t work_items) {
  int t = (int)std::min<int64_t>(16, std::max<int64_t>(1, work_items / 250000));   // (a thread per 250k adjacency entries, at most 16)
  if (config().louvain_threads > 0) t = config().louvain_threads;   // DYNAALIGN_LOUVAIN_THREADS
  const unsigned hw = std::thread::hardware_concurrency();
  if (hw > 0 && (unsigned)t > hw) t = (int)hw;
  return t;
}
template <typename F> void run_threads(int nthreads, F body) {   // body(thread index)
  if (nthreads <= 1) { body(0); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; ++t) th.emplace_back(body, t);
  for (auto &x : th) x.join();
}
// vertex range [v0, v1) of thread t when the vertices are cut into ranges of roughly equal adjacency volume
void volume_range(const std::vector<int64_t> &ptr, int64_t n, int t, int nthreads, int64_t *v0, int64_t *v1) {
  const int64_t total = ptr[(size_t)n];
  const int64_t lo = total * t / nthreads, hi = total * (t + 1) / nthreads;
  *v0 = t == 0 ? 0 : std::lower_bound(ptr.begin(), ptr.end() - 1, lo) - ptr.begin();
  *v1 = (t + 1 == nthreads) ? n : std::lower_bound(ptr.begin(), ptr.end() - 1, hi) - ptr.begin();
}

// std::vector::resize value-initialises: zeroing the 3.5 GB of a 2.9e8-entry adjacency on one thread costs 0.3 s before the
// threads that fill it even start (and places every page on the resizing thread's NUMA node).  The big arrays therefore use an
// allocator whose default construction does nothing; every element is written before it is read.
template <typename T> struct NoInitAlloc : std::allocator<T> {
  template <typename U> struct rebind { using other = NoInitAlloc<U>; };
  NoInitAlloc() = default;
  template <typename U> NoInitAlloc(const NoInitAlloc<U> &) {}
  template <typename U, typename... A> void construct(U *p, A &&...a) {
    if constexpr (sizeof...(A) == 0) ::new (static_cast<void *>(p)) U;
    else ::new (static_cast<void *>(p)) U(std::forward<A>(a)...);
  }
};

// symmetric CSR without self-loops; self-loop weights kept per vertex
struct Graph {
  int32_t n = 0;
  std::vector<int64_t> ptr;     // n + 1
  std::vector<int32_t, NoInitAlloc<int32_t>> adj;   // neighbour ids, ascending per vertex
  std::vector<double, NoInitAlloc<double>> w;       // edge weights
  std::vector<double> loop;     // self-loop weight per vertex (0 if none)
  double total = 0.0;           // 2m = sum of strengths
};

// k[v] = 2 loop(v) + sum of v's edge weights in canonical order; vertices are independent, a few threads share them
void strengths(const Graph &g, std::vector<double> &k) {
  k.assign((size_t)g.n, 0.0);
  const int nthreads = host_threads((int64_t)g.adj.size());
  run_threads(nthreads, [&](int t) {
    int64_t v0, v1;
    volume_range(g.ptr, g.n, t, nthreads, &v0, &v1);
    for (int64_t v = v0; v < v1; ++v) {
      double s = 2.0 * g.loop[(size_t)v];
      for (int64_t q = g.ptr[(size_t)v]; q < g.ptr[(size_t)v + 1]; ++q) s += g.w[(size_t)q];
      k[(size_t)v] = s;
    }
  });
}

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
  // loops: summed in ascending order of their position in a canonical (sorted) view -- a vertex's loops are
  // gathered first, sorted, then added, so the sum does not depend on arrival order
  std::vector<std::pair<int32_t, double>> loops;
  for (int64_t e = 0; e < m; ++e)
    if (ei[e] == ej[e]) loops.emplace_back(ei[e], ew[e]);
  std::sort(loops.begin(), loops.end());
  for (const auto &l : loops) g.loop[(size_t)l.first] += l.second;
  // fill + sort each neighbour list by (id, weight): thread t owns a contiguous vertex range of roughly equal adjacency volume,
  // scans the whole edge list and places the endpoints that fall in its range (random writes stay inside the thread's slice),
  // then sorts its lists -- vertices are independent and the sort canonicalises the order, so the result does not depend on the
  // thread count.  Duplicate entries are merged afterwards, sequentially and only if there are any.
  {
    const int nthreads = host_threads((int64_t)g.adj.size());
    std::atomic<int> any_dup(0);
    std::vector<int64_t> fill(g.ptr.begin(), g.ptr.end() - 1);
    auto work = [&](int t) {
      int64_t v0, v1;
      volume_range(g.ptr, n, t, nthreads, &v0, &v1);
      for (int64_t e = 0; e < m; ++e) {
        const int32_t a = ei[e], b = ej[e];
        if (a == b) continue;
        if (a >= v0 && a < v1) { g.adj[(size_t)fill[(size_t)a]] = b; g.w[(size_t)fill[(size_t)a]++] = ew[e]; }
        if (b >= v0 && b < v1) { g.adj[(size_t)fill[(size_t)b]] = a; g.w[(size_t)fill[(size_t)b]++] = ew[e]; }
      }
      std::vector<std::pair<int32_t, double>> tmp;
      bool dup = false;
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
    run_threads(nthreads, work);
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
  std::vector<double> k;
  strengths(g, k);
  for (int64_t v = 0; v < n; ++v) g.total += k[(size_t)v];     // (sequential: the order of this sum is part of the result)
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
//
// The decisions are strictly sequential (vertex i sees every move made before it), but what dominates a pass on a dense graph
// -- scanning a vertex's adjacency and summing the edge weights per neighbouring community (2 900 entries per vertex at
// N = 100k, thresh_p = .8) -- only depends on the communities of the vertex's NEIGHBOURS.  After the first pass few vertices
// move (3 371, 2 064, ... of 100 000), so helper threads run those scans AHEAD of the deciding thread (a window of positions in
// the visiting order) and the deciding thread uses a scan iff no neighbour of the vertex has moved since the scan started;
// otherwise it scans itself.  Two ways to know: in a pass with few moves the decider stamps the neighbours of every vertex it
// moves with a move counter (2 900 writes per move, a compare per use); in a pass where most vertices move (the first) that
// would cost as much as the scans, so it keeps a log of the moved vertices instead, lets the helpers run only a few positions
// ahead and looks the handful of moves made since a scan up in the vertex's sorted adjacency (binary searches).  A scan is the same
// canonical-order sum whoever runs it, so the membership is bit-identical to the one-thread run whatever the timing; when most
// scans of a pass turn out stale (the first pass: everything moves) the helpers are parked for the rest of the pass.
struct Scan {                       // neighbouring communities of one vertex in order of first appearance + weight to each
  std::vector<int32_t> cid;
  std::vector<double> cw;
  uint64_t epoch = 0;               // move counter read before the first community lookup
};
struct ScanScratch {
  std::vector<double> wto;
  std::vector<char> seen;
  explicit ScanScratch(size_t n) : wto(n, 0.0), seen(n, 0) {}
};
inline void scan_vertex(const Graph &g, const int32_t *comm, int32_t v, ScanScratch &sc, Scan &out) {
  out.cid.clear();
  out.cw.clear();
  for (int64_t q = g.ptr[(size_t)v]; q < g.ptr[(size_t)v + 1]; ++q) {
    const int32_t c = __atomic_load_n(&comm[(size_t)g.adj[(size_t)q]], __ATOMIC_RELAXED);
    if (!sc.seen[(size_t)c]) { sc.seen[(size_t)c] = 1; out.cid.push_back(c); }
    sc.wto[(size_t)c] += g.w[(size_t)q];
  }
  out.cw.resize(out.cid.size());
  for (size_t t = 0; t < out.cid.size(); ++t) {
    const int32_t c = out.cid[t];
    out.cw[t] = sc.wto[(size_t)c];
    sc.wto[(size_t)c] = 0.0;
    sc.seen[(size_t)c] = 0;
  }
}

int32_t one_level(const Graph &g, double resolution, Mt19937 &rng, std::vector<int32_t> &comm, bool *moved) {
  const int32_t n = g.n;
  comm.resize((size_t)n);
  std::iota(comm.begin(), comm.end(), 0);
  // per community: tot = sum of member strengths, in = internal weight (every internal edge twice, loops twice) -- both
  // kept up to date move by move, so the modularity after a pass costs O(n) instead of another sweep over all edges
  std::vector<double> k, tot, in((size_t)n);
  strengths(g, k);
  tot = k;
  for (int32_t v = 0; v < n; ++v) in[(size_t)v] = 2.0 * g.loop[(size_t)v];
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
  *moved = false;
  const double m2 = g.total;
  if (m2 <= 0.0) return n;

  // ---- helpers (only worth it on big graphs)
  const bool debug = config().louvain_debug;
  const int nthreads = host_threads((int64_t)g.adj.size());
  const int nhelp = nthreads - 1;
  constexpr int32_t WINDOW = 512, WINDOW_MIN = 32, WINDOW_MIN_LOG = 4;             // positions of the visiting order the helpers may run ahead: the ring /
  std::atomic<int32_t> window(WINDOW_MIN);                    // what the decider currently allows (few moves -> far ahead; many -> close)
  std::vector<Scan> ring(nhelp > 0 ? (size_t)WINDOW : 1);
  std::vector<std::atomic<uint8_t>> state(nhelp > 0 ? (size_t)n : 0);   // per position: 0 free, 1 helper scanning, 2 helper done, 3 decider's own
  std::vector<uint64_t> stamp(nhelp > 0 ? (size_t)n : 0, 0);  // per vertex: move counter when a neighbour last moved (decider only; stamp mode)
  constexpr uint64_t MOVE_LOG = 4096;                         // the last MOVE_LOG moved vertices, move e at slot e % MOVE_LOG (decider only)
  std::vector<int32_t> move_log(nhelp > 0 ? (size_t)MOVE_LOG : 0, -1);
  // did a neighbour of v move after move number `since`?  (moves since + 1 .. now; adjacency lists are sorted by neighbour id)
  auto neighbour_moved = [&](int32_t v, uint64_t since, uint64_t now) {
    if (now - since > MOVE_LOG) return true;                  // the log no longer reaches back that far
    const int32_t *b = g.adj.data() + g.ptr[(size_t)v], *e = g.adj.data() + g.ptr[(size_t)v + 1];
    for (uint64_t m = since + 1; m <= now; ++m)
      if (std::binary_search(b, e, move_log[(size_t)(m % MOVE_LOG)])) return true;
    return false;
  };
  std::atomic<uint64_t> epoch(0);                             // moves so far
  std::atomic<int32_t> decided(0), next_scan(0);              // positions finished by the decider / handed to helpers
  std::atomic<int> phase(0);                                  // 0 parked, 1 scanning, 2 quit
  std::atomic<int> pass_id(0), active(0);                     // pass counter; helpers inside their scanning loop
  std::mutex park_m;                                          // parked helpers SLEEP on park_cv (they used to spin on yield() and took cores from the
  std::condition_variable park_cv;                            // decider for the rest of a dense pass); phase / pass_id change under park_m when they wake helpers
  std::atomic<uint64_t> dbg_ns(0), dbg_cnt(0);                // (debug: time inside the helpers' scans)
  int32_t *comm_p = comm.data();
  std::vector<std::thread> helpers;
  for (int h = 0; h < nhelp; ++h)
    helpers.emplace_back([&]() {
      ScanScratch sc((size_t)n);
      int done_pass = 0;
      for (;;) {
        int ph;
        {
          std::unique_lock<std::mutex> lk(park_m);
          park_cv.wait(lk, [&] {
            ph = phase.load(std::memory_order_acquire);
            return ph == 2 || (ph == 1 && pass_id.load(std::memory_order_acquire) != done_pass);
          });
        }
        if (ph == 2) return;
        active.fetch_add(1, std::memory_order_acq_rel);
        const int cur = pass_id.load(std::memory_order_acquire);
        while (phase.load(std::memory_order_acquire) == 1 && pass_id.load(std::memory_order_acquire) == cur) {
          const int32_t i = next_scan.fetch_add(1, std::memory_order_relaxed);
          if (i >= n) break;
          bool open = true;                                   // the ring slot of position i is free once position i - WINDOW is decided
          while (decided.load(std::memory_order_acquire) + window.load(std::memory_order_relaxed) <= i) {
            if (phase.load(std::memory_order_acquire) != 1) { open = false; break; }
            std::this_thread::yield();
          }
          if (!open) break;                                   // parked mid-pass: position i stays free, the decider takes it
          uint8_t expect = 0;
          if (!state[(size_t)i].compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) continue;   // the decider took it
          Scan &slot = ring[(size_t)(i % WINDOW)];
          slot.epoch = epoch.load(std::memory_order_acquire);
          if (!debug) scan_vertex(g, comm_p, order[(size_t)i], sc, slot);
          else {
            const auto t0 = std::chrono::steady_clock::now();
            scan_vertex(g, comm_p, order[(size_t)i], sc, slot);
            dbg_ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
            dbg_cnt.fetch_add(1);
          }
          state[(size_t)i].store(2, std::memory_order_release);
        }
        done_pass = cur;
        active.fetch_sub(1, std::memory_order_acq_rel);
      }
    });

  ScanScratch my_sc((size_t)n);
  Scan my_scan;
  double q_prev = modularity_now();
  int64_t prev_changed = n;                                     // the first pass moves (nearly) everything
  for (int pass = 0;; ++pass) {
    // many moves expected (the first pass of a level; a later pass after one -- other than the first -- that still moved half the
    // vertices): move log + close window; few (typically everything after the first pass): stamps + far window.  A wrong guess only
    // costs speed: stale scans are recomputed, mostly-stale passes park the helpers.
    const bool log_mode = pass == 0 || (pass >= 2 && prev_changed * 2 > (int64_t)n);
    const int32_t wmin = log_mode ? WINDOW_MIN_LOG : WINDOW_MIN;
    int64_t changed = 0, used = 0, stale = 0, used_blk = 0, stale_blk = 0;
    uint64_t looked_up_blk = 0;                                 // moves looked up by the validity checks of this block
    const auto t_pass = std::chrono::steady_clock::now();
    bool use_scans = nhelp > 0;
    if (nhelp > 0) {                                          // open the pass for the helpers (none of them is scanning any more)
      while (active.load(std::memory_order_acquire) != 0) std::this_thread::yield();
      for (int32_t i = 0; i < n; ++i) state[(size_t)i].store(0, std::memory_order_relaxed);
      decided.store(0, std::memory_order_relaxed);
      next_scan.store(0, std::memory_order_relaxed);
      window.store(wmin, std::memory_order_relaxed);
      {
        std::lock_guard<std::mutex> lk(park_m);                 // (the resets above are ordered before the helpers' reads by this release + their acquire)
        pass_id.fetch_add(1, std::memory_order_acq_rel);
        phase.store(1, std::memory_order_release);
      }
      park_cv.notify_all();
    }
    for (int32_t idx = 0; idx < n; ++idx) {
      const int32_t v = order[(size_t)idx];
      const int32_t old = comm[(size_t)v];
      const Scan *sn = nullptr;
      if (nhelp > 0) {
        uint8_t expect = 0;
        if (!state[(size_t)idx].compare_exchange_strong(expect, 3, std::memory_order_acq_rel)) {
          while (state[(size_t)idx].load(std::memory_order_acquire) != 2) std::this_thread::yield();    // a helper is finishing it
          const Scan &slot = ring[(size_t)(idx % WINDOW)];
          const uint64_t now = epoch.load(std::memory_order_relaxed);
          looked_up_blk += now - slot.epoch;
          if (use_scans && (log_mode ? !neighbour_moved(v, slot.epoch, now) : stamp[(size_t)v] <= slot.epoch)) { sn = &slot; ++used; ++used_blk; }
          else { ++stale; ++stale_blk; }
        }
      }
      if (!sn) { scan_vertex(g, comm_p, v, my_sc, my_scan); sn = &my_scan; }
      double w_old = 0.0;
      for (size_t t = 0; t < sn->cid.size(); ++t)
        if (sn->cid[t] == old) { w_old = sn->cw[t]; break; }
      tot[(size_t)old] -= k[(size_t)v];                         // take v out of its community
      const double kv = k[(size_t)v];
      int32_t best = old;
      double w_best = w_old;
      double best_gain = w_old - resolution * tot[(size_t)old] * kv / m2;   // staying put
      for (size_t t = 0; t < sn->cid.size(); ++t) {
        const int32_t c = sn->cid[t];
        if (c == old) continue;
        const double gain = sn->cw[t] - resolution * tot[(size_t)c] * kv / m2;
        if (gain > best_gain) { best_gain = gain; best = c; w_best = sn->cw[t]; }
      }
      tot[(size_t)best] += kv;
      if (best != old) {
        ++changed;
        const double lv = 2.0 * g.loop[(size_t)v];
        in[(size_t)old] -= 2.0 * w_old + lv;
        in[(size_t)best] += 2.0 * w_best + lv;
        __atomic_store_n(&comm_p[(size_t)v], best, __ATOMIC_RELAXED);
        const uint64_t e = epoch.load(std::memory_order_relaxed) + 1;
        if (nhelp > 0) {                                         // scans of v's neighbours that started before move e are stale
          if (log_mode) move_log[(size_t)(e % MOVE_LOG)] = v;
          else if (use_scans)
            for (int64_t q = g.ptr[(size_t)v]; q < g.ptr[(size_t)v + 1]; ++q) stamp[(size_t)g.adj[(size_t)q]] = e;
        }
        epoch.store(e, std::memory_order_release);
      }
      if (nhelp > 0) {
        decided.store(idx + 1, std::memory_order_release);
        if (use_scans && (idx & 1023) == 1023) {                   // every 1024 positions: how many scans were stale?
          const int32_t w = window.load(std::memory_order_relaxed);
          if (2 * stale_blk > used_blk + stale_blk) {              // most: stay closer behind the decider, or -- already close (the
            if (w > wmin) window.store(std::max(wmin, w / 4), std::memory_order_relaxed);   // first pass: everything
            else { phase.store(0, std::memory_order_release); use_scans = false; }   // moves) -- park the helpers
          } else if (log_mode && looked_up_blk > 3 * 1024 && w > wmin) {   // every look-up is a binary search in a cold adjacency list: keep
            window.store(std::max(wmin, w / 2), std::memory_order_relaxed);   // the moves made between a scan and its use to ~2
          } else if (10 * stale_blk < used_blk + stale_blk && (!log_mode || looked_up_blk < 1024) && w < WINDOW)
            window.store(w * 2, std::memory_order_relaxed);
          used_blk = stale_blk = 0;
          looked_up_blk = 0;
        }
      }
    }
    if (nhelp > 0) phase.store(0, std::memory_order_release);
    if (debug)
      fprintf(stderr, "[louvain] n=%d pass: %lld moved, %.3f s (%d helper threads: %lld scans used, %lld stale)\n", n, (long long)changed,
              std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pass).count(), nhelp, (long long)used, (long long)stale);
    if (debug && nhelp > 0) fprintf(stderr, "   helper scans so far %llu, avg %.2f us\n", (unsigned long long)dbg_cnt.load(), dbg_cnt.load() ? dbg_ns.load() / 1e3 / dbg_cnt.load() : 0.0);
    prev_changed = changed;
    if (changed == 0) break;
    *moved = true;
    const double q_now = modularity_now();
    if (!(q_now > q_prev)) break;                               // igraph: keep passing only while modularity improves
    q_prev = q_now;
  }
  {
    std::lock_guard<std::mutex> lk(park_m);
    phase.store(2, std::memory_order_release);
  }
  park_cv.notify_all();
  for (auto &h : helpers) h.join();
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
  // one community at a time, handed out by an atomic counter to a few threads (each community's sums run over its members in
  // ascending vertex order and their canonical adjacency, whoever computes them); the lists are concatenated in community order
  std::vector<std::vector<int32_t>> cadj((size_t)nc);
  std::vector<std::vector<double>> cwt((size_t)nc);
  const int nthreads = host_threads((int64_t)g.adj.size());
  std::atomic<int32_t> next_c(0);
  run_threads(nthreads, [&](int) {
    std::vector<double> acc((size_t)nc, 0.0);
    std::vector<char> seen((size_t)nc, 0);
    std::vector<int32_t> touched;
    for (;;) {
      const int32_t c = next_c.fetch_add(1, std::memory_order_relaxed);
      if (c >= nc) break;
      touched.clear();
      double inner = 0.0, loops = 0.0;                         // inner: sum over ordered pairs inside c (each edge twice)
      for (int64_t t = mptr[(size_t)c]; t < mptr[(size_t)c + 1]; ++t) {
        const int32_t v = members[(size_t)t];
        loops += g.loop[(size_t)v];
        for (int64_t q = g.ptr[(size_t)v]; q < g.ptr[(size_t)v + 1]; ++q) {
          const int32_t d = comm[(size_t)g.adj[(size_t)q]];
          if (d == c) { inner += g.w[(size_t)q]; continue; }
          if (!seen[(size_t)d]) { seen[(size_t)d] = 1; touched.push_back(d); }
          acc[(size_t)d] += g.w[(size_t)q];
        }
      }
      out.loop[(size_t)c] = loops + 0.5 * inner;               // an undirected edge inside c = a loop of that weight
      std::sort(touched.begin(), touched.end());
      cadj[(size_t)c].assign(touched.begin(), touched.end());
      cwt[(size_t)c].resize(touched.size());
      for (size_t t = 0; t < touched.size(); ++t) {
        const int32_t d = touched[t];
        cwt[(size_t)c][t] = acc[(size_t)d];
        acc[(size_t)d] = 0.0;
        seen[(size_t)d] = 0;
      }
    }
  });
  for (int32_t c = 0; c < nc; ++c) {
    out.adj.insert(out.adj.end(), cadj[(size_t)c].begin(), cadj[(size_t)c].end());
    out.w.insert(out.w.end(), cwt[(size_t)c].begin(), cwt[(size_t)c].end());
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

// the multilevel loop on a canonical graph (shared by the two entry points)
static int run_levels(const Graph &g0, int64_t n_vertices, double resolution, uint32_t seed, int32_t *membership_out,
                      double *modularity_out, int32_t *levels_out) {
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

// The same clustering from a graph that is ALREADY canonical: symmetric CSR (both directions of every edge, columns ascending,
// no duplicates, no diagonal entries), weights given as codes into a value table, loops per vertex (0xFFFF = none) -- what
// da_dev_edges_to_csr produces on the device from the thresholded similarity (no host-side sort, no 16-byte-per-edge list).
// Identical result to da_louvain on the corresponding edge list: the graph is the one build_graph would build, entry for entry.
extern "C" int da_louvain_csr(int64_t n_vertices, const int64_t *ptr, const int32_t *adj, const uint16_t *codes, const uint16_t *loop_codes,
                              const double *values, int32_t n_values, double resolution, uint32_t seed, int32_t *membership_out,
                              double *modularity_out, int32_t *levels_out) {
  if (n_vertices < 0 || n_vertices > 0x7fffffffLL) return fail(DA_ERR_BAD_ARG, "bad vertex count");
  if (n_vertices > 0 && (!membership_out || !ptr)) return fail(DA_ERR_BAD_ARG, "NULL pointer");
  if (!(resolution >= 0.0)) return fail(DA_ERR_BAD_ARG, "resolution must be >= 0");
  if (modularity_out) *modularity_out = 0.0;
  if (levels_out) *levels_out = 0;
  if (n_vertices == 0) return DA_OK;
  const int64_t nnz = ptr[n_vertices];
  if (ptr[0] != 0 || nnz < 0 || (nnz > 0 && (!adj || !codes)) || !values || n_values <= 0) return fail(DA_ERR_BAD_ARG, "bad CSR arguments");
  const auto t_start = std::chrono::steady_clock::now();
  Graph g0;
  g0.n = (int32_t)n_vertices;
  g0.ptr.assign(ptr, ptr + n_vertices + 1);
  g0.adj.resize((size_t)nnz);
  g0.w.resize((size_t)nnz);
  g0.loop.assign((size_t)n_vertices, 0.0);
  for (int64_t v = 0; v < n_vertices; ++v) {
    if (g0.ptr[(size_t)v + 1] < g0.ptr[(size_t)v]) return fail(DA_ERR_BAD_ARG, "CSR row pointers must not decrease");
    if (loop_codes && loop_codes[v] != 0xFFFFu) {
      if ((int32_t)loop_codes[v] >= n_values) return fail(DA_ERR_BAD_ARG, "loop code out of range");
      g0.loop[(size_t)v] = values[loop_codes[v]];
    }
  }
  std::atomic<int> bad(0);
  {
    const int nthreads = host_threads(nnz);
    run_threads(nthreads, [&](int t) {
      int64_t v0, v1;
      volume_range(g0.ptr, n_vertices, t, nthreads, &v0, &v1);
      for (int64_t v = v0; v < v1; ++v) {
        int32_t prev = -1;
        for (int64_t q = g0.ptr[(size_t)v]; q < g0.ptr[(size_t)v + 1]; ++q) {
          const int32_t a = adj[q];
          if (a < 0 || a >= n_vertices || a == v || a <= prev || (int32_t)codes[q] >= n_values) { bad.store(1); return; }
          prev = a;
          g0.adj[(size_t)q] = a;
          g0.w[(size_t)q] = values[codes[q]];
        }
      }
    });
  }
  if (bad.load()) return fail(DA_ERR_BAD_ARG, "CSR rows must hold ascending, distinct neighbours != the row, codes < n_values");
  g0.total = 0.0;
  std::vector<double> k;
  strengths(g0, k);
  for (int64_t v = 0; v < n_vertices; ++v) g0.total += k[(size_t)v];
  if (config().louvain_debug)
    fprintf(stderr, "[louvain] graph taken from CSR in %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
  return run_levels(g0, n_vertices, resolution, seed, membership_out, modularity_out, levels_out);
}

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
  if (config().louvain_debug)
    fprintf(stderr, "[louvain] graph built in %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
  return run_levels(g0, n_vertices, resolution, seed, membership_out, modularity_out, levels_out);
}
