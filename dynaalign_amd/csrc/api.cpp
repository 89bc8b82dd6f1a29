// api.cpp -- the C ABI of libdynaalign_hip.so (see include/dynaalign.h).
//
// Host-side marshalling only: argument validation with the reference's error
// order and message texts, packing, H2D/D2H, and kernel launches.  There is
// deliberately NO CPU implementation of the hot path in this library.
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <condition_variable>
#include <cstddef>
#include <map>
#include <memory>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "da_common.hpp"

namespace da { size_t destroy_cached_comms(); }   // defined with the multi-device entry points

namespace da {

const signed char *matrix_table_host(int id);
const char *matrix_name_host(int id);
int matrix_count_host();

std::string &last_error_ref() {
  static thread_local std::string msg;
  return msg;
}

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error_ref() = buf;
  return code;
}

namespace {

// Large device buffers are kept for the next call instead of going back to the driver: hipMalloc / hipFree of the
// 80 GB result buffer cost up to 3.2 s every other call on MI355X (profiles/r02_b_host_path_trace.txt) -- more than the
// 1.5 s the PCIe copy of the result takes -- and clusterbreak calls sim_fn again and again.  Per device up to MAX_PARKED
// buffers of >= 1 MiB totalling at most 30 % of the device's memory are parked (one call of the duplicate-collapsing routes uses four:
// plan, table, gathered table, plane workspace); a request takes the smallest parked buffer that fits it and is at most
// twice its size (a 300 MB request must not walk away with the 9 GB buffer the next allocation of the same call wants);
// when room is needed the smallest parked buffers go first (cheapest to allocate again).  da_release_device_memory()
// returns everything to the driver, and so does an allocation of ours that would otherwise fail.
struct BigCache {
  static constexpr size_t MIN_BYTES = (size_t)1 << 20;     // (hipFree synchronises the device: even the 10 MB plan is worth keeping)
  static constexpr int MAX_PARKED = 16;
  struct Ent { int dev; void *p; size_t bytes; };
  std::mutex m;
  std::vector<Ent> parked;
  std::vector<size_t> budget;                                   // per device: 30 % of its memory (0 = not asked yet)
  size_t budget_of(int dev) {
    if ((size_t)dev >= budget.size()) budget.resize((size_t)dev + 1, 0);
    if (!budget[(size_t)dev]) {
      size_t total_b = 0;
      const int pct = da::config().buffer_cache_pct;         // DYNAALIGN_BUFFER_CACHE_PCT: share of the device's memory that may stay parked (30)
      budget[(size_t)dev] = hipDeviceTotalMem(&total_b, dev) == hipSuccess && total_b ? total_b / 100 * (size_t)pct : (size_t)64 << 30;
      if (!budget[(size_t)dev]) budget[(size_t)dev] = 1;       // 0 % = park nothing (but do not ask again)
    }
    return budget[(size_t)dev];
  }
  void *take(int dev, size_t bytes, size_t *cap) {
    std::lock_guard<std::mutex> g(m);
    size_t best = parked.size();
    for (size_t i = 0; i < parked.size(); ++i)
      if (parked[i].dev == dev && parked[i].bytes >= bytes && parked[i].bytes / 2 <= bytes &&
          (best == parked.size() || parked[i].bytes < parked[best].bytes)) best = i;
    if (best == parked.size()) return nullptr;
    void *p = parked[best].p;
    *cap = parked[best].bytes;
    parked.erase(parked.begin() + (long)best);
    // a buffer may have been parked by an early return with kernels still in flight on some stream: nothing may still be
    // touching it when its next owner -- possibly on another stream or thread -- starts writing (the device is idle here in
    // the normal case: the call that parked it ended with a synchronisation, so this costs microseconds)
    (void)hipDeviceSynchronize();
    return p;
  }
  void park(int dev, void *p, size_t bytes) {                   // (the caller has made `dev` current)
    std::lock_guard<std::mutex> g(m);
    const size_t cap_bytes = budget_of(dev);
    if (bytes > cap_bytes) { (void)hipFree(p); return; }
    for (;;) {
      int mine = 0;
      size_t sum = 0, smallest = parked.size();
      for (size_t i = 0; i < parked.size(); ++i)
        if (parked[i].dev == dev) {
          ++mine; sum += parked[i].bytes;
          if (smallest == parked.size() || parked[i].bytes < parked[smallest].bytes) smallest = i;
        }
      if (mine < MAX_PARKED && sum + bytes <= cap_bytes) break;
      if (parked[smallest].bytes >= bytes) { (void)hipFree(p); return; }   // everything parked is bigger: the newcomer goes
      (void)hipFree(parked[smallest].p);
      parked.erase(parked.begin() + (long)smallest);
    }
    parked.push_back({dev, p, bytes});
  }
  size_t release(int dev_or_all) {
    std::lock_guard<std::mutex> g(m);
    size_t freed = 0;
    for (size_t i = 0; i < parked.size();)
      if (dev_or_all < 0 || parked[i].dev == dev_or_all) { (void)hipFree(parked[i].p); freed += parked[i].bytes; parked.erase(parked.begin() + (long)i); }
      else ++i;
    return freed;
  }
};
BigCache &big_cache() { static BigCache *c = new BigCache; return *c; }   // never destroyed: the HIP runtime may be gone at exit

// RAII device buffer
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int dev = 0;
  ~DevBuf() {
    if (!p) return;
    if (cap >= BigCache::MIN_BYTES && !da::config().no_buffer_cache) big_cache().park(dev, p, cap);
    else (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    if (bytes == 0) bytes = 16;
    (void)hipGetDevice(&dev);
    if (bytes >= BigCache::MIN_BYTES && (p = big_cache().take(dev, bytes, &cap)) != nullptr) return DA_OK;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipErrorOutOfMemory && big_cache().release(dev) > 0) {   // parked buffers are the first thing to give back
      (void)hipGetLastError();
      e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) {
      p = nullptr;
      return fail(e == hipErrorOutOfMemory ? DA_ERR_NOMEM : DA_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", bytes,
                  hipGetErrorString(e));
    }
    cap = bytes;
    return DA_OK;
  }
  template <typename T> T *as() const { return static_cast<T *>(p); }
};

int require_device() {
  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess || cnt <= 0)
    return fail(DA_ERR_NO_DEVICE,
                "no usable HIP device (%s); libdynaalign_hip has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  return DA_OK;
}

// reference src/minHash.cpp:121-131, in that order
int validate_mh(int64_t n, int k, int n_hash) {
  if (n <= 0) return fail(DA_ERR_EMPTY_INPUT, "%s", da_status_message(DA_ERR_EMPTY_INPUT));
  if (k <= 0) return fail(DA_ERR_BAD_K, "%s", da_status_message(DA_ERR_BAD_K));
  if (n_hash <= 0) return fail(DA_ERR_BAD_NHASH, "%s", da_status_message(DA_ERR_BAD_NHASH));
  return DA_OK;
}

int64_t sig_ld_for(int n_hash) { return ((int64_t)n_hash + 31) / 32 * 32; }

int check_offsets(const int64_t *offsets, int64_t n, int64_t *total, int64_t *max_len) {
  if (!offsets) return fail(DA_ERR_BAD_ARG, "offsets is NULL");
  int64_t mx = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t len = offsets[i + 1] - offsets[i];
    if (len < 0) return fail(DA_ERR_BAD_ARG, "offsets must be non-decreasing (sequence %lld)", (long long)i);
    mx = std::max(mx, len);
  }
  if (offsets[0] != 0) return fail(DA_ERR_BAD_ARG, "offsets[0] must be 0");
  *total = offsets[n];
  *max_len = mx;
  return DA_OK;
}

// Upload the packed sequences (+ optionally seeds) to the current device.
struct DeviceInput {
  DevBuf res, off, seeds;
  int upload(const uint8_t *residues, const int64_t *offsets, int64_t n, int64_t total,
             const uint32_t *seedv, int n_hash) {
    int rc;
    if ((rc = res.alloc((size_t)total)) != DA_OK) return rc;
    if ((rc = off.alloc((size_t)(n + 1) * sizeof(int64_t))) != DA_OK) return rc;
    if (total) DA_HIP_TRY(hipMemcpy(res.p, residues, (size_t)total, hipMemcpyHostToDevice));
    DA_HIP_TRY(hipMemcpy(off.p, offsets, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    if (seedv) {
      if ((rc = seeds.alloc((size_t)n_hash * sizeof(uint32_t))) != DA_OK) return rc;
      DA_HIP_TRY(hipMemcpy(seeds.p, seedv, (size_t)n_hash * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    return DA_OK;
  }
};


// Device -> caller-owned pageable host memory (the R matrix).  A plain hipMemcpy of pageable memory
// staged at ~17 GB/s (measured, profiles/r01_d_host_path.json); this ring of pinned staging buffers
// keeps the DMA engine streaming while host threads copy finished chunks into the destination.
// (hipHostRegister of the destination would pin up to 80 GB of R's heap -- not ours to pin.)
// A few persistent host threads that copy pieces of a pinned chunk into the destination (the destination's
// first-touch page faults are what they parallelise; one pool per call, not one thread per chunk).
class CopyPool {
 public:
  explicit CopyPool(int workers) {
    for (int w = 0; w < workers; ++w) th_.emplace_back([this, w]() { run(w); });
  }
  ~CopyPool() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; ++gen_; }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  // fn(b, e) over [0, len) split into one piece per worker + the calling thread; returns when all pieces are done
  void run_split(size_t len, const std::function<void(size_t, size_t)> &fn) {
    const size_t parts = th_.size() + 1, part = (len + parts - 1) / parts;
    { std::lock_guard<std::mutex> g(m_); fn_ = &fn; len_ = len; part_ = part; pending_ = (int)th_.size(); ++gen_; }
    cv_.notify_all();
    fn(0, std::min(part, len));
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [this]() { return pending_ == 0; });
  }

 private:
  void run(int w) {
    uint64_t seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> g(m_);
      cv_.wait(g, [&]() { return gen_ != seen; });
      seen = gen_;
      if (stop_) return;
      const std::function<void(size_t, size_t)> *fn = fn_;
      const size_t len = len_, part = part_;
      g.unlock();
      const size_t b = (size_t)(w + 1) * part;
      if (b < len) (*fn)(b, std::min(len, b + part));
      g.lock();
      if (--pending_ == 0) done_.notify_one();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  uint64_t gen_ = 0;
  bool stop_ = false;
  const std::function<void(size_t, size_t)> *fn_ = nullptr;
  size_t len_ = 0, part_ = 0;
  int pending_ = 0;
};

// dst[i] = table[src[i]] with streaming stores (the destination is written once: no read-for-ownership traffic)
void widen_u16_to_f64(double *dst, const uint16_t *src, size_t b, size_t e, const double *table) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  size_t i = b;
  if (i < e && (reinterpret_cast<uintptr_t>(dst + i) & 15)) { dst[i] = table[src[i]]; ++i; }
  for (; i + 2 <= e; i += 2) {
    const v2d v = {table[src[i]], table[src[i + 1]]};
    __builtin_nontemporal_store(v, reinterpret_cast<v2d *>(dst + i));
  }
  if (i < e) dst[i] = table[src[i]];
}

// Device -> caller-owned pageable host memory through the pinned ring.  With `table` the device buffer holds uint16 codes and
// the HOST widens them while copying: dst (double) [i] = table[code[i]] -- a quarter of the bytes cross PCIe (the 80 GB float64
// result of N = 100k took 1.48 s at 54 GB/s; as codes it is 20 GB), and the divide behind every table entry is the same IEEE
// operation on the host as on the device (src/minHash.cpp:174, src/pairwiseSeqAlign.cpp:311).
int d2h_pipelined(void *dst, const void *d_src, size_t bytes, const double *table = nullptr) {
  constexpr size_t CHUNK = (size_t)64 << 20, SMALL = (size_t)8 << 20;   // SMALL: codes of n <= 2048 sequences
  constexpr int RING = 4, MAX_WORKERS = 32;
  int WORKERS = table ? 16 : 8;          // host threads per chunk (first-touch page faults of the destination parallelise)
  if (da::config().d2h_threads > 0) WORKERS = std::min(MAX_WORKERS, da::config().d2h_threads);   // DYNAALIGN_D2H_THREADS
  {
    const unsigned hw = std::thread::hardware_concurrency();
    cpu_set_t cs;
    int avail = (sched_getaffinity(0, sizeof cs, &cs) == 0) ? CPU_COUNT(&cs) : (int)hw;
    if (avail > 0 && WORKERS > avail) WORKERS = avail;
  }
  if (bytes <= CHUNK && !table) {
    DA_HIP_TRY(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return DA_OK;
  }
  if (bytes <= SMALL && table) {
    // small results (clusterbreak calls sim_fn on thousands of small subsets): ONE plain copy of the codes and an inline
    // widening -- no pinned ring (4 x 64 MiB of hipHostMalloc), no stream / events, no thread pool per call
    std::vector<uint16_t> codes(bytes / 2);
    DA_HIP_TRY(hipMemcpy(codes.data(), d_src, bytes, hipMemcpyDeviceToHost));
    widen_u16_to_f64(static_cast<double *>(dst), codes.data(), 0, bytes / 2, table);
    return DA_OK;
  }
  if (da::config().plain_d2h && !table) {
    DA_HIP_TRY(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return DA_OK;
  }
  struct Slot { void *pin = nullptr; hipEvent_t ev = nullptr; };
  Slot ring[RING];
  hipStream_t st = nullptr;
  int rc = DA_OK;
  auto cleanup = [&]() {                 // copies still in flight on `st` must not outlive the pinned slots they write
    if (st) (void)hipStreamSynchronize(st);
    for (auto &s : ring) { if (s.pin) (void)hipHostFree(s.pin); if (s.ev) (void)hipEventDestroy(s.ev); }
    if (st) (void)hipStreamDestroy(st);
  };
  auto check = [&](hipError_t e, const char *what) {
    if (e != hipSuccess && rc == DA_OK) rc = fail(DA_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
    return e == hipSuccess;
  };
  if (!check(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate")) { cleanup(); return rc; }
  for (auto &s : ring)
    if (!check(hipHostMalloc(&s.pin, CHUNK, hipHostMallocDefault), "hipHostMalloc") ||
        !check(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming), "hipEventCreate")) { cleanup(); return rc; }
  if (!check(hipDeviceSynchronize(), "hipDeviceSynchronize")) { cleanup(); return rc; }   // the producing kernels ran on the null stream
  {
    CopyPool pool(WORKERS - 1);
    const size_t nchunk = (bytes + CHUNK - 1) / CHUNK;
    for (size_t c = 0; c < nchunk + RING - 1 && rc == DA_OK; ++c) {
      if (c < nchunk) {                  // slot c % RING was drained RING iterations ago
        const size_t off = c * CHUNK, len = std::min(CHUNK, bytes - off);
        if (!check(hipMemcpyAsync(ring[c % RING].pin, static_cast<const char *>(d_src) + off, len, hipMemcpyDeviceToHost, st), "hipMemcpyAsync") ||
            !check(hipEventRecord(ring[c % RING].ev, st), "hipEventRecord")) break;
      }
      if (c >= RING - 1) {
        const size_t done = c - (RING - 1), off = done * CHUNK, len = std::min(CHUNK, bytes - off);
        if (!check(hipEventSynchronize(ring[done % RING].ev), "hipEventSynchronize")) break;
        const char *src = static_cast<const char *>(ring[done % RING].pin);
        if (table) {
          double *d = static_cast<double *>(dst) + off / 2;                 // `bytes` counts the uint16 source
          const uint16_t *s16 = reinterpret_cast<const uint16_t *>(src);
          pool.run_split(len / 2, [=](size_t b, size_t e) { widen_u16_to_f64(d, s16, b, e, table); });
        } else {
          char *d = static_cast<char *>(dst) + off;
          pool.run_split(len, [=](size_t b, size_t e) { memcpy(d + b, src + b, e - b); });
        }
      }
    }
  }
  cleanup();
  return rc;
}

// DYNAALIGN_TRACE=1: wall-clock checkpoints of the host-pointer paths on stderr (where does T_h go?)
struct Trace {
  const bool on = da::config().trace;
  const char *what;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
  explicit Trace(const char *w) : what(w) {}
  void mark(const char *step) {
    if (!on) return;
    (void)hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[dynaalign] %s: %-28s %9.3f ms (total %9.3f ms)\n", what, step,
            std::chrono::duration<double, std::milli>(now - last).count(), std::chrono::duration<double, std::milli>(now - t0).count());
    last = now;
  }
};

// How many result rows to keep on the device at once (host-pointer paths).
int64_t rows_per_block(int64_t n, size_t bytes_per_elem) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)8 << 30;
  {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(big_cache().m);       // what is parked for reuse is as good as free
    for (const auto &e : big_cache().parked) if (e.dev == dev) free_b += e.bytes;
  }
  size_t budget = free_b / 2;
  if (da::config().block_bytes) budget = (size_t)da::config().block_bytes;   // DYNAALIGN_BLOCK_BYTES
  int64_t rows = (int64_t)(budget / ((size_t)n * bytes_per_elem));
  rows = rows / 128 * 128;
  if (rows < 128) rows = 128;
  return std::min(rows, n);
}

}  // namespace
}  // namespace da

using namespace da;

// ---- run-time switches: parsed once (da_common.hpp Config) -----------------------------------------------------------------------
namespace da {
namespace {
bool env_flag(const char *name) { return getenv(name) != nullptr; }
int64_t env_i64(const char *name, int64_t dflt) { const char *e = getenv(name); return e ? atoll(e) : dflt; }
uint64_t env_u64(const char *name, uint64_t dflt) { const char *e = getenv(name); return e ? strtoull(e, nullptr, 10) : dflt; }
Config parse_config() {
  Config c;
  c.mh_no_dedup = env_flag("DYNAALIGN_MH_NO_DEDUP");
  c.mh_no_sparse = env_flag("DYNAALIGN_MH_NO_SPARSE");
  c.mh_no_pipe = env_flag("DYNAALIGN_MH_NO_PIPE");
  c.mh_pipe_one_stream = env_flag("DYNAALIGN_MH_PIPE_ONE_STREAM");
  c.mh_dedup_min_n = env_i64("DYNAALIGN_MH_DEDUP_MIN_N", -1);
  c.mh_dedup_max_pct = env_i64("DYNAALIGN_MH_DEDUP_MAX_PCT", -1);
  c.mh_sparse_max_pairs = std::min<uint64_t>(env_u64("DYNAALIGN_MH_SPARSE_MAX_PAIRS", 400000000ull), 0xfffffff0ull);   // (32-bit entry offsets)
  c.mh_no_hybrid = env_flag("DYNAALIGN_MH_NO_HYBRID");
  c.mh_hybrid_min_n = env_i64("DYNAALIGN_MH_HYBRID_MIN_N", -1);
  c.mh_hybrid_dedup = env_flag("DYNAALIGN_MH_HYBRID_DEDUP");
  if (const char *e = getenv("DYNAALIGN_MH_EXPAND"))
    c.mh_expand = !strcmp(e, "rows") ? 1 : !strcmp(e, "rowspipe") ? 2 : !strcmp(e, "pipe") ? 3 : !strcmp(e, "tiles") ? 4 : 0;
  c.mh_pipe_step = (int)std::max<int64_t>(0, env_i64("DYNAALIGN_MH_PIPE_STEP", 0));
  c.mh_pipe_wg = (int)std::max<int64_t>(0, std::min<int64_t>(4, env_i64("DYNAALIGN_MH_PIPE_WG", 0)));
  c.mh_pipe_head = (int)std::max<int64_t>(0, env_i64("DYNAALIGN_MH_PIPE_HEAD", 0));
  const int pb = (int)env_i64("DYNAALIGN_PLANE_BITS", 0);
  c.plane_bits = (pb == 32 || pb == 16 || pb == 15 || pb == 14 || pb == 12) ? pb : 0;
  c.k2_no_asm = env_flag("DYNAALIGN_K2_NO_ASM");
  c.k2_persist = env_flag("DYNAALIGN_K2_PERSIST");
  c.k2_wg_per_cu = (int)std::max<int64_t>(0, env_i64("DYNAALIGN_K2_WG_PER_CU", 0));
  c.nw_no_dedup = env_flag("DYNAALIGN_NW_NO_DEDUP");
  c.nw_int32 = env_flag("DYNAALIGN_NW_INT32");
  c.nw_no_prefix = env_flag("DYNAALIGN_NW_NO_PREFIX_SHARE");
  c.nw_dedup_min_n = env_i64("DYNAALIGN_NW_DEDUP_MIN_N", -1);
  c.no_host_widen = env_flag("DYNAALIGN_NO_HOST_WIDEN");
  c.plain_d2h = env_flag("DYNAALIGN_PLAIN_D2H");
  c.no_buffer_cache = env_flag("DYNAALIGN_NO_BUFFER_CACHE");
  c.d2h_threads = (int)std::max<int64_t>(0, env_i64("DYNAALIGN_D2H_THREADS", 0));
  c.buffer_cache_pct = (int)std::max<int64_t>(0, std::min<int64_t>(90, env_i64("DYNAALIGN_BUFFER_CACHE_PCT", 30)));
  c.block_bytes = env_u64("DYNAALIGN_BLOCK_BYTES", 0);
  c.trace = env_flag("DYNAALIGN_TRACE");
  c.louvain_debug = env_flag("DYNAALIGN_LOUVAIN_DEBUG");
  c.louvain_threads = (int)std::max<int64_t>(0, env_i64("DYNAALIGN_LOUVAIN_THREADS", 0));
  return c;
}
std::atomic<const Config *> g_config{nullptr};
}  // namespace
const Config &config() {
  const Config *c = g_config.load(std::memory_order_acquire);
  if (!c) {
    const Config *fresh = new Config(parse_config());
    if (g_config.compare_exchange_strong(c, fresh, std::memory_order_acq_rel)) c = fresh;
    else delete fresh;
  }
  return *c;
}
}  // namespace da
// test hook: parse the environment again (the previous struct stays allocated: another host thread may still be reading it)
extern "C" void da_config_reload(void) { da::g_config.store(new da::Config(da::parse_config()), std::memory_order_release); }

extern "C" {

const char *da_last_error(void) { return last_error_ref().c_str(); }

const char *da_status_message(int status) {
  switch (status) {
    case DA_ERR_EMPTY_INPUT: return "Input sequences vector cannot be empty";      // src/minHash.cpp:122
    case DA_ERR_BAD_K: return "'k' must be a positive integer";                   // src/minHash.cpp:126
    case DA_ERR_BAD_NHASH: return "Number of hash functions must be positive";     // src/minHash.cpp:130
    case DA_ERR_BAD_MATRIX: return "Invalid substitution matrix name: %s";         // src/pairwiseSeqAlign.cpp:204
    case DA_ERR_BAD_RESIDUE_SEQ1: return "Invalid amino acid in sequence1: %c";    // :242
    case DA_ERR_BAD_RESIDUE_SEQ2: return "Invalid amino acid in sequence2: %c";    // :249
    case DA_ERR_NOMEM: return "out of memory";
    case DA_ERR_NO_DEVICE: return "no usable HIP device";
    case DA_ERR_HIP: return "HIP runtime error";
    case DA_ERR_UNSUPPORTED: return "unsupported by the gfx950 kernels";
    case DA_ERR_BAD_ARG: return "bad argument";
    default: return "";
  }
}

int da_abi_version(void) { return DA_ABI_VERSION; }

size_t da_release_device_memory(void) {
  (void)da::destroy_cached_comms();          // cached RCCL communicators (multi-device entry points) go too
#ifdef DA_K2_EXPERIMENTS
  return big_cache().release(-1) + da::release_compare_scratch();
#else
  return big_cache().release(-1);
#endif
}

int da_device_count(void) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
  return cnt;
}

// std::mt19937 restated (ISO C++ [rand.predef]); the reference draws the seeds
// through uniform_int_distribution<uint32_t> over the full range, which returns
// the raw engine output on libstdc++ (reference src/minHash.cpp:75-80).
int da_hash_family_seeds(uint32_t seed, int n_hash, uint32_t *seeds_out) {
  if (n_hash < 0 || (n_hash > 0 && !seeds_out)) return fail(DA_ERR_BAD_ARG, "bad n_hash / seeds_out");
  uint32_t s[624];
  s[0] = seed;
  for (uint32_t i = 1; i < 624; ++i) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + i;
  int pos = 624;
  for (int o = 0; o < n_hash; ++o) {
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
    seeds_out[o] = y;
  }
  return DA_OK;
}

uint32_t da_random_seed(void) { return std::random_device{}(); }  // src/minHash.cpp:73

int da_matrix_id(const char *matrix_name) {
  if (matrix_name)
    for (int i = 0; i < matrix_count_host(); ++i)
      if (strcmp(matrix_name, matrix_name_host(i)) == 0) return i;
  fail(DA_ERR_BAD_MATRIX, "Invalid substitution matrix name: %s", matrix_name ? matrix_name : "");
  return -1;
}

// --------------------------------------------------------------- dev entry

int da_dev_minhash_signatures(const uint8_t *d_residues, const int64_t *d_offsets, int64_t n,
                              int64_t total_residues, int64_t max_len, int k, int n_hash,
                              const uint32_t *d_seeds, uint32_t *d_sig, int64_t ld_sig, void *stream) {
  (void)total_residues; (void)max_len;
  int rc = validate_mh(n, k, n_hash);
  if (rc != DA_OK) return rc;
  if (!d_residues || !d_offsets || !d_seeds || !d_sig) return fail(DA_ERR_BAD_ARG, "NULL device pointer");
  if (ld_sig < n_hash) return fail(DA_ERR_BAD_ARG, "ld_sig (%lld) < n_hash (%d)", (long long)ld_sig, n_hash);
  return launch_minhash_signatures(d_residues, d_offsets, n, k, n_hash, d_seeds, d_sig, ld_sig,
                                   static_cast<hipStream_t>(stream));
}

// signatures -> compare operand: dictionary codes (16 planes / group) when n allows, raw values otherwise
static int env_plane_bits() {   // DYNAALIGN_PLANE_BITS: 32 = raw planes, 12 / 16 = at least that many code planes
  return da::config().plane_bits;
}
// The sparse route (minhash_kernels.hip "SPARSE route"): the caller of build_planes asks for it by passing a SparseHint; when the
// dictionary says the signatures rarely agree (few matching incidences, no large class) build_planes returns with take = true and
// WITHOUT building the bit planes -- the caller then runs launch_mh_sparse on the dictionary codes in the workspace.
struct SparseHint { bool take = false; int max_ids = 0; uint64_t pairs = 0, max_mult = 0; };
// The heavy / rare split (dict_kernels.hip k_hy_split): asked for the same way; when the dictionaries need 12 - 16 code planes but all except a few
// matching incidences sit on each column's 254 most frequent values, build_planes returns EIGHT planes of dense codes for those (take = true,
// *bits_out = 8) and leaves the codes of the rare repeated values in the workspace (sparse_codes): the caller runs launch_mh_sparse_lists on them
// and, after the dense compare, launch_mh_sparse_fixup.
constexpr int HY_KEEP = 254;
struct HybridHint { int only_from_bits = 12; bool take = false; int max_ids = 0, bits_before = 0; uint64_t pairs = 0, max_mult = 0; const uint16_t *sparse_codes = nullptr; int64_t ld_ids = 0; };
// min_bits: 0 = as few code planes as the data needs, 12 / 16 = at least that many, 32 = raw planes
static int build_planes(const uint32_t *d_sig, int64_t ld_sig, int64_t n, int n_hash, int min_bits, void *d_work,
                        size_t work_bytes, uint32_t *d_planes, int *bits_out, hipStream_t stream, SparseHint *sparse = nullptr,
                        HybridHint *hybrid = nullptr) {
  int rc;
  const int env = env_plane_bits();
  if (env > min_bits) min_bits = env;
  if (n <= DA_DICT_MAX_N && min_bits != 32) {
    if (!d_work || work_bytes < mh_planes_workspace_bytes(n, n_hash))
      return fail(DA_ERR_BAD_ARG, "workspace smaller than da_mh_planes_workspace_bytes(n, n_hash)");
    int *d_status = nullptr;
    if ((rc = launch_mh_dictionary(d_sig, ld_sig, n, n_hash, d_work, &d_status, stream)) != DA_OK) return rc;
    int status[2] = {0, 0};
    DA_HIP_TRY(hipMemcpyAsync(status, d_status, sizeof(status), hipMemcpyDeviceToHost, stream));
    DA_HIP_TRY(hipStreamSynchronize(stream));
    int bits_data = status[0] == 0 ? mh_plane_bits_for(status[1]) : 32;
    int64_t hy_min_n = 16384;                                       // below that the compare is too short to pay for the list kernels
    if (da::config().mh_hybrid_min_n >= 0) hy_min_n = da::config().mh_hybrid_min_n;   // DYNAALIGN_MH_HYBRID_MIN_N (tests lower it)
    const bool sparse_ok = sparse && status[0] == 0 && status[1] >= 1 && status[1] <= 32768 && min_bits == 0 && n >= 2048 && n_hash + 1 <= 2048 &&
                           !da::config().mh_no_sparse;
    const bool hybrid_ok = hybrid && status[0] == 0 && bits_data >= hybrid->only_from_bits && bits_data <= 16 && min_bits == 0 && status[1] <= 32768 && n >= hy_min_n && n <= 131072 &&
                           n_hash > 32 && n_hash + 1 <= 2048 && !da::config().mh_no_hybrid && !da::config().k2_no_asm;
    if (sparse_ok || hybrid_ok) {
      // how many (pair, hash function) incidences match, and how large is the largest class -- over all repeated values (sparse route) and over the values
      // outside each column's HY_KEEP most frequent ones (heavy / rare split)?  One kernel, one more read-back (~0.15 ms)
      int64_t ld_ids = 0;
      const uint16_t *ids = mh_dictionary_codes(d_work, n, n_hash, &ld_ids);
      unsigned long long *d_stats = reinterpret_cast<unsigned long long *>(d_status + 4);     // behind the two status words (the workspace ends 256 bytes later)
      const uint16_t *dense = nullptr, *rare = nullptr;
      unsigned long long st[4] = {0, 0, 0, 0};                        // rare incidences, largest rare class, all incidences, largest class
      if (hybrid_ok) {
        if ((rc = launch_mh_heavy_split(d_work, n, n_hash, status[1], HY_KEEP, &dense, &rare, d_stats, stream)) != DA_OK) return rc;
        DA_HIP_TRY(hipMemcpyAsync(st, d_stats, sizeof(st), hipMemcpyDeviceToHost, stream));
      } else {
        if ((rc = launch_mh_sparse_count(ids, ld_ids, n, n_hash, status[1], d_stats, stream)) != DA_OK) return rc;
        DA_HIP_TRY(hipMemcpyAsync(st + 2, d_stats, 16, hipMemcpyDeviceToHost, stream));
      }
      DA_HIP_TRY(hipStreamSynchronize(stream));
      const double pairs_all = 0.5 * (double)n * (double)(n - 1);
      if (sparse_ok) {
        sparse->pairs = st[2]; sparse->max_mult = st[3]; sparse->max_ids = status[1];
        // admission: the bucket phase costs ~37 ps per incidence (5.1 ms for 1.37e8) and the tile pass ~2.5 ps per pair, the dense compare ~6.3 ps
        // per pair at n_hash = 500 -- so the route pays while the average number of matches per pair stays below ~0.1 (x n_hash / 500);
        // plus a memory cap on the bucket and a cap on the class size (a thread walks its class)
        if (st[2] <= mh_sparse_pairs_limit() && st[3] <= 4096 && (double)st[2] <= pairs_all * (double)n_hash / 5000.0) {
          sparse->take = true; *bits_out = 0; return DA_OK;
        }
      }
      // heavy / rare split: four to eight planes less = 30 - 47 % of the compare's plane work (~2 ps per pair and 4 planes at n_hash = 500), against
      // ~37 ps per rare incidence in the list kernels + ~1 ms of fixed cost: taken while the rare values match in < 0.02 of the pairs (x n_hash / 500)
      if (hybrid_ok && st[0] <= mh_sparse_pairs_limit() && st[1] <= 4096 && (double)st[0] <= pairs_all * (double)n_hash / 25000.0) {
        hybrid->take = true; hybrid->max_ids = status[1]; hybrid->bits_before = bits_data; hybrid->pairs = st[0]; hybrid->max_mult = st[1];
        hybrid->sparse_codes = rare; hybrid->ld_ids = ld_ids;
        *bits_out = 8;
        return launch_ids_to_planes(d_work, n, n_hash, 8, d_planes, stream, dense);
      }
    }
    if (status[0] == 0) {
      int bits = bits_data;
      if (bits < min_bits) bits = min_bits;
      *bits_out = bits;
      return launch_ids_to_planes(d_work, n, n_hash, bits, d_planes, stream);
    }
  }
  *bits_out = 32;
  return launch_sig_to_planes(d_sig, ld_sig, n, n_hash, d_planes, stream);
}

size_t da_mh_planes_workspace_bytes(int64_t n, int n_hash) { return mh_planes_workspace_bytes(n, n_hash); }
int64_t da_mh_planes_words(int64_t n, int n_hash) { return mh_planes_words(n, n_hash); }

int da_dev_mh_planes(const uint32_t *d_sig, int64_t ld_sig, int64_t n, int n_hash, int min_plane_bits, void *d_work,
                     size_t work_bytes, uint32_t *d_planes, int64_t planes_words, int *plane_bits_out, void *stream) {
  if (n <= 0) return fail(DA_ERR_EMPTY_INPUT, "%s", da_status_message(DA_ERR_EMPTY_INPUT));
  if (n_hash <= 0) return fail(DA_ERR_BAD_NHASH, "%s", da_status_message(DA_ERR_BAD_NHASH));
  if (!d_sig || !d_planes || !plane_bits_out) return fail(DA_ERR_BAD_ARG, "NULL pointer");
  if (ld_sig < n_hash) return fail(DA_ERR_BAD_ARG, "ld_sig (%lld) < n_hash (%d)", (long long)ld_sig, n_hash);
  if (min_plane_bits != 0 && min_plane_bits != 12 && min_plane_bits != 14 && min_plane_bits != 15 && min_plane_bits != 16 && min_plane_bits != 32)
    return fail(DA_ERR_BAD_ARG, "min_plane_bits must be 0, 12, 14, 15, 16 or 32 (got %d)", min_plane_bits);
  if ((reinterpret_cast<uintptr_t>(d_planes) & 15) || planes_words < mh_planes_words(n, n_hash))
    return fail(DA_ERR_BAD_ARG, "bit-plane buffer must be 16-byte aligned and hold da_mh_planes_words(n, n_hash) = %lld words",
                (long long)mh_planes_words(n, n_hash));
  if (reinterpret_cast<uintptr_t>(d_work) & 255) return fail(DA_ERR_BAD_ARG, "workspace must be 256-byte aligned");
  return build_planes(d_sig, ld_sig, n, n_hash, min_plane_bits, d_work, work_bytes, d_planes, plane_bits_out,
                      static_cast<hipStream_t>(stream));
}

int da_dev_mh_compare(const uint32_t *d_planes, int plane_bits, int64_t n, int n_hash,
                      int64_t row_begin, int64_t row_end, int symmetric, int kind, void *d_out,
                      int64_t ld, void *stream) {
  if (n <= 0) return fail(DA_ERR_EMPTY_INPUT, "%s", da_status_message(DA_ERR_EMPTY_INPUT));
  if (n_hash <= 0) return fail(DA_ERR_BAD_NHASH, "%s", da_status_message(DA_ERR_BAD_NHASH));
  if (!d_planes || !d_out) return fail(DA_ERR_BAD_ARG, "NULL device pointer");
  if (row_begin < 0 || row_end > n || row_begin > row_end) return fail(DA_ERR_BAD_ARG, "bad row range");
  if (symmetric && (row_begin != 0 || row_end != n))
    return fail(DA_ERR_BAD_ARG, "symmetric mode needs the full row range");
  if (ld < n) return fail(DA_ERR_BAD_ARG, "ld (%lld) < n (%lld)", (long long)ld, (long long)n);
  if (kind != DA_OUT_F64 && kind != DA_OUT_COMPACT) return fail(DA_ERR_BAD_ARG, "bad output kind");
  if (n_hash > 65535)
    return fail(DA_ERR_UNSUPPORTED, "the compare kernel counts in 16 bits: n_hash <= 65535 (got %d)", n_hash);
  if (reinterpret_cast<uintptr_t>(d_planes) & 15) return fail(DA_ERR_BAD_ARG, "bit-plane buffer must be 16-byte aligned");
  if (plane_bits != 8 && plane_bits != 12 && plane_bits != 14 && plane_bits != 15 && plane_bits != 16 && plane_bits != 32)
    return fail(DA_ERR_BAD_ARG, "plane_bits must be 8, 12, 14, 15, 16 or 32 (got %d)", plane_bits);
  return launch_mh_compare(d_planes, n, n_hash, row_begin, row_end, symmetric != 0, kind, d_out, ld,
                           static_cast<hipStream_t>(stream), plane_bits);
}

// similarityMH, packed residues in HBM -> dense float64 n x n in HBM, as ONE call (K1 + K1b + K2 and the route below).
// Byte-identical sequences have identical signatures, hence identical rows and columns of the result: when at least 15 % of the
// sequences are duplicates the pipeline runs on the table of U unique strings (the plan of the NW route: nw_kernels.hip
// "duplicate sequences"; MinHash similarity IS symmetric, so the plain symmetric compare on U rows does) -- K2's work shrinks by
// (U/n)^2 -- and the n x n matrix is an index expansion of the U x U count table (launch_expand_unique's two streaming passes,
// 5.7 TB/s of stores).  Exact.  Taken when at most 60 % of the sequences are unique (below that K2's saving outweighs the expansion);
// otherwise (uniform peptides: nothing to collapse), or when the expansion's fast passes do not cover the shape
// (U > 65536, n_hash > 2047, n < 2048), the direct kernels run.  Synchronises `stream` (K1b reads the dictionary sizes back).
// DYNAALIGN_MH_NO_DEDUP=1 switches the route off.
// side stream + events of the pipelined form, pooled per device (creating them costs more than the route's small kernels)
struct PipeRes { hipStream_t side = nullptr, alt[2] = {nullptr, nullptr}; std::vector<hipEvent_t> ev; int dev = 0; };
struct PipePool { std::mutex m; std::vector<PipeRes *> idle[64]; };
static PipePool &pipe_pool() { static PipePool *p = new PipePool; return *p; }   // never destroyed: the HIP runtime may be gone at exit
static PipeRes *pipe_acquire(size_t events) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  PipeRes *r = nullptr;
  {
    std::lock_guard<std::mutex> g(pipe_pool().m);
    auto &v = pipe_pool().idle[dev & 63];
    if (!v.empty()) { r = v.back(); v.pop_back(); }
  }
  if (!r) {
    r = new PipeRes;
    r->dev = dev;
    if (hipStreamCreateWithFlags(&r->side, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&r->alt[0], hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&r->alt[1], hipStreamNonBlocking) != hipSuccess) { delete r; return nullptr; }
  }
  while (r->ev.size() < events) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) break;
    r->ev.push_back(e);
  }
  if (r->ev.size() < events) { std::lock_guard<std::mutex> g(pipe_pool().m); pipe_pool().idle[dev & 63].push_back(r); return nullptr; }
  return r;
}
static void pipe_release(PipeRes *r) {
  if (!r) return;
  std::lock_guard<std::mutex> g(pipe_pool().m);
  pipe_pool().idle[r->dev & 63].push_back(r);
}
struct MhRoute { int64_t n = 0, unique = 0, hybrid_pairs = -1; int taken = 0, plane_bits = 0, chunks = 0, expand_launches = 0, hybrid_bits_before = 0; float ms[6] = {0, 0, 0, 0, 0, 0}; };   // plan, K1 + K1b, K2, column gather, k_expand_rows, diagonal / border tiles
static MhRoute &mh_route() { static thread_local MhRoute r; return r; }

static int mh_full_symmetric(const uint8_t *d_res, const int64_t *d_off, int64_t n, int64_t total, int k, int n_hash,
                             const uint32_t *d_seeds, double *d_out, int64_t ld, hipStream_t stream) {
  int rc;
  MhRoute &route = mh_route();
  route = MhRoute();
  route.n = route.unique = n;
  hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  struct EvGuard { hipEvent_t *e; ~EvGuard() { for (int i = 0; i < 7; ++i) if (e[i]) (void)hipEventDestroy(e[i]); } } guard{ev};
  for (auto &e : ev) DA_HIP_TRY(hipEventCreate(&e));
  DA_HIP_TRY(hipEventRecord(ev[0], stream));
  const int64_t lds = sig_ld_for(n_hash);
  int64_t min_n = 2048;
  if (da::config().mh_dedup_min_n >= 0) min_n = da::config().mh_dedup_min_n;   // DYNAALIGN_MH_DEDUP_MIN_N: tests lower it to reach the route with small inputs
  const bool eligible = n >= min_n && n <= 0x7ffffff0LL && total > 0 && n_hash <= 2047 && !da::config().mh_no_dedup;
  DevBuf plan_work;
  NwDedupPlan p{};
  int64_t U = n;
  bool take = false;
  std::vector<int32_t> ub;
  if (eligible) {
    if ((rc = plan_work.alloc(nw_dedup_workspace_bytes(n, total))) != DA_OK) return rc;
    p = nw_dedup_layout(plan_work.p, n, total);
    if ((rc = launch_nw_dedup_count(d_res, d_off, n, p, stream)) != DA_OK) return rc;
    int32_t M = 0, S = 0;
    DA_HIP_TRY(hipMemcpyAsync(&M, p.pm + n, 4, hipMemcpyDeviceToHost, stream));
    DA_HIP_TRY(hipMemcpyAsync(&S, p.ps + n, 4, hipMemcpyDeviceToHost, stream));
    DA_HIP_TRY(hipStreamSynchronize(stream));
    U = (int64_t)M + S;
    route.unique = U;
    // K2 on U rows + column gather + expansion (0.5 + 1.6 f + 22 f^2 + 4.1 f + 14.4 ms at N = 100k, f = U / n) against K1 + K1b + K2 with its
    // own float64 stores (27.2 ms): the route pays below f = 0.63
    // ... with the tile expansion.  With the row expansion (no gather, no copy of the table) the route costs ~1.5 + 26 f^2 + 14 ms one kernel after the
    // other: 24.4 ms at f = 0.60, 26.6 at f = 0.65 against 27 (12 code planes) - 30 ms (14) for the direct kernels (profiles/r03_q_dedup_threshold.txt): 0.68
    int64_t max_pct = expand_stream_ok(n, std::min<int64_t>(U, 65536), n_hash, d_out, ld) ? 68 : 60;
    if (da::config().mh_dedup_max_pct >= 0) max_pct = da::config().mh_dedup_max_pct;   // DYNAALIGN_MH_DEDUP_MAX_PCT
    take = U > 0 && U * 100 <= n * max_pct && expand_rows_workspace_bytes(n, U, DA_OUT_F64, false, n_hash, 0) != 0;
    if (take && (rc = launch_nw_dedup_build(d_res, d_off, n, U, p, stream, true)) != DA_OK) return rc;   // ids by first occurrence
    const int form0 = da::config().mh_expand;                      // DYNAALIGN_MH_EXPAND: 0 default, 1 rows, 2 rowspipe, 3 pipe, 4 tiles
    const bool rows_form = form0 <= 2 && expand_stream_ok(n, U, n_hash, d_out, ld);
    if (take && !rows_form) {                                      // (the row expansion needs no schedule of output bands: one read-back less)
      // ub[b] = unique ids the input rows [0, 1024 b) use (prefix counts of the plan, read at the band boundaries): the TILE pipeline's schedule
      const int64_t NB = mh_sym_bands(n);
      std::vector<int32_t> a((size_t)NB), b((size_t)NB);
      DA_HIP_TRY(hipMemcpy2DAsync(a.data(), 4, p.pm, 4096, 4, (size_t)NB, hipMemcpyDeviceToHost, stream));
      DA_HIP_TRY(hipMemcpy2DAsync(b.data(), 4, p.ps, 4096, 4, (size_t)NB, hipMemcpyDeviceToHost, stream));
      DA_HIP_TRY(hipStreamSynchronize(stream));
      ub.resize((size_t)NB + 1);
      for (int64_t q = 0; q < NB; ++q) ub[(size_t)q] = a[(size_t)q] + b[(size_t)q];
      ub[(size_t)NB] = (int32_t)U;
    }
  }
  DA_HIP_TRY(hipEventRecord(ev[1], stream));
  const int64_t m = take ? U : n;                                   // rows the kernels see
  const uint8_t *res_m = take ? p.ucodes : d_res;
  const int64_t *off_m = take ? p.uoff : d_off;
  DevBuf sig, planes, pwork;
  const size_t wb = mh_planes_workspace_bytes(m, n_hash);
  if ((rc = sig.alloc((size_t)m * lds * sizeof(uint32_t))) != DA_OK) return rc;
  if ((rc = planes.alloc((size_t)mh_planes_words(m, n_hash) * sizeof(uint32_t))) != DA_OK) return rc;
  if ((rc = pwork.alloc(wb)) != DA_OK) return rc;
  if ((rc = launch_minhash_signatures(res_m, off_m, m, k, n_hash, d_seeds, sig.as<uint32_t>(), lds, stream)) != DA_OK) return rc;
  int bits = 32;
  SparseHint sp;
  // the sparse route is only worth asking about when the input has few duplicates (clustered inputs have large classes: the dense kernels win)
  const bool ask_sparse = !take && U * 10 >= n * 9;
  // the heavy / rare split (8 dense planes + incidence lists for the rare values): the direct route; the duplicate route's table compare on request
  // (duplicate route: when the table's dictionaries need more than 12 planes -- there is no banded kernel for those, so the split is what lets the
  // route pipeline at all; with 12 planes the pipelined route gains 0.2 ms of 16.8 from it at 100k, profiles/r04_l_*: on request only)
  HybridHint hy;
  if (take && !da::config().mh_hybrid_dedup) hy.only_from_bits = 13;
  if ((rc = build_planes(sig.as<uint32_t>(), lds, m, n_hash, 0, pwork.p, wb, planes.as<uint32_t>(), &bits, stream, ask_sparse ? &sp : nullptr, &hy)) != DA_OK) return rc;
  DevBuf hy_scratch, hy_entries32, hy_entries;
  if (hy.take) {
    // the split is an optimisation: when the lists' scratch does not fit, the planes are rebuilt with all the code bits
    const bool fits = hy_scratch.alloc(mh_sparse_scratch_words(m, n_hash, hy.max_ids, hy.ld_ids) * 4) == DA_OK &&
                      hy_entries32.alloc((size_t)hy.pairs * 4 + 16) == DA_OK && hy_entries.alloc((size_t)hy.pairs * 2 + 16) == DA_OK;
    if (!fits) {
      (void)hipGetLastError();
      hy = HybridHint();
      if ((rc = build_planes(sig.as<uint32_t>(), lds, m, n_hash, 0, pwork.p, wb, planes.as<uint32_t>(), &bits, stream, nullptr, nullptr)) != DA_OK) return rc;
    }
  }
  route.plane_bits = bits;
  route.hybrid_pairs = hy.take ? (int64_t)hy.pairs : -1;
  route.hybrid_bits_before = hy.bits_before;
  DA_HIP_TRY(hipEventRecord(ev[2], stream));
  // lists of the rare values' incidences (tile by tile) and, after a compare, their addition to its result
  auto hy_lists = [&](hipStream_t st) -> int {
    return launch_mh_sparse_lists(hy.sparse_codes, hy.ld_ids, m, n_hash, hy.max_ids, hy.pairs, hy_scratch.as<uint32_t>(), hy_entries32.as<uint32_t>(),
                                  hy_entries.as<uint16_t>(), st);
  };
  auto hy_fixup = [&](int kind, void *out, int64_t ld_out, int64_t tr0, int64_t tr1, hipStream_t st) -> int {
    return launch_mh_sparse_fixup(hy_scratch.as<uint32_t>(), hy_entries.as<uint16_t>(), m, n_hash, hy.max_ids, hy.ld_ids, kind, out, ld_out, tr0, tr1, st);
  };
  const int64_t TR_ALL = (m + 127) / 128;
  if (!take && sp.take) {
    // SPARSE route: the signatures rarely agree -- the matching incidences are enumerated from the dictionary codes, bucketed per output
    // tile and every tile written once (exact; minhash_kernels.hip).  ms[2] = link + walk + scan (buckets), ms[4] = the tile pass.
    int64_t ld_ids = 0;
    const uint16_t *ids = mh_dictionary_codes(pwork.p, n, n_hash, &ld_ids);
    DevBuf scratch, entries32, entries;
    // the route is an optimisation: when its scratch does not fit (or the incidence count exceeds what its 32-bit offsets cover) the
    // call does not fail -- the dense compare below needs none of it
    const bool fits = sp.pairs <= 0xfffffff0ull && scratch.alloc(mh_sparse_scratch_words(n, n_hash, sp.max_ids, ld_ids) * 4) == DA_OK &&
                      entries32.alloc((size_t)sp.pairs * 4 + 16) == DA_OK && entries.alloc((size_t)sp.pairs * 2 + 16) == DA_OK;
    if (!fits) {
      (void)hipGetLastError();
      sp.take = false;
      if ((rc = build_planes(sig.as<uint32_t>(), lds, m, n_hash, 0, pwork.p, wb, planes.as<uint32_t>(), &bits, stream, nullptr)) != DA_OK) return rc;
      route.plane_bits = bits;
      DA_HIP_TRY(hipEventRecord(ev[2], stream));
    }
    else if ((rc = launch_mh_sparse(ids, ld_ids, n, n_hash, sp.max_ids, sp.pairs, scratch.as<uint32_t>(), entries32.as<uint32_t>(), entries.as<uint16_t>(), d_out, ld,
                               stream, ev[3])) != DA_OK) return rc;
    if (sp.take) {
    DA_HIP_TRY(hipEventRecord(ev[4], stream));
    DA_HIP_TRY(hipStreamSynchronize(stream));
    route.taken = 2;
    route.unique = (int64_t)sp.pairs;                               // (reported as `sparse_pairs` by the Python wrapper)
    (void)hipEventElapsedTime(&route.ms[0], ev[0], ev[1]);
    (void)hipEventElapsedTime(&route.ms[1], ev[1], ev[2]);
    (void)hipEventElapsedTime(&route.ms[2], ev[2], ev[3]);
    (void)hipEventElapsedTime(&route.ms[4], ev[3], ev[4]);
    return DA_OK;
    }
  }
  if (!take) {
    // heavy / rare split: the list kernels (latency-bound, ~0.9 ms at 100k) run on a side stream beside the dense compare; without a side stream, before it
    PipeRes *pr = hy.take ? pipe_acquire(1) : nullptr;
    struct PipeGuard { PipeRes *r; ~PipeGuard() { if (r) { (void)hipStreamSynchronize(r->side); pipe_release(r); } } } pguard{pr};
    if (pr) {
      DA_HIP_TRY(hipStreamWaitEvent(pr->side, ev[2], 0));
      if ((rc = hy_lists(pr->side)) != DA_OK) return rc;
      DA_HIP_TRY(hipEventRecord(pr->ev[0], pr->side));
    } else if (hy.take && (rc = hy_lists(stream)) != DA_OK) return rc;
    if ((rc = launch_mh_compare(planes.as<uint32_t>(), n, n_hash, 0, n, true, DA_OUT_F64, d_out, ld, stream, bits)) != DA_OK) return rc;
    if (pr) DA_HIP_TRY(hipStreamWaitEvent(stream, pr->ev[0], 0));
    if (hy.take && (rc = hy_fixup(DA_OUT_F64, d_out, ld, 0, TR_ALL, stream)) != DA_OK) return rc;
    DA_HIP_TRY(hipEventRecord(ev[3], stream));
    DA_HIP_TRY(hipStreamSynchronize(stream));
    (void)hipEventElapsedTime(&route.ms[0], ev[0], ev[1]);
    (void)hipEventElapsedTime(&route.ms[1], ev[1], ev[2]);
    (void)hipEventElapsedTime(&route.ms[2], ev[2], ev[3]);
    return DA_OK;
  }
  const int64_t ld_d = (U + 7) / 8 * 8;
  DevBuf dtab, ftab;
  if ((rc = dtab.alloc((size_t)U * (size_t)ld_d * 2)) != DA_OK) return rc;
  // The expansion has two kernels families and each a pipelined form (DYNAALIGN_MH_EXPAND = rowspipe | rows | pipe | tiles; default: the first
  // whose shape test passes; DYNAALIGN_MH_NO_PIPE=1 takes the pipelined forms out):
  //   rows    K2 on the table, then k_expand_stream: every output row written once from its table row in LDS (no gathered copy)
  //   tiles   K2, column gather (k_gather_columns), tile expansion (k_expand_rows) + diagonal / border tiles, one after the other
  //   rowspipe / pipe   the same kernels with K2 run band by band on a side stream while finished table rows are expanded
  static const char *const form_names[5] = {"", "rows", "rowspipe", "pipe", "tiles"};
  const std::string form = form_names[da::config().mh_expand];
  const bool may_pipe = !da::config().mh_no_pipe && mh_compare_bands_ok(U, n_hash, bits, dtab.p, ld_d);
  if ((form.empty() || form == "rows" || form == "rowspipe") && expand_stream_ok(n, U, n_hash, d_out, ld)) {
    DevBuf lists;
    if ((rc = lists.alloc(expand_stream_scratch_bytes(n, U))) != DA_OK) return rc;
    if (form != "rows" && may_pipe) do {
      // the row expansion PIPELINED with K2: the first `head` bands of the table at full occupancy, the rest by the persistent kernel with `wg`
      // workgroup(s) per CU on a side stream (one 36 KB ring fits beside the expansion's 94 KB row) while the finished table rows are expanded
      const int64_t KB = mh_sym_bands(U);
      // K2 rings (36 KB, 128 VGPRs x 4 waves) that fit beside one expansion workgroup: two when its LDS row is packed (n_hash <= 511), else one --
      // and a slower K2 wants a head start and smaller chunks (profiles/r03_m_pipeline_rows_*.txt: head x step sweeps for both)
      const bool two = expand_stream_packed(n_hash);
      int wg = two ? 2 : 1, head = two ? 1 : 4, step = two ? 8 : 4;
      if (da::config().mh_pipe_step > 0) step = da::config().mh_pipe_step;       // DYNAALIGN_MH_PIPE_STEP / _WG / _HEAD / _ONE_STREAM: schedule experiments
      if (da::config().mh_pipe_wg > 0) wg = da::config().mh_pipe_wg;
      if (da::config().mh_pipe_head > 0) head = da::config().mh_pipe_head;
      const bool alt = !da::config().mh_pipe_one_stream;
      std::vector<int64_t> cuts{0};
      while (cuts.back() < KB) cuts.push_back(std::min(KB, cuts.back() + (cuts.size() == 1 ? head : step)));
      const size_t C = cuts.size() - 1;
      PipeRes *pr = pipe_acquire(3 * C + 2);
      if (!pr) break;   // no side stream / events to be had: the one-stream form below needs none
      struct PipeGuard { PipeRes *r; ~PipeGuard() { (void)hipStreamSynchronize(r->side); (void)hipStreamSynchronize(r->alt[0]); (void)hipStreamSynchronize(r->alt[1]); pipe_release(r); } } pguard{pr};
      hipEvent_t *pe = pr->ev.data();                                // per chunk: table rows done, rows begin / end; [3 C]: lists done
      // beside the first band on the side stream: the table's diagonal / border tiles and the copy lists, both on the caller's stream
      DA_HIP_TRY(hipStreamWaitEvent(pr->side, ev[2], 0));
      if ((rc = launch_mh_compare_edges_u16(planes.as<uint32_t>(), U, n_hash, dtab.as<uint16_t>(), ld_d, stream, bits)) != DA_OK) return rc;
      if ((rc = launch_expand_stream_lists(p.uidx, n, U, lists.p, stream)) != DA_OK) return rc;
      DA_HIP_TRY(hipEventRecord(pe[3 * C], stream));
      if (hy.take) {     // the rare values' lists behind them on the caller's stream (on the idle expansion stream instead: 17.3 vs 16.6 ms, round 4)
        if ((rc = hy_lists(stream)) != DA_OK) return rc;
        DA_HIP_TRY(hipEventRecord(pe[3 * C + 1], stream));
      }
      if (alt) { DA_HIP_TRY(hipStreamWaitEvent(pr->alt[0], pe[3 * C], 0)); DA_HIP_TRY(hipStreamWaitEvent(pr->alt[1], pe[3 * C], 0)); }
      for (size_t c = 0; c < C; ++c) {
        if ((rc = launch_mh_compare_bands_u16(planes.as<uint32_t>(), U, n_hash, dtab.as<uint16_t>(), ld_d, cuts[c], cuts[c + 1], c == 0 ? 4 : wg, pr->side, bits)) != DA_OK) return rc;
        if (hy.take) {     // the rare values' incidences of these tile rows (direct + mirrored elements), once the edge tiles and the lists are there
          if (c == 0) { DA_HIP_TRY(hipStreamWaitEvent(pr->side, pe[3 * C], 0)); DA_HIP_TRY(hipStreamWaitEvent(pr->side, pe[3 * C + 1], 0)); }
          if ((rc = hy_fixup(DA_OUT_COMPACT, dtab.p, ld_d, cuts[c] * 8, cuts[c + 1] * 8, pr->side)) != DA_OK) return rc;
        }
        DA_HIP_TRY(hipEventRecord(pe[3 * c], pr->side));
        const hipStream_t es = alt ? pr->alt[c & 1] : stream;
        DA_HIP_TRY(hipStreamWaitEvent(es, pe[3 * c], 0));
        DA_HIP_TRY(hipEventRecord(pe[3 * c + 1], es));
        if ((rc = launch_expand_stream_rows(dtab.as<uint16_t>(), ld_d, p.uidx, n, U, n_hash, d_out, ld, lists.p, cuts[c] * 1024, cuts[c + 1] * 1024, es, (int)c)) != DA_OK) return rc;
        DA_HIP_TRY(hipEventRecord(pe[3 * c + 2], es));
      }
      if (alt) for (size_t c = (C >= 2 ? C - 2 : 0); c < C; ++c) DA_HIP_TRY(hipStreamWaitEvent(stream, pe[3 * c + 2], 0));
      DA_HIP_TRY(hipEventRecord(ev[5], stream));
      DA_HIP_TRY(hipStreamSynchronize(stream));
      route.taken = 5;
      route.chunks = route.expand_launches = (int)C;
      (void)hipEventElapsedTime(&route.ms[0], ev[0], ev[1]);
      (void)hipEventElapsedTime(&route.ms[1], ev[1], ev[2]);
      (void)hipEventElapsedTime(&route.ms[2], ev[2], pe[3 * (C - 1)]);   // the compare's span on the side stream
      (void)hipEventElapsedTime(&route.ms[3], ev[2], pe[3 * C]);          // the copy lists (beside the compare)
      float covered_to = 0;
      for (size_t c = 0; c < C; ++c) {                                    // ms[4]: the time some k_expand_stream launch was running (union)
        float b = 0, e = 0;
        (void)hipEventElapsedTime(&b, ev[2], pe[3 * c + 1]);
        (void)hipEventElapsedTime(&e, ev[2], pe[3 * c + 2]);
        if (e > covered_to) { route.ms[4] += e - std::max(b, covered_to); covered_to = e; }
      }
      return DA_OK;
    } while (0);
    if (hy.take && (rc = hy_lists(stream)) != DA_OK) return rc;
    if ((rc = launch_mh_compare(planes.as<uint32_t>(), U, n_hash, 0, U, true, DA_OUT_COMPACT, dtab.p, ld_d, stream, bits)) != DA_OK) return rc;
    if (hy.take && (rc = hy_fixup(DA_OUT_COMPACT, dtab.p, ld_d, 0, TR_ALL, stream)) != DA_OK) return rc;
    DA_HIP_TRY(hipEventRecord(ev[3], stream));
    if ((rc = launch_expand_stream(dtab.as<uint16_t>(), ld_d, p.uidx, n, U, n_hash, d_out, ld, lists.p, stream, ev[4])) != DA_OK) return rc;
    DA_HIP_TRY(hipEventRecord(ev[5], stream));
    DA_HIP_TRY(hipStreamSynchronize(stream));
    route.taken = 4;
    route.chunks = route.expand_launches = 1;
    (void)hipEventElapsedTime(&route.ms[0], ev[0], ev[1]);
    (void)hipEventElapsedTime(&route.ms[1], ev[1], ev[2]);
    (void)hipEventElapsedTime(&route.ms[2], ev[2], ev[3]);
    (void)hipEventElapsedTime(&route.ms[3], ev[3], ev[4]);             // the copy lists (positions of every id, work items)
    (void)hipEventElapsedTime(&route.ms[4], ev[4], ev[5]);             // k_expand_stream
    return DA_OK;
  }
  if ((rc = ftab.alloc(expand_rows_workspace_bytes(n, U, DA_OUT_F64, false, n_hash, 0))) != DA_OK) return rc;
  // (the pipelined tile form leaves only diagonal / border tiles to launch_expand_unique: the output must qualify for the streaming passes)
  if (form != "tiles" && !ub.empty() && may_pipe && (ld & 1) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0) do {
    // PIPELINED form.  The table is compared band by band (1024 unique rows each, in order) by the persistent kernel on
    // a side stream with only `wg` workgroups per CU, so the rest of every CU stays free; the output row bands whose strings are all
    // numbered below the finished table rows are gathered (on `stream`) and expanded (on two alternating streams: chunk c + 1 fills
    // the CUs as chunk c drains) meanwhile.  The compare is VALU work, the expansion store work; unique ids are numbered by first
    // occurrence, so both sweep their triangles in the same direction and the compare stays ahead.  Same kernels, same bits.
    const int64_t KB = mh_sym_bands(U), NB = (int64_t)ub.size() - 1, ld_f = ceil_div(n, 8) * 8;
    int step = 4, wg = 2;
    if (da::config().mh_pipe_step > 0) step = da::config().mh_pipe_step;
    if (da::config().mh_pipe_wg > 0) wg = da::config().mh_pipe_wg;
    const bool alt = !da::config().mh_pipe_one_stream;
    struct Chunk { int64_t kb0, kb1, ob0, ob1; };
    std::vector<Chunk> chunks;
    for (int64_t kb = 0, ob = 0; kb < KB;) {
      const int64_t kb1 = std::min(KB, kb + (chunks.empty() ? 1 : step));   // a first chunk of one band: the expansion starts early
      int64_t ob1 = ob;
      if (kb1 == KB) ob1 = NB;
      else while (ob1 < NB && (int64_t)ub[(size_t)ob1 + 1] <= kb1 * 1024) ++ob1;   // rows below 1024 (ob1 + 1) only use ids < ub[ob1 + 1]
      chunks.push_back({kb, kb1, ob, ob1});
      kb = kb1; ob = ob1;
    }
    const size_t C = chunks.size();
    PipeRes *pr = pipe_acquire(5 * C);
    if (!pr) break;   // no side stream / events to be had: the one-stream form below needs none
    // leaves the side streams idle before the buffers declared above go back to the cache (error returns included)
    struct PipeGuard { PipeRes *r; ~PipeGuard() { (void)hipStreamSynchronize(r->side); (void)hipStreamSynchronize(r->alt[0]); (void)hipStreamSynchronize(r->alt[1]); pipe_release(r); } } pguard{pr};
    hipEvent_t *pe = pr->ev.data();                                  // per chunk: table rows done, gather begin / end, rows begin / end
    DA_HIP_TRY(hipStreamWaitEvent(pr->side, ev[2], 0));
    if ((rc = launch_mh_compare_edges_u16(planes.as<uint32_t>(), U, n_hash, dtab.as<uint16_t>(), ld_d, pr->side, bits)) != DA_OK) return rc;
    if (hy.take && (rc = hy_lists(pr->side)) != DA_OK) return rc;
    for (size_t c = 0; c < C; ++c) {
      const Chunk &ch = chunks[c];
      if ((rc = launch_mh_compare_bands_u16(planes.as<uint32_t>(), U, n_hash, dtab.as<uint16_t>(), ld_d, ch.kb0, ch.kb1, wg, pr->side, bits)) != DA_OK) return rc;
      if (hy.take && (rc = hy_fixup(DA_OUT_COMPACT, dtab.p, ld_d, ch.kb0 * 8, ch.kb1 * 8, pr->side)) != DA_OK) return rc;
      DA_HIP_TRY(hipEventRecord(pe[5 * c], pr->side));
      DA_HIP_TRY(hipStreamWaitEvent(stream, pe[5 * c], 0));
      DA_HIP_TRY(hipEventRecord(pe[5 * c + 1], stream));
      if ((rc = launch_gather_columns(dtab.as<uint16_t>(), ld_d, p.uidx, p.ufirst, n, U, ftab.as<uint16_t>(), ld_f, false, stream, 1, 0,
                                      ch.kb0 * 1024, std::min(U, ch.kb1 * 1024))) != DA_OK) return rc;
      DA_HIP_TRY(hipEventRecord(pe[5 * c + 2], stream));
      const hipStream_t es = alt ? pr->alt[c & 1] : stream;
      if (alt) DA_HIP_TRY(hipStreamWaitEvent(es, pe[5 * c + 2], 0));
      DA_HIP_TRY(hipEventRecord(pe[5 * c + 3], es));
      if ((rc = launch_expand_rows(ftab.as<uint16_t>(), ld_f, p.uidx, n, false, n_hash, 0, d_out, ld, ch.ob0, ch.ob1, es)) != DA_OK) return rc;
      DA_HIP_TRY(hipEventRecord(pe[5 * c + 4], es));
    }
    if (alt) for (size_t c = (C >= 2 ? C - 2 : 0); c < C; ++c) DA_HIP_TRY(hipStreamWaitEvent(stream, pe[5 * c + 4], 0));
    DA_HIP_TRY(hipEventRecord(ev[5], stream));
    if ((rc = launch_expand_unique(dtab.as<uint16_t>(), ld_d, p.uidx, n, DA_OUT_F64, false, n_hash, d_out, ld, stream, 0, ftab.as<uint16_t>(),
                                   p.ufirst, U, nullptr, nullptr, 1, 0, true)) != DA_OK) return rc;
    DA_HIP_TRY(hipEventRecord(ev[6], stream));
    DA_HIP_TRY(hipStreamSynchronize(stream));
    route.taken = 3;
    route.chunks = (int)C;
    for (const Chunk &ch : chunks) route.expand_launches += ch.ob1 > ch.ob0 ? 1 : 0;
    (void)hipEventElapsedTime(&route.ms[0], ev[0], ev[1]);
    (void)hipEventElapsedTime(&route.ms[1], ev[1], ev[2]);
    (void)hipEventElapsedTime(&route.ms[2], ev[2], pe[5 * (C - 1)]);       // the compare's span on the side stream (it overlaps what follows)
    float covered_to = 0;                                                  // ms[4]: the time some k_expand_rows launch was running (their
    for (size_t c = 0; c < C; ++c) {                                       // intervals overlap on the alternating streams: union, not sum)
      float g = 0, b = 0, e = 0;
      (void)hipEventElapsedTime(&g, pe[5 * c + 1], pe[5 * c + 2]);
      (void)hipEventElapsedTime(&b, ev[2], pe[5 * c + 3]);
      (void)hipEventElapsedTime(&e, ev[2], pe[5 * c + 4]);
      route.ms[3] += g;                                                    // gathers: summed (each includes its wait for free LDS beside the expansion)
      if (chunks[c].ob1 > chunks[c].ob0 && e > covered_to) { route.ms[4] += e - std::max(b, covered_to); covered_to = e; }
    }
    (void)hipEventElapsedTime(&route.ms[5], ev[5], ev[6]);
    return DA_OK;
  } while (0);
  if (hy.take && (rc = hy_lists(stream)) != DA_OK) return rc;
  if ((rc = launch_mh_compare(planes.as<uint32_t>(), U, n_hash, 0, U, true, DA_OUT_COMPACT, dtab.p, ld_d, stream, bits)) != DA_OK) return rc;
  if (hy.take && (rc = hy_fixup(DA_OUT_COMPACT, dtab.p, ld_d, 0, TR_ALL, stream)) != DA_OK) return rc;
  DA_HIP_TRY(hipEventRecord(ev[3], stream));
  if ((rc = launch_expand_unique(dtab.as<uint16_t>(), ld_d, p.uidx, n, DA_OUT_F64, false, n_hash, d_out, ld, stream, 0, ftab.as<uint16_t>(),
                                 p.ufirst, U, ev[4], ev[5])) != DA_OK) return rc;
  DA_HIP_TRY(hipEventRecord(ev[6], stream));
  DA_HIP_TRY(hipStreamSynchronize(stream));                         // the buffers go back to the parked-buffer cache here
  route.taken = 1;
  route.chunks = route.expand_launches = 1;
  (void)hipEventElapsedTime(&route.ms[0], ev[0], ev[1]);
  (void)hipEventElapsedTime(&route.ms[1], ev[1], ev[2]);
  (void)hipEventElapsedTime(&route.ms[2], ev[2], ev[3]);
  (void)hipEventElapsedTime(&route.ms[3], ev[3], ev[4]);
  (void)hipEventElapsedTime(&route.ms[4], ev[4], ev[5]);
  (void)hipEventElapsedTime(&route.ms[5], ev[5], ev[6]);
  return DA_OK;
}

int da_dev_similarity_mh(const uint8_t *d_residues, const int64_t *d_offsets, int64_t n, int64_t total_residues, int k, int n_hash,
                         const uint32_t *d_seeds, double *d_out, int64_t ld, void *stream) {
  int rc = validate_mh(n, k, n_hash);
  if (rc != DA_OK) return rc;
  if (!d_residues || !d_offsets || !d_seeds || !d_out) return fail(DA_ERR_BAD_ARG, "NULL device pointer");
  if (ld < n) return fail(DA_ERR_BAD_ARG, "ld (%lld) < n (%lld)", (long long)ld, (long long)n);
  if (n_hash > 65535) return fail(DA_ERR_UNSUPPORTED, "the compare kernel counts in 16 bits: n_hash <= 65535 (got %d); da_similarity_mh handles more", n_hash);
  return mh_full_symmetric(d_residues, d_offsets, n, total_residues, k, n_hash, d_seeds, d_out, ld, static_cast<hipStream_t>(stream));
}

int da_mh_last_route_chunks(int *chunks_out, int *expand_launches_out) {
  const MhRoute &r = mh_route();
  if (chunks_out) *chunks_out = r.chunks;
  if (expand_launches_out) *expand_launches_out = r.expand_launches;
  return DA_OK;
}

int da_mh_last_route_split(int64_t *rare_incidences_out, int *plane_bits_without_out) {
  const MhRoute &r = mh_route();
  if (rare_incidences_out) *rare_incidences_out = r.hybrid_pairs;
  if (plane_bits_without_out) *plane_bits_without_out = r.hybrid_bits_before;
  return DA_OK;
}

int da_mh_last_route(int64_t *n_out, int64_t *unique_out, int *dedup_taken_out, int *plane_bits_out, double *ms6_out) {
  const MhRoute &r = mh_route();
  if (n_out) *n_out = r.n;
  if (unique_out) *unique_out = r.unique;
  if (dedup_taken_out) *dedup_taken_out = r.taken;
  if (plane_bits_out) *plane_bits_out = r.plane_bits;
  if (ms6_out) for (int i = 0; i < 6; ++i) ms6_out[i] = r.ms[i];
  return DA_OK;
}

// ---- pieces of the duplicate-collapsing routes for callers that orchestrate the steps themselves (the one-process-per-GPU
// sharded drivers in dynaalign_amd/sharding.py: every rank builds the same plan, computes ITS shard of the unique table,
// all-gathers the shards -- 0.2x the bytes of the N x N shards at N = 100k -- and expands locally)
size_t da_dev_unique_plan_bytes(int64_t n, int64_t total_bytes) {
  if (n <= 0) return 256;
  return nw_dedup_workspace_bytes(n, total_bytes > 0 ? total_bytes : 1);
}

int da_dev_unique_plan(const uint8_t *d_bytes, const int64_t *d_offsets, int64_t n, int64_t total_bytes, void *d_work, size_t work_bytes,
                       da_unique_plan *plan, void *stream_v) {
  if (!plan || plan->struct_size < sizeof(da_unique_plan)) return fail(DA_ERR_BAD_ARG, "da_unique_plan: NULL or struct_size too small");
  if (n <= 0) return fail(DA_ERR_EMPTY_INPUT, "%s", da_status_message(DA_ERR_EMPTY_INPUT));
  if (n > 0x7ffffff0LL) return fail(DA_ERR_UNSUPPORTED, "unique plan: n too large");
  if (!d_bytes || !d_offsets || !d_work) return fail(DA_ERR_BAD_ARG, "NULL device pointer");
  if ((reinterpret_cast<uintptr_t>(d_work) & 255) || work_bytes < da_dev_unique_plan_bytes(n, total_bytes))
    return fail(DA_ERR_BAD_ARG, "workspace must be 256-byte aligned and hold da_dev_unique_plan_bytes(n, total_bytes) bytes");
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  int rc;
  const NwDedupPlan p = nw_dedup_layout(d_work, n, total_bytes > 0 ? total_bytes : 1);
  if ((rc = launch_nw_dedup_count(d_bytes, d_offsets, n, p, stream)) != DA_OK) return rc;
  int32_t M = 0, S = 0;
  DA_HIP_TRY(hipMemcpyAsync(&M, p.pm + n, 4, hipMemcpyDeviceToHost, stream));
  DA_HIP_TRY(hipMemcpyAsync(&S, p.ps + n, 4, hipMemcpyDeviceToHost, stream));
  DA_HIP_TRY(hipStreamSynchronize(stream));
  const int64_t U = (int64_t)M + S;
  if (U <= 0 || U > n) return fail(DA_ERR_HIP, "unique plan: inconsistent counts (%lld unique of %lld)", (long long)U, (long long)n);
  if ((rc = launch_nw_dedup_build(d_bytes, d_offsets, n, U, p, stream)) != DA_OK) return rc;
  plan->n = n; plan->unique = U;
  plan->d_uidx = p.uidx; plan->d_ufirst = p.ufirst; plan->d_ulast = p.ulast;
  plan->d_ubytes = p.ucodes; plan->d_uoffsets = p.uoff;
  plan->d_minfirst = p.minfirst; plan->d_maxlast = p.maxlast;
  return DA_OK;
}

int da_dev_shards_to_table(const void *d_gathered, int64_t ld_g, int64_t n, int world, int value_bits, uint16_t *d_table, int64_t ld_table,
                           void *stream) {
  if (n <= 0) return fail(DA_ERR_EMPTY_INPUT, "%s", da_status_message(DA_ERR_EMPTY_INPUT));
  if (!d_gathered || !d_table) return fail(DA_ERR_BAD_ARG, "NULL device pointer");
  if (world < 1 || ld_table < n) return fail(DA_ERR_BAD_ARG, "bad world / leading dimension");
  if (value_bits != 0 && (value_bits < 1 || value_bits > 16)) return fail(DA_ERR_BAD_ARG, "value_bits must be 0 (uint16 blocks) or 1..16");
  const ShardGeom g = shard_geom(n, world, 128);
  if (value_bits == 0 && ld_g < g.W) return fail(DA_ERR_BAD_ARG, "ld_g (%lld) < da_shard_ld (%lld)", (long long)ld_g, (long long)g.W);
  return launch_shards_to_table(d_gathered, ld_g, g, value_bits, d_table, ld_table, static_cast<hipStream_t>(stream));
}

size_t da_dev_expand_workspace_bytes(int64_t n, int64_t unique, int is_nw, int n_hash, int nw_max_len) {
  const size_t b = expand_rows_workspace_bytes(n, unique, DA_OUT_F64, is_nw != 0, n_hash, nw_max_len);
  return b ? b : 256;                                          // (shapes the two streaming passes do not cover use no workspace)
}

int da_dev_expand_unique(const uint16_t *d_table, int64_t ld_table, int table_world, const da_unique_plan *plan, int is_nw, int n_hash,
                         int nw_max_len, void *d_work, size_t work_bytes, double *d_out, int64_t ld, void *stream) {
  if (!plan || plan->struct_size < sizeof(da_unique_plan) || !plan->d_uidx || !plan->d_ufirst) return fail(DA_ERR_BAD_ARG, "bad da_unique_plan");
  if (!d_table || !d_out) return fail(DA_ERR_BAD_ARG, "NULL device pointer");
  const int64_t n = plan->n, U = plan->unique;
  if (n <= 0 || U <= 0 || U > n || ld < n || ld_table < U) return fail(DA_ERR_BAD_ARG, "bad sizes / leading dimensions");
  if (!is_nw && (n_hash <= 0 || n_hash > 65535)) return fail(DA_ERR_BAD_NHASH, "%s", da_status_message(DA_ERR_BAD_NHASH));
  if (table_world < 1) return fail(DA_ERR_BAD_ARG, "table_world must be >= 1");
  const int64_t rows_local = table_world > 1 ? ceil_div(ceil_div(U, 128), table_world) * 128 : 0;
  // MinHash (symmetric table, one block): the ROW expansion -- every output row written once from its table row in LDS, no gathered copy
  // (minhash_kernels.hip k_expand_stream; DYNAALIGN_EXPAND_NO_STREAM=1 keeps the tile passes).  The copy lists live in the caller's workspace.
  if (!is_nw && table_world == 1 && expand_stream_ok(n, U, n_hash, d_out, ld) && (ld_table & 7) == 0 && ld_table <= 65536 &&
      (reinterpret_cast<uintptr_t>(d_table) & 15) == 0) {
    const size_t lists_bytes = expand_stream_scratch_bytes(n, U);
    if (d_work && work_bytes >= lists_bytes && (reinterpret_cast<uintptr_t>(d_work) & 15) == 0)
      return launch_expand_stream(d_table, ld_table, plan->d_uidx, n, U, n_hash, d_out, ld, d_work, static_cast<hipStream_t>(stream));
    DevBuf lists;
    int rc = lists.alloc(lists_bytes);
    if (rc != DA_OK) return rc;
    rc = launch_expand_stream(d_table, ld_table, plan->d_uidx, n, U, n_hash, d_out, ld, lists.p, static_cast<hipStream_t>(stream));
    DA_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));   // the lists go away with this frame
    return rc;
  }
  const size_t need = expand_rows_workspace_bytes(n, U, DA_OUT_F64, is_nw != 0, n_hash, nw_max_len);
  uint16_t *d_F = (need && d_work && work_bytes >= need && (reinterpret_cast<uintptr_t>(d_work) & 15) == 0) ? static_cast<uint16_t *>(d_work) : nullptr;
  if ((ld_table & 7) || (reinterpret_cast<uintptr_t>(d_table) & 15)) d_F = nullptr;     // the column gather reads the table in 16-byte units
  return launch_expand_unique(d_table, ld_table, plan->d_uidx, n, DA_OUT_F64, is_nw != 0, n_hash, d_out, ld, static_cast<hipStream_t>(stream),
                              nw_max_len, d_F, plan->d_ufirst, U, nullptr, nullptr, table_world, rows_local);
}

size_t da_dev_unique_rows_bytes(int64_t n, int64_t unique) {
  if (n <= 0 || unique <= 0) return 256;
  return (size_t)unique * (size_t)(ceil_div(n, 8) * 8) * 2;
}

int da_dev_unique_rows(const uint16_t *d_table, int64_t ld_table, const da_unique_plan *plan, uint16_t *d_rows, void *stream) {
  if (!plan || plan->struct_size < sizeof(da_unique_plan) || !plan->d_uidx || !plan->d_ufirst) return fail(DA_ERR_BAD_ARG, "bad da_unique_plan");
  if (!d_table || !d_rows) return fail(DA_ERR_BAD_ARG, "NULL device pointer");
  if (ld_table < plan->unique) return fail(DA_ERR_BAD_ARG, "ld_table < unique");
  return launch_gather_columns(d_table, ld_table, plan->d_uidx, plan->d_ufirst, plan->n, plan->unique, d_rows, ceil_div(plan->n, 8) * 8, true,
                               static_cast<hipStream_t>(stream));
}

int da_dev_nw_unique_rows(const da_unique_plan *plan, int64_t max_len, int matrix_id, int gap_open, int gap_ext, int rank, int world,
                          uint16_t *d_out, int64_t ld, void *stream) {
  if (!plan || plan->struct_size < sizeof(da_unique_plan) || !plan->d_ubytes || !plan->d_uoffsets || !plan->d_ufirst || !plan->d_minfirst ||
      !plan->d_maxlast)
    return fail(DA_ERR_BAD_ARG, "bad da_unique_plan");
  if (!d_out) return fail(DA_ERR_BAD_ARG, "NULL device pointer");
  const int64_t U = plan->unique;
  if (world < 1 || rank < 0 || rank >= world || ld < U) return fail(DA_ERR_BAD_ARG, "bad rank / world / leading dimension");
  if (matrix_id < 0 || matrix_id >= matrix_count_host()) return fail(DA_ERR_BAD_MATRIX, "%s", da_status_message(DA_ERR_BAD_MATRIX));
  if (max_len < 1 || max_len > 64 || gap_open < 0 || gap_ext < 0)
    return fail(DA_ERR_UNSUPPORTED, "the ordered unique-table sweep serves sequences of 1..64 residues and penalties >= 0");
  // world = 1: the whole table in natural row order; world > 1: this rank's cyclic 128-row units, stored back to back -- one launch
  return launch_nw(plan->d_ubytes, plan->d_uoffsets, U, max_len, matrix_id, gap_open, gap_ext, 0, U, false, DA_OUT_COMPACT, d_out, ld, nullptr, 0,
                   static_cast<hipStream_t>(stream), rank, world > 1 ? world : 0, plan->d_ufirst, plan->d_minfirst, plan->d_maxlast);
}

int da_dev_nw_encode(const uint8_t *d_residues, int64_t total_residues, uint8_t *d_codes,
                     int32_t *d_bad, void *stream) {
  if (total_residues > 0 && (!d_residues || !d_codes || !d_bad)) return fail(DA_ERR_BAD_ARG, "NULL device pointer");
  return launch_nw_encode(d_residues, total_residues, d_codes, d_bad, static_cast<hipStream_t>(stream));
}

// similarityNW over the WHOLE pair space (symmetric mode) with byte-identical sequences collapsed first (nw_kernels.hip:
// "duplicate sequences"): plan -> (host reads the unique count: one stream synchronisation) -> ordered DP on the unique table
// -> index expansion to the n x n result.  Exact.  Taken when at least 15 % of the sequences are duplicates; otherwise, or
// for what the route does not cover (sequences > 64 residues, out-of-range penalties, score output, 32-bit packed output,
// n < 2048), the direct kernel runs.  Synchronises `stream` before returning when the route is taken (its buffers are parked
// for the next call).  DYNAALIGN_NW_NO_DEDUP=1 switches it off.
// what the last whole-matrix NW call of this thread did (da_nw_last_route: bench.py's roofline object needs the DP
// kernel's own duration, and the call is one C entry point)
struct NwRoute { int64_t n = 0, unique = 0; int taken = 0; float plan_ms = 0, dp_ms = 0, expand_ms = 0; };
static NwRoute &nw_route() { static thread_local NwRoute r; return r; }

static int nw_full_symmetric(const uint8_t *d_codes, const int64_t *d_offsets, int64_t n, int64_t total, int64_t max_len, int mid,
                             int gap_open, int gap_ext, int kind, void *d_out, int64_t ld, hipStream_t stream, int64_t *unique_out = nullptr) {
  int rc;
  if (unique_out) *unique_out = n;
  NwRoute &route = nw_route();
  route = NwRoute();
  route.n = route.unique = n;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  struct EvGuard { hipEvent_t *e; ~EvGuard() { for (int i = 0; i < 4; ++i) if (e[i]) (void)hipEventDestroy(e[i]); } } guard{ev};
  for (auto &e : ev) DA_HIP_TRY(hipEventCreate(&e));
  DA_HIP_TRY(hipEventRecord(ev[0], stream));
  int64_t min_n = 2048;
  if (da::config().nw_dedup_min_n >= 0) min_n = da::config().nw_dedup_min_n;   // DYNAALIGN_NW_DEDUP_MIN_N: tests lower it to reach the route with tiny inputs
  const bool eligible = n >= min_n && n <= 0x7ffffff0LL && max_len <= 64 && max_len >= 1 && (kind == DA_OUT_F64 || kind == DA_OUT_COMPACT) &&
                        gap_open >= 0 && gap_ext >= 0 && !da::config().nw_no_dedup;
  if (eligible) {
    DevBuf work;
    if ((rc = work.alloc(nw_dedup_workspace_bytes(n, total))) != DA_OK) return rc;
    const NwDedupPlan p = nw_dedup_layout(work.p, n, total);
    if ((rc = launch_nw_dedup_count(d_codes, d_offsets, n, p, stream)) != DA_OK) return rc;
    int32_t M = 0, S = 0;
    DA_HIP_TRY(hipMemcpyAsync(&M, p.pm + n, 4, hipMemcpyDeviceToHost, stream));
    DA_HIP_TRY(hipMemcpyAsync(&S, p.ps + n, 4, hipMemcpyDeviceToHost, stream));
    DA_HIP_TRY(hipStreamSynchronize(stream));
    const int64_t U = (int64_t)M + S;
    if (unique_out) *unique_out = U;
    route.unique = U;
    if (U > 0 && U * 100 <= n * 85) {
      if ((rc = launch_nw_dedup_build(d_codes, d_offsets, n, U, p, stream)) != DA_OK) return rc;
      const int64_t ld_d = (U + 7) / 8 * 8;
      DevBuf dtab;
      if ((rc = dtab.alloc((size_t)U * (size_t)ld_d * 2)) != DA_OK) return rc;
      // prefix sharing of the ordered DP (k_nw_short<.., PFX>): the unique strings' lexicographic order and neighbour prefixes, sequences of
      // up to 20 residues (0.1 ms; DYNAALIGN_NW_NO_PREFIX_SHARE=1: rows in id order, every row from scratch)
      DevBuf sortw;
      const int32_t *perm = nullptr;
      const uint8_t *lcp = nullptr;
      if (max_len <= 20 && !da::config().nw_no_prefix) {
        const size_t sb = nw_sort_unique_workspace_bytes(U);
        if (sortw.alloc(sb) == DA_OK) {
          if ((rc = launch_nw_sort_unique(p.ucodes, p.uoff, U, sortw.p, sb, &perm, &lcp, stream)) != DA_OK) return rc;
        } else (void)hipGetLastError();                     // (an optimisation: without its scratch the rows run in id order)
      }
      DA_HIP_TRY(hipEventRecord(ev[1], stream));
      if ((rc = launch_nw(p.ucodes, p.uoff, U, max_len, mid, gap_open, gap_ext, 0, U, false, DA_OUT_COMPACT, dtab.p, ld_d, nullptr, 0,
                          stream, 0, 0, p.ufirst, p.minfirst, p.maxlast, perm, lcp)) != DA_OK) return rc;
      DA_HIP_TRY(hipEventRecord(ev[2], stream));
      DevBuf ftab;                                        // column-gathered twin of the table for the two-pass expansion
      const size_t fbytes = expand_rows_workspace_bytes(n, U, kind, true, 0, (int)max_len);
      if (fbytes && (rc = ftab.alloc(fbytes)) != DA_OK) return rc;
      if ((rc = launch_expand_unique(dtab.as<uint16_t>(), ld_d, p.uidx, n, kind, true, 0, d_out, ld, stream, (int)max_len,
                                     fbytes ? ftab.as<uint16_t>() : nullptr, p.ufirst, U)) != DA_OK) return rc;
      DA_HIP_TRY(hipEventRecord(ev[3], stream));
      DA_HIP_TRY(hipStreamSynchronize(stream));          // work / dtab go back to the parked-buffer cache here
      route.taken = 1;
      (void)hipEventElapsedTime(&route.plan_ms, ev[0], ev[1]);
      (void)hipEventElapsedTime(&route.dp_ms, ev[1], ev[2]);
      (void)hipEventElapsedTime(&route.expand_ms, ev[2], ev[3]);
      return DA_OK;
    }
    DA_HIP_TRY(hipStreamSynchronize(stream));
  }
  DA_HIP_TRY(hipEventRecord(ev[1], stream));
  if ((rc = launch_nw(d_codes, d_offsets, n, max_len, mid, gap_open, gap_ext, 0, n, true, kind, d_out, ld, nullptr, 0, stream)) != DA_OK) return rc;
  DA_HIP_TRY(hipEventRecord(ev[2], stream));
  DA_HIP_TRY(hipEventSynchronize(ev[2]));                // (the direct route used to be asynchronous; the timing costs that)
  (void)hipEventElapsedTime(&route.plan_ms, ev[0], ev[1]);
  (void)hipEventElapsedTime(&route.dp_ms, ev[1], ev[2]);
  return DA_OK;
}

int da_nw_last_route(int64_t *n_out, int64_t *unique_out, int *dedup_taken_out, double *ms3_out) {
  const NwRoute &r = nw_route();
  if (n_out) *n_out = r.n;
  if (unique_out) *unique_out = r.unique;
  if (dedup_taken_out) *dedup_taken_out = r.taken;
  if (ms3_out) { ms3_out[0] = r.plan_ms; ms3_out[1] = r.dp_ms; ms3_out[2] = r.expand_ms; }
  return DA_OK;
}

int da_dev_nw(const uint8_t *d_codes, const int64_t *d_offsets, int64_t n, int64_t max_len,
              int matrix_id, int gap_open, int gap_ext, int64_t row_begin, int64_t row_end,
              int symmetric, int kind, void *d_out, int64_t ld, int32_t *d_score, int64_t ld_score,
              void *stream) {
  if (n <= 0) return DA_OK;
  if (!d_codes || !d_offsets || !d_out) return fail(DA_ERR_BAD_ARG, "NULL device pointer");
  if (row_begin < 0 || row_end > n || row_begin > row_end) return fail(DA_ERR_BAD_ARG, "bad row range");
  if (symmetric && (row_begin != 0 || row_end != n))
    return fail(DA_ERR_BAD_ARG, "symmetric mode needs the full row range");
  if (ld < n || (d_score && ld_score < n)) return fail(DA_ERR_BAD_ARG, "leading dimension < n");
  if (kind != DA_OUT_F64 && kind != DA_OUT_COMPACT && kind != DA_OUT_PACK32) return fail(DA_ERR_BAD_ARG, "bad output kind");
  if (symmetric && !d_score && kind != DA_OUT_PACK32) {   // whole pair space: duplicates are collapsed first when that pays
    int64_t total = 0;                                     // (bytes of all sequences: last offset; one small device read)
    DA_HIP_TRY(hipMemcpyAsync(&total, d_offsets + n, 8, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    DA_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return nw_full_symmetric(d_codes, d_offsets, n, total, max_len, matrix_id, gap_open, gap_ext, kind, d_out, ld,
                             static_cast<hipStream_t>(stream));
  }
  return launch_nw(d_codes, d_offsets, n, max_len, matrix_id, gap_open, gap_ext, row_begin, row_end,
                   symmetric != 0, kind, d_out, ld, d_score, ld_score, static_cast<hipStream_t>(stream));
}

int da_dev_symmetrize(void *d_mat, int64_t n, int64_t ld, int kind, void *stream) {
  if (!d_mat || ld < n) return fail(DA_ERR_BAD_ARG, "bad matrix / ld");
  return launch_symmetrize(d_mat, n, ld, kind, static_cast<hipStream_t>(stream));
}

int da_dev_widen(const uint16_t *d_in, double *d_out, int64_t count, int is_nw, int n_hash, void *stream) {
  if (count > 0 && (!d_in || !d_out)) return fail(DA_ERR_BAD_ARG, "NULL device pointer");
  return launch_widen(d_in, d_out, count, is_nw != 0, n_hash, static_cast<hipStream_t>(stream));
}

// ---- row-sharding over `world` ranks: rank p owns tile rows p, p+world, ... (cyclic, so
// the upper-triangular work is balanced); the unit is 128 rows for both kinds.
int64_t da_shard_rows(int64_t n, int world, int is_nw) {
  if (n <= 0 || world <= 0) return 0;
  return shard_geom(n, world, 128).rows;
}
int64_t da_shard_ld(int64_t n, int world, int is_nw) {
  if (n <= 0 || world <= 0) return 0;
  return shard_geom(n, world, 128).W;
}

int da_dev_mh_compare_shard(const uint32_t *d_planes, int plane_bits, int64_t n, int n_hash, int rank,
                            int world, uint16_t *d_local, int64_t ld, void *stream) {
  if (n <= 0) return fail(DA_ERR_EMPTY_INPUT, "%s", da_status_message(DA_ERR_EMPTY_INPUT));
  if (n_hash <= 0) return fail(DA_ERR_BAD_NHASH, "%s", da_status_message(DA_ERR_BAD_NHASH));
  if (n_hash > 65535) return fail(DA_ERR_UNSUPPORTED, "the compare kernel counts in 16 bits: n_hash <= 65535 (got %d)", n_hash);
  if (!d_planes || !d_local || world <= 0 || rank < 0 || rank >= world) return fail(DA_ERR_BAD_ARG, "bad shard arguments");
  const ShardGeom sg = shard_geom(n, world, 128);
  if (ld < sg.W) return fail(DA_ERR_BAD_ARG, "ld (%lld) < da_shard_ld (%lld)", (long long)ld, (long long)sg.W);
  if (reinterpret_cast<uintptr_t>(d_planes) & 15) return fail(DA_ERR_BAD_ARG, "bit-plane buffer must be 16-byte aligned");
  if (plane_bits != 8 && plane_bits != 12 && plane_bits != 14 && plane_bits != 15 && plane_bits != 16 && plane_bits != 32)
    return fail(DA_ERR_BAD_ARG, "plane_bits must be 8, 12, 14, 15, 16 or 32 (got %d)", plane_bits);
  if ((int64_t)rank * 128 >= n) return DA_OK;  // this rank owns no rows
  return launch_mh_compare(d_planes, n, n_hash, (int64_t)rank * 128, n, false, DA_OUT_COMPACT, d_local, ld,
                           static_cast<hipStream_t>(stream), plane_bits, world, true, sg.Q, sg.W);
}

// ---- the MH exchange in value_bits = bits(n_hash) instead of 16 bits per count
static int packed_args_ok(int64_t n, int world, int value_bits) {
  if (n <= 0 || world <= 0) return fail(DA_ERR_BAD_ARG, "bad shard arguments");
  if (value_bits < 1 || value_bits > 16) return fail(DA_ERR_BAD_ARG, "value_bits must be in [1, 16] (got %d)", value_bits);
  return DA_OK;
}
int64_t da_shard_packed_bytes(int64_t n, int world, int value_bits) {
  if (n <= 0 || world <= 0 || value_bits < 1 || value_bits > 16) return 0;
  return shard_packed_bytes(shard_geom(n, world, 128), value_bits);
}
int da_dev_pack_shard(const uint16_t *d_local, int64_t ld, int64_t n, int world, int value_bits, uint8_t *d_packed,
                      void *stream) {
  int rc = packed_args_ok(n, world, value_bits);
  if (rc != DA_OK) return rc;
  const ShardGeom sg = shard_geom(n, world, 128);
  if (!d_local || !d_packed || ld < sg.W) return fail(DA_ERR_BAD_ARG, "bad block / ld");
  if (reinterpret_cast<uintptr_t>(d_packed) & 7) return fail(DA_ERR_BAD_ARG, "packed block must be 8-byte aligned");
  return launch_pack_shard(d_local, ld, sg, value_bits, d_packed, static_cast<hipStream_t>(stream));
}
int da_dev_finalize_shards_packed(const uint8_t *d_gathered, int64_t n, int world, int value_bits, int n_hash,
                                  double *d_out, int64_t ld_out, void *stream) {
  int rc = packed_args_ok(n, world, value_bits);
  if (rc != DA_OK) return rc;
  if (!d_gathered || !d_out || ld_out < n) return fail(DA_ERR_BAD_ARG, "bad arguments");
  if (n_hash <= 0 || n_hash >= (1 << value_bits)) return fail(DA_ERR_BAD_ARG, "n_hash (%d) does not fit value_bits (%d)", n_hash, value_bits);
  return launch_finalize_packed(d_gathered, shard_geom(n, world, 128), value_bits, n_hash, d_out, ld_out,
                                static_cast<hipStream_t>(stream));
}

int da_dev_nw_shard(const uint8_t *d_codes, const int64_t *d_offsets, int64_t n, int64_t max_len, int matrix_id,
                    int gap_open, int gap_ext, int rank, int world, uint16_t *d_local, int64_t ld, void *stream) {
  if (n <= 0) return DA_OK;
  if (!d_codes || !d_offsets || !d_local || world <= 0 || rank < 0 || rank >= world)
    return fail(DA_ERR_BAD_ARG, "bad shard arguments");
  if (ld < shard_geom(n, world, 128).W) return fail(DA_ERR_BAD_ARG, "ld < da_shard_ld");
  return launch_nw(d_codes, d_offsets, n, max_len, matrix_id, gap_open, gap_ext, 0, n, false, DA_OUT_COMPACT, d_local, ld,
                   nullptr, 0, static_cast<hipStream_t>(stream), rank, world);
}

int da_dev_finalize_shards(const uint16_t *d_gathered, int64_t ld_g, int64_t n, int world, int is_nw, int n_hash,
                           double *d_out, int64_t ld_out, void *stream) {
  if (n <= 0) return DA_OK;
  if (!d_gathered || !d_out || world <= 0 || ld_out < n) return fail(DA_ERR_BAD_ARG, "bad finalize arguments");
  const ShardGeom sg = shard_geom(n, world, 128);
  if (ld_g < sg.W) return fail(DA_ERR_BAD_ARG, "ld_g < da_shard_ld");
  return launch_finalize_sharded(d_gathered, ld_g, sg, is_nw != 0, n_hash, d_out, ld_out, static_cast<hipStream_t>(stream));
}


// ---- threshold + sparsify: the step clusterbreak applies right after sim_fn -------------

int da_dev_upper_histogram(const uint16_t *d_compact, int64_t ld, int64_t n, int nbins, uint64_t *d_hist, void *stream) {
  if (n <= 0) return DA_OK;
  if (!d_compact || !d_hist || ld < n || nbins <= 0 || nbins > 65536) return fail(DA_ERR_BAD_ARG, "bad histogram arguments");
  return launch_upper_histogram(d_compact, ld, n, nbins, reinterpret_cast<unsigned long long *>(d_hist),
                                static_cast<hipStream_t>(stream));
}

// the same two steps reading the n x n count matrix THROUGH a row map (row i = row d_rowmap[i] of d_rows: da_dev_unique_rows'
// table with the plan's d_uidx) -- the duplicate rows are never materialised
int da_dev_upper_histogram_rows(const uint16_t *d_rows, int64_t ld, const int32_t *d_rowmap, int64_t n, int nbins, uint64_t *d_hist,
                                void *stream) {
  if (n <= 0) return DA_OK;
  if (!d_rows || !d_rowmap || !d_hist || ld < n || nbins <= 0 || nbins > 65536) return fail(DA_ERR_BAD_ARG, "bad histogram arguments");
  return launch_upper_histogram(d_rows, ld, n, nbins, reinterpret_cast<unsigned long long *>(d_hist), static_cast<hipStream_t>(stream), 0, 0,
                                d_rowmap);
}

int da_dev_extract_edges_rows(const uint16_t *d_rows, int64_t ld, const int32_t *d_rowmap, int64_t n, const uint8_t *d_keep, int nbins,
                              int include_diagonal, int32_t *d_i, int32_t *d_j, uint16_t *d_v, int64_t capacity, uint64_t *d_count,
                              void *stream) {
  if (n <= 0) return DA_OK;
  if (!d_rows || !d_rowmap || !d_keep || !d_i || !d_j || !d_v || !d_count || ld < n || nbins <= 0 || nbins > 65536 || capacity < 0)
    return fail(DA_ERR_BAD_ARG, "bad edge-extraction arguments");
  if (n > 0x7fffffffLL) return fail(DA_ERR_UNSUPPORTED, "edge indices are int32");
  return launch_extract_edges(d_rows, ld, n, d_keep, nbins, include_diagonal != 0, d_i, d_j, d_v, capacity,
                              reinterpret_cast<unsigned long long *>(d_count), static_cast<hipStream_t>(stream), 0, 0, d_rowmap);
}

// The same two steps on ONE RANK'S folded shard block (da_dev_mh_compare_shard output): every unordered
// pair lives on exactly one rank, so summing the ranks' histograms (an all-reduce of n_hash+1 words) gives
// the global histogram, and the ranks' edge lists are disjoint -- no N x N exchange at all.
int da_dev_shard_histogram(const uint16_t *d_local, int64_t ld, int64_t n, int rank, int world, int nbins,
                           uint64_t *d_hist, void *stream) {
  if (n <= 0) return DA_OK;
  if (!d_local || !d_hist || world <= 0 || rank < 0 || rank >= world || nbins <= 0 || nbins > 65536 ||
      ld < shard_geom(n, world, 128).W)
    return fail(DA_ERR_BAD_ARG, "bad shard histogram arguments");
  return launch_upper_histogram(d_local, ld, n, nbins, reinterpret_cast<unsigned long long *>(d_hist),
                                static_cast<hipStream_t>(stream), rank, world);
}

int da_dev_shard_extract_edges(const uint16_t *d_local, int64_t ld, int64_t n, int rank, int world,
                               const uint8_t *d_keep, int nbins, int include_diagonal, int32_t *d_i, int32_t *d_j,
                               uint16_t *d_v, int64_t capacity, uint64_t *d_count, void *stream) {
  if (n <= 0) return DA_OK;
  if (!d_local || !d_keep || !d_i || !d_j || !d_v || !d_count || world <= 0 || rank < 0 || rank >= world || nbins <= 0 ||
      nbins > 65536 || capacity < 0 || ld < shard_geom(n, world, 128).W)
    return fail(DA_ERR_BAD_ARG, "bad shard edge-extraction arguments");
  if (n > 0x7fffffffLL) return fail(DA_ERR_UNSUPPORTED, "edge indices are int32");
  return launch_extract_edges(d_local, ld, n, d_keep, nbins, include_diagonal != 0, d_i, d_j, d_v, capacity,
                              reinterpret_cast<unsigned long long *>(d_count), static_cast<hipStream_t>(stream), rank, world);
}

int da_dev_extract_edges(const uint16_t *d_compact, int64_t ld, int64_t n, const uint8_t *d_keep, int nbins,
                         int include_diagonal, int32_t *d_i, int32_t *d_j, uint16_t *d_v, int64_t capacity,
                         uint64_t *d_count, void *stream) {
  if (n <= 0) return DA_OK;
  if (!d_compact || !d_keep || !d_i || !d_j || !d_v || !d_count || ld < n || nbins <= 0 || nbins > 65536 || capacity < 0)
    return fail(DA_ERR_BAD_ARG, "bad edge-extraction arguments");
  if (n > 0x7fffffffLL) return fail(DA_ERR_UNSUPPORTED, "edge indices are int32");
  return launch_extract_edges(d_compact, ld, n, d_keep, nbins, include_diagonal != 0, d_i, d_j, d_v, capacity,
                              reinterpret_cast<unsigned long long *>(d_count), static_cast<hipStream_t>(stream));
}

// the device edge list (da_dev_extract_edges[_rows]: i <= j, codes) as the symmetric CSR da_louvain_csr clusters -- sorted on the
// device (one radix sort of 2 m keys), so the host neither sorts 1.45e8 edges nor carries 16 bytes per edge
size_t da_dev_edges_to_csr_bytes(int64_t n_edges, int64_t n) { return edges_to_csr_workspace_bytes(n_edges, n); }

int da_dev_edges_to_csr(const int32_t *d_i, const int32_t *d_j, const uint16_t *d_v, int64_t n_edges, int64_t n, void *d_work,
                        size_t work_bytes, int64_t *d_ptr, int32_t *d_adj, uint16_t *d_codes, uint16_t *d_loops, void *stream) {
  if (n <= 0) return fail(DA_ERR_EMPTY_INPUT, "%s", da_status_message(DA_ERR_EMPTY_INPUT));
  if (n_edges < 0 || !d_ptr || !d_loops || (n_edges > 0 && (!d_i || !d_j || !d_v || !d_adj || !d_codes || !d_work)))
    return fail(DA_ERR_BAD_ARG, "bad edges -> CSR arguments");
  if (n_edges > 0 && (reinterpret_cast<uintptr_t>(d_work) & 255)) return fail(DA_ERR_BAD_ARG, "workspace must be 256-byte aligned");
  return launch_edges_to_csr(d_i, d_j, d_v, n_edges, n, d_work, work_bytes, d_ptr, d_adj, d_codes, d_loops, static_cast<hipStream_t>(stream));
}

// R's quantile(x, p, type = 7) (stats::quantile.default, the default the reference's
// R/clusterbreak.R:219 uses) of the multiset {values[b] x hist[b]}, values ascending:
//   index = 1 + (N-1) p; lo = floor(index); hi = ceiling(index); q = x[lo];
//   if (index > lo && x[hi] != q) q = (1-h) q + h x[hi],  h = index - lo
int da_quantile_type7(const uint64_t *hist, const double *values, int nbins, double p, double *q_out) {
  if (!hist || !values || !q_out || nbins <= 0 || !(p >= 0.0 && p <= 1.0)) return fail(DA_ERR_BAD_ARG, "bad quantile arguments");
  uint64_t total = 0;
  for (int b = 0; b < nbins; ++b) total += hist[b];
  if (total == 0) return fail(DA_ERR_BAD_ARG, "quantile of an empty set");
  const double index = 1.0 + (double)(total - 1) * p;
  const double lo = std::floor(index), hi = std::ceil(index);
  auto at = [&](double pos1) {  // pos1: 1-based rank
    const uint64_t r = (uint64_t)pos1;
    uint64_t cum = 0;
    for (int b = 0; b < nbins; ++b) {
      cum += hist[b];
      if (r <= cum) return values[b];
    }
    return values[nbins - 1];
  };
  double q = at(lo);
  const double xh = at(hi);
  if (index > lo && xh != q) {
    const double h = index - lo;
    q = (1.0 - h) * q + h * xh;
  }
  *q_out = q;
  return DA_OK;
}

static int nw_validate(const uint8_t *residues, const int64_t *offsets, int64_t n);

// Shared tail of the *_edges entry points: d_cnt is the dense n x n uint16 code matrix, d_hist the
// histogram of its strict upper triangle (nbins codes), values[b] the similarity a code stands for.
//   threshold <- quantile(S[upper.tri(S)], thresh_p)   (R type 7, R/clusterbreak.R:219)
//   S[S < threshold] <- 0; a zero weight is no edge     (:221; igraph, weighted = TRUE)
// The diagonal (1.0) always survives.  Edges come back sorted by (i, j).
struct EdgeSet {
  double threshold = 0.0;
  std::vector<int32_t> i, j;
  std::vector<double> w;
};

static int edges_from_counts(const uint16_t *d_cnt, int64_t n, int nbins, const unsigned long long *d_hist,
                             const std::vector<double> &values, double thresh_p, bool want_edges, EdgeSet &es,
                             int64_t *n_edges_out) {
  int rc;
  std::vector<uint64_t> h(nbins);
  DA_HIP_TRY(hipMemcpy(h.data(), d_hist, (size_t)nbins * 8, hipMemcpyDeviceToHost));
  // order statistics need the codes in ascending order of their VALUE (MH: already so; NW: ratios)
  std::vector<int> order(nbins);
  for (int b = 0; b < nbins; ++b) order[b] = b;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return values[a] < values[b]; });
  std::vector<uint64_t> hs(nbins);
  std::vector<double> vs(nbins);
  for (int b = 0; b < nbins; ++b) { hs[b] = h[order[b]]; vs[b] = values[order[b]]; }
  double thr;
  if ((rc = da_quantile_type7(hs.data(), vs.data(), nbins, thresh_p, &thr)) != DA_OK) return rc;
  es.threshold = thr;
  std::vector<uint8_t> kp(nbins);
  int64_t n_edges = n;                                                       // the diagonal (1.0) always survives
  for (int b = 0; b < nbins; ++b) {
    kp[b] = (values[b] != 0.0 && !(values[b] < thr)) ? 1 : 0;
    if (kp[b]) n_edges += (int64_t)h[b];
  }
  *n_edges_out = n_edges;
  if (!want_edges || n_edges == 0) return DA_OK;
  DevBuf keep, cnt_edges, di, dj, dv;
  if ((rc = keep.alloc((size_t)nbins)) != DA_OK) return rc;
  if ((rc = cnt_edges.alloc(8)) != DA_OK) return rc;
  if ((rc = di.alloc((size_t)n_edges * 4)) != DA_OK) return rc;
  if ((rc = dj.alloc((size_t)n_edges * 4)) != DA_OK) return rc;
  if ((rc = dv.alloc((size_t)n_edges * 2)) != DA_OK) return rc;
  DA_HIP_TRY(hipMemcpy(keep.p, kp.data(), (size_t)nbins, hipMemcpyHostToDevice));
  DA_HIP_TRY(hipMemset(cnt_edges.p, 0, 8));
  if ((rc = launch_extract_edges(d_cnt, n, n, keep.as<uint8_t>(), nbins, true, di.as<int32_t>(), dj.as<int32_t>(),
                                 dv.as<uint16_t>(), n_edges, cnt_edges.as<unsigned long long>(), nullptr)) != DA_OK) return rc;
  uint64_t got = 0;
  DA_HIP_TRY(hipMemcpy(&got, cnt_edges.p, 8, hipMemcpyDeviceToHost));
  if ((int64_t)got != n_edges) return fail(DA_ERR_HIP, "edge count mismatch: histogram says %lld, extraction found %llu",
                                           (long long)n_edges, (unsigned long long)got);
  std::vector<int32_t> hi_(n_edges), hj_(n_edges);
  std::vector<uint16_t> hv_(n_edges);
  DA_HIP_TRY(hipMemcpy(hi_.data(), di.p, (size_t)n_edges * 4, hipMemcpyDeviceToHost));
  DA_HIP_TRY(hipMemcpy(hj_.data(), dj.p, (size_t)n_edges * 4, hipMemcpyDeviceToHost));
  DA_HIP_TRY(hipMemcpy(hv_.data(), dv.p, (size_t)n_edges * 2, hipMemcpyDeviceToHost));
  // the device appends in arrival order; hand the edges back sorted by (i, j): a counting sort by row (1.45e8 edges at
  // N = 100k: a comparison sort of the whole list took ~20 s on one thread), then every row's columns are sorted -- rows are
  // independent, a few host threads share them; (i, j) pairs are unique, so the order is fully determined
  std::vector<int64_t> rptr((size_t)n + 1, 0);
  for (int64_t e = 0; e < n_edges; ++e) ++rptr[(size_t)hi_[(size_t)e] + 1];
  for (int64_t v = 0; v < n; ++v) rptr[(size_t)v + 1] += rptr[(size_t)v];
  std::vector<int32_t> sj((size_t)n_edges);
  std::vector<uint16_t> sv((size_t)n_edges);
  {
    std::vector<int64_t> cur(rptr.begin(), rptr.end() - 1);
    for (int64_t e = 0; e < n_edges; ++e) {
      const int64_t pos = cur[(size_t)hi_[(size_t)e]]++;
      sj[(size_t)pos] = hj_[(size_t)e];
      sv[(size_t)pos] = hv_[(size_t)e];
    }
  }
  hj_.clear(); hj_.shrink_to_fit(); hv_.clear(); hv_.shrink_to_fit();
  es.i.resize(n_edges); es.j.resize(n_edges); es.w.resize(n_edges);
  {
    unsigned nt = std::thread::hardware_concurrency();
    nt = std::max(1u, std::min(16u, nt ? nt : 1u));
    if (n_edges < 1000000) nt = 1;
    std::vector<std::thread> th;
    auto work = [&](unsigned t) {
      std::vector<std::pair<int32_t, uint16_t>> tmp;
      const int64_t e_lo = n_edges * t / nt, e_hi = n_edges * (t + 1) / nt;      // rows cut at roughly equal edge counts
      int64_t v0 = t == 0 ? 0 : std::lower_bound(rptr.begin(), rptr.end() - 1, e_lo) - rptr.begin();
      int64_t v1 = t + 1 == nt ? n : std::lower_bound(rptr.begin(), rptr.end() - 1, e_hi) - rptr.begin();
      for (int64_t v = v0; v < v1; ++v) {
        const int64_t b = rptr[(size_t)v], e = rptr[(size_t)v + 1];
        tmp.clear();
        for (int64_t q = b; q < e; ++q) tmp.emplace_back(sj[(size_t)q], sv[(size_t)q]);
        std::sort(tmp.begin(), tmp.end());
        for (int64_t q = b; q < e; ++q) {
          es.i[(size_t)q] = (int32_t)v;
          es.j[(size_t)q] = tmp[(size_t)(q - b)].first;
          es.w[(size_t)q] = values[tmp[(size_t)(q - b)].second];
        }
      }
    };
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
  }
  return DA_OK;
}

// legacy calling convention of the *_edges entry points on top of an EdgeSet: size query (all buffers NULL) or fill
static int edges_deliver(const EdgeSet &es, int64_t n_edges, double *threshold_out, int64_t *n_edges_out, int64_t capacity,
                         int32_t *ei, int32_t *ej, double *ew) {
  *threshold_out = es.threshold;
  *n_edges_out = n_edges;
  if (!ei && !ej && !ew) return DA_OK;
  if (capacity < n_edges) return fail(DA_ERR_BAD_ARG, "edge buffers hold %lld entries, %lld needed", (long long)capacity, (long long)n_edges);
  if (n_edges) {
    memcpy(ei, es.i.data(), (size_t)n_edges * 4);
    memcpy(ej, es.j.data(), (size_t)n_edges * 4);
    memcpy(ew, es.w.data(), (size_t)n_edges * 8);
  }
  return DA_OK;
}

// similarityNW + clusterbreak's threshold step as an edge list.  The uint16 code (matches << 8 | length)
// takes few distinct values, so the same histogram / quantile / extraction path as for MinHash applies;
// value(code) = matches / length with the reference's divide (src/pairwiseSeqAlign.cpp:311).
static int nw_edges_core(const uint8_t *residues, const int64_t *offsets, int64_t n, const char *matrix_name,
                         int gap_open, int gap_ext, double thresh_p, bool want_edges, EdgeSet &es, int64_t *n_edges_out) {
  const int mid = da_matrix_id(matrix_name);  // reference :338 -> :190-206, before anything else
  if (mid < 0) return DA_ERR_BAD_MATRIX;
  if (!residues) return fail(DA_ERR_BAD_ARG, "NULL pointer");
  if (n < 2) return fail(DA_ERR_BAD_ARG, "the threshold is a quantile of the strict upper triangle: need >= 2 sequences");
  if (!(thresh_p >= 0.0 && thresh_p <= 1.0)) return fail(DA_ERR_BAD_ARG, "thresh_p must be in [0, 1]");
  int64_t total, max_len;
  int rc;
  if ((rc = check_offsets(offsets, n, &total, &max_len)) != DA_OK) return rc;
  if ((rc = nw_validate(residues, offsets, n)) != DA_OK) return rc;
  for (int64_t i = 0; i < n; ++i)
    if (offsets[i + 1] == offsets[i])
      return fail(DA_ERR_UNSUPPORTED, "sequence %lld is empty: its similarities are 0/0 = NaN and R's quantile() refuses NaN",
                  (long long)(i + 1));
  if (max_len > 127) return fail(DA_ERR_UNSUPPORTED, "the NW edge list works on uint16 codes: sequences up to 127 residues");
  if ((rc = require_device()) != DA_OK) return rc;
  DeviceInput in;
  if ((rc = in.upload(residues, offsets, n, total, nullptr, 0)) != DA_OK) return rc;
  DevBuf codes, bad, cnt, hist;
  if ((rc = codes.alloc((size_t)total)) != DA_OK) return rc;
  if ((rc = bad.alloc(sizeof(int32_t))) != DA_OK) return rc;
  DA_HIP_TRY(hipMemset(bad.p, 0, sizeof(int32_t)));
  if ((rc = launch_nw_encode(in.res.as<uint8_t>(), total, codes.as<uint8_t>(), bad.as<int32_t>(), nullptr)) != DA_OK)
    return rc;
  if ((rc = cnt.alloc((size_t)n * (size_t)n * 2)) != DA_OK) return rc;      // uint16 codes stay on the device
  const int nbins = (int)((max_len << 8) | (2 * max_len)) + 1;              // matches <= max_len, length <= 2 * max_len
  if ((rc = hist.alloc((size_t)nbins * 8)) != DA_OK) return rc;
  DA_HIP_TRY(hipMemset(hist.p, 0, (size_t)nbins * 8));
  if ((rc = nw_full_symmetric(codes.as<uint8_t>(), in.off.as<int64_t>(), n, total, max_len, mid, gap_open, gap_ext, DA_OUT_COMPACT,
                              cnt.p, n, nullptr)) != DA_OK) return rc;
  if ((rc = launch_upper_histogram(cnt.as<uint16_t>(), n, n, nbins, hist.as<unsigned long long>(), nullptr)) != DA_OK) return rc;
  std::vector<double> values(nbins);
  for (int b = 0; b < nbins; ++b) {
    const int ln = b & 255;
    values[b] = ln ? (double)(b >> 8) / (double)ln : 0.0;                   // length 0 cannot occur (no empty sequences)
  }
  return edges_from_counts(cnt.as<uint16_t>(), n, nbins, hist.as<unsigned long long>(), values, thresh_p, want_edges, es, n_edges_out);
}

static int mh_edges_core(const uint8_t *residues, const int64_t *offsets, int64_t n, int k, int n_hash,
                         const uint32_t *seeds, double thresh_p, bool want_edges, EdgeSet &es, int64_t *n_edges_out) {
  int rc = validate_mh(n, k, n_hash);
  if (rc != DA_OK) return rc;
  if (!residues || !seeds) return fail(DA_ERR_BAD_ARG, "NULL pointer");
  if (n < 2) return fail(DA_ERR_BAD_ARG, "the threshold is a quantile of the strict upper triangle: need >= 2 sequences");
  if (!(thresh_p >= 0.0 && thresh_p <= 1.0)) return fail(DA_ERR_BAD_ARG, "thresh_p must be in [0, 1]");
  if (n_hash > 65535) return fail(DA_ERR_UNSUPPORTED, "the compare kernel counts in 16 bits: n_hash <= 65535 (got %d)", n_hash);
  int64_t total, max_len;
  if ((rc = check_offsets(offsets, n, &total, &max_len)) != DA_OK) return rc;
  if ((rc = require_device()) != DA_OK) return rc;
  DeviceInput in;
  if ((rc = in.upload(residues, offsets, n, total, seeds, n_hash)) != DA_OK) return rc;
  const int64_t lds = sig_ld_for(n_hash);
  DevBuf sig, planes, cnt, hist;
  if ((rc = sig.alloc((size_t)n * lds * 4)) != DA_OK) return rc;
  if ((rc = planes.alloc((size_t)mh_planes_words(n, n_hash) * 4)) != DA_OK) return rc;
  if ((rc = cnt.alloc((size_t)n * (size_t)n * 2)) != DA_OK) return rc;   // uint16 counts stay on the device
  const int nbins = n_hash + 1;
  if ((rc = hist.alloc((size_t)nbins * 8)) != DA_OK) return rc;
  DA_HIP_TRY(hipMemset(hist.p, 0, (size_t)nbins * 8));
  if ((rc = launch_minhash_signatures(in.res.as<uint8_t>(), in.off.as<int64_t>(), n, k, n_hash, in.seeds.as<uint32_t>(),
                                      sig.as<uint32_t>(), lds, nullptr)) != DA_OK) return rc;
  int bits = 32;
  {
    DevBuf work;
    const size_t wb = mh_planes_workspace_bytes(n, n_hash);
    if ((rc = work.alloc(wb)) != DA_OK) return rc;
    if ((rc = build_planes(sig.as<uint32_t>(), lds, n, n_hash, 0, work.p, wb, planes.as<uint32_t>(), &bits, nullptr)) != DA_OK)
      return rc;
    DA_HIP_TRY(hipStreamSynchronize(nullptr));   // the workspace is released here
  }
  if ((rc = launch_mh_compare(planes.as<uint32_t>(), n, n_hash, 0, n, true, DA_OUT_COMPACT, cnt.p, n, nullptr, bits)) != DA_OK)
    return rc;
  if ((rc = launch_upper_histogram(cnt.as<uint16_t>(), n, n, nbins, hist.as<unsigned long long>(), nullptr)) != DA_OK) return rc;
  std::vector<double> values(nbins);
  for (int b = 0; b < nbins; ++b) values[b] = (double)b / n_hash;          // src/minHash.cpp:174
  return edges_from_counts(cnt.as<uint16_t>(), n, nbins, hist.as<unsigned long long>(), values, thresh_p, want_edges, es, n_edges_out);
}

// ---- public forms.  da_similarity_*_edges: size query (buffers NULL) or fill -- each call runs the whole pipeline.
// da_similarity_*_edges_begin / da_edges_fetch / da_edges_free: ONE pass; the result waits in a handle until fetched.
struct da_edges { EdgeSet es; int64_t n_edges = 0; };

int da_similarity_nw_edges(const uint8_t *residues, const int64_t *offsets, int64_t n, const char *matrix_name,
                           int gap_open, int gap_ext, double thresh_p, double *threshold_out, int64_t *n_edges_out,
                           int64_t capacity, int32_t *ei, int32_t *ej, double *ew) {
  if (!threshold_out || !n_edges_out) return fail(DA_ERR_BAD_ARG, "NULL pointer");
  if ((ei || ej || ew) && (!ei || !ej || !ew)) return fail(DA_ERR_BAD_ARG, "NULL edge buffer");
  EdgeSet es;
  int64_t m = 0;
  int rc = nw_edges_core(residues, offsets, n, matrix_name, gap_open, gap_ext, thresh_p, ei != nullptr, es, &m);
  if (rc != DA_OK) return rc;
  return edges_deliver(es, m, threshold_out, n_edges_out, capacity, ei, ej, ew);
}

int da_similarity_mh_edges(const uint8_t *residues, const int64_t *offsets, int64_t n, int k, int n_hash,
                           const uint32_t *seeds, double thresh_p, double *threshold_out, int64_t *n_edges_out,
                           int64_t capacity, int32_t *ei, int32_t *ej, double *ew) {
  if (!threshold_out || !n_edges_out) return fail(DA_ERR_BAD_ARG, "NULL pointer");
  if ((ei || ej || ew) && (!ei || !ej || !ew)) return fail(DA_ERR_BAD_ARG, "NULL edge buffer");
  EdgeSet es;
  int64_t m = 0;
  int rc = mh_edges_core(residues, offsets, n, k, n_hash, seeds, thresh_p, ei != nullptr, es, &m);
  if (rc != DA_OK) return rc;
  return edges_deliver(es, m, threshold_out, n_edges_out, capacity, ei, ej, ew);
}

int da_similarity_mh_edges_begin(const uint8_t *residues, const int64_t *offsets, int64_t n, int k, int n_hash,
                                 const uint32_t *seeds, double thresh_p, da_edges **handle_out, double *threshold_out,
                                 int64_t *n_edges_out) {
  if (!handle_out || !threshold_out || !n_edges_out) return fail(DA_ERR_BAD_ARG, "NULL pointer");
  *handle_out = nullptr;
  std::unique_ptr<da_edges> h(new da_edges);
  int rc = mh_edges_core(residues, offsets, n, k, n_hash, seeds, thresh_p, true, h->es, &h->n_edges);
  if (rc != DA_OK) return rc;
  *threshold_out = h->es.threshold;
  *n_edges_out = h->n_edges;
  *handle_out = h.release();
  return DA_OK;
}

int da_similarity_nw_edges_begin(const uint8_t *residues, const int64_t *offsets, int64_t n, const char *matrix_name,
                                 int gap_open, int gap_ext, double thresh_p, da_edges **handle_out, double *threshold_out,
                                 int64_t *n_edges_out) {
  if (!handle_out || !threshold_out || !n_edges_out) return fail(DA_ERR_BAD_ARG, "NULL pointer");
  *handle_out = nullptr;
  std::unique_ptr<da_edges> h(new da_edges);
  int rc = nw_edges_core(residues, offsets, n, matrix_name, gap_open, gap_ext, thresh_p, true, h->es, &h->n_edges);
  if (rc != DA_OK) return rc;
  *threshold_out = h->es.threshold;
  *n_edges_out = h->n_edges;
  *handle_out = h.release();
  return DA_OK;
}

int da_edges_fetch(const da_edges *h, int64_t capacity, int32_t *ei, int32_t *ej, double *ew) {
  if (!h || !ei || !ej || !ew) return fail(DA_ERR_BAD_ARG, "NULL pointer");
  double thr;
  int64_t m;
  return edges_deliver(h->es, h->n_edges, &thr, &m, capacity, ei, ej, ew);
}

void da_edges_free(da_edges *h) { delete h; }


int64_t da_sig_ld(int n_hash) { return sig_ld_for(n_hash); }

// -------------------------------------------------------------- host entry

int da_minhash_signatures(const uint8_t *residues, const int64_t *offsets, int64_t n, int k,
                          int n_hash, const uint32_t *seeds, uint32_t *sig_out) {
  int rc = validate_mh(n, k, n_hash);
  if (rc != DA_OK) return rc;
  if (!residues || !seeds || !sig_out) return fail(DA_ERR_BAD_ARG, "NULL pointer");
  int64_t total, max_len;
  if ((rc = check_offsets(offsets, n, &total, &max_len)) != DA_OK) return rc;
  if ((rc = require_device()) != DA_OK) return rc;
  DeviceInput in;
  if ((rc = in.upload(residues, offsets, n, total, seeds, n_hash)) != DA_OK) return rc;
  const int64_t ld = sig_ld_for(n_hash);
  DevBuf sig;
  if ((rc = sig.alloc((size_t)n * ld * sizeof(uint32_t))) != DA_OK) return rc;
  rc = launch_minhash_signatures(in.res.as<uint8_t>(), in.off.as<int64_t>(), n, k, n_hash,
                                 in.seeds.as<uint32_t>(), sig.as<uint32_t>(), ld, nullptr);
  if (rc != DA_OK) return rc;
  DA_HIP_TRY(hipMemcpy2D(sig_out, (size_t)n_hash * 4, sig.p, (size_t)ld * 4, (size_t)n_hash * 4, (size_t)n,
                         hipMemcpyDeviceToHost));
  return DA_OK;
}

// similarityMH with more hash functions than the compare kernels' 16-bit counters hold (n_hash > 65535; the reference has
// no limit, src/minHash.cpp:129-131 only asks for > 0): chunks of <= 65504 hash functions go through K1 / K1b / K2 (uint16
// counts) one after the other, the counts are summed in 32 bits on the device and divided by n_hash once.  Needs the whole
// n x n count matrix resident (4 + 2 + 8 bytes per pair).
static int mh_host_chunked(const uint8_t *residues, const int64_t *offsets, int64_t n, int k, int n_hash, const uint32_t *seeds,
                           double *out) {
  constexpr int CHUNK = 65504;                              // a multiple of 32 below 65536
  int64_t total, max_len;
  int rc;
  if ((rc = check_offsets(offsets, n, &total, &max_len)) != DA_OK) return rc;
  if ((rc = require_device()) != DA_OK) return rc;
  if (rows_per_block(n, 4 + 2 + 8) < n)
    return fail(DA_ERR_UNSUPPORTED, "n_hash = %d (> 65535) needs the %lld x %lld count matrix resident in HBM (14 bytes per pair)", n_hash,
                (long long)n, (long long)n);
  DeviceInput in;
  if ((rc = in.upload(residues, offsets, n, total, seeds, n_hash)) != DA_OK) return rc;
  const int64_t lds = sig_ld_for(CHUNK);
  DevBuf sig, planes, work, acc, cnt, dout;
  const size_t wb = mh_planes_workspace_bytes(n, CHUNK);
  if ((rc = sig.alloc((size_t)n * lds * 4)) != DA_OK) return rc;
  if ((rc = planes.alloc((size_t)mh_planes_words(n, CHUNK) * 4)) != DA_OK) return rc;
  if ((rc = work.alloc(wb)) != DA_OK) return rc;
  if ((rc = acc.alloc((size_t)n * (size_t)n * 4)) != DA_OK) return rc;
  if ((rc = cnt.alloc((size_t)n * (size_t)n * 2)) != DA_OK) return rc;
  for (int h0 = 0; h0 < n_hash; h0 += CHUNK) {
    const int nh = std::min(CHUNK, n_hash - h0);
    if ((rc = launch_minhash_signatures(in.res.as<uint8_t>(), in.off.as<int64_t>(), n, k, nh, in.seeds.as<uint32_t>() + h0,
                                        sig.as<uint32_t>(), lds, nullptr)) != DA_OK) return rc;
    int bits = 32;
    if ((rc = build_planes(sig.as<uint32_t>(), lds, n, nh, 0, work.p, wb, planes.as<uint32_t>(), &bits, nullptr)) != DA_OK) return rc;
    if ((rc = launch_mh_compare(planes.as<uint32_t>(), n, nh, 0, n, true, DA_OUT_COMPACT, cnt.p, n, nullptr, bits)) != DA_OK) return rc;
    if ((rc = launch_acc_counts(acc.as<uint32_t>(), cnt.as<uint16_t>(), n * n, h0 == 0, nullptr)) != DA_OK) return rc;
  }
  if ((rc = dout.alloc((size_t)n * (size_t)n * 8)) != DA_OK) return rc;
  if ((rc = launch_counts32_to_f64(acc.as<uint32_t>(), dout.as<double>(), n * n, n_hash, nullptr)) != DA_OK) return rc;
  rc = d2h_pipelined(out, dout.p, (size_t)n * (size_t)n * 8);
  DA_HIP_TRY(hipDeviceSynchronize());
  return rc;
}

static int mh_host_common(const uint8_t *residues, const int64_t *offsets, int64_t n, int k, int n_hash,
                          const uint32_t *seeds, int64_t row_begin, int64_t row_end, int kind, void *out) {
  int rc = validate_mh(n, k, n_hash);
  if (rc != DA_OK) return rc;
  if (!residues || !seeds || !out) return fail(DA_ERR_BAD_ARG, "NULL pointer");
  if (row_begin < 0 || row_end > n || row_begin > row_end) return fail(DA_ERR_BAD_ARG, "bad row range");
  if (n_hash > 65535 && kind == DA_OUT_F64 && row_begin == 0 && row_end == n)
    return mh_host_chunked(residues, offsets, n, k, n_hash, seeds, static_cast<double *>(out));
  if (n_hash > 65535)
    return fail(DA_ERR_UNSUPPORTED, "uint16 match counts cannot hold n_hash = %d (> 65535); da_similarity_mh handles it", n_hash);
  int64_t total, max_len;
  if ((rc = check_offsets(offsets, n, &total, &max_len)) != DA_OK) return rc;
  if ((rc = require_device()) != DA_OK) return rc;
  Trace tr("similarityMH host path");
  DeviceInput in;
  if ((rc = in.upload(residues, offsets, n, total, seeds, n_hash)) != DA_OK) return rc;
  const int64_t lds = sig_ld_for(n_hash);
  DevBuf sig, planes;
  if ((rc = sig.alloc((size_t)n * lds * sizeof(uint32_t))) != DA_OK) return rc;
  if ((rc = planes.alloc((size_t)mh_planes_words(n, n_hash) * sizeof(uint32_t))) != DA_OK) return rc;
  tr.mark("upload + small allocations");
  rc = launch_minhash_signatures(in.res.as<uint8_t>(), in.off.as<int64_t>(), n, k, n_hash,
                                 in.seeds.as<uint32_t>(), sig.as<uint32_t>(), lds, nullptr);
  if (rc != DA_OK) return rc;
  int bits = 32;
  {
    DevBuf work;
    const size_t wb = mh_planes_workspace_bytes(n, n_hash);
    if ((rc = work.alloc(wb)) != DA_OK) return rc;
    if ((rc = build_planes(sig.as<uint32_t>(), lds, n, n_hash, 0, work.p, wb, planes.as<uint32_t>(), &bits, nullptr)) != DA_OK)
      return rc;
    DA_HIP_TRY(hipStreamSynchronize(nullptr));   // the workspace is released here
  }
  tr.mark("signatures + codes");
  if (kind == DA_OUT_F64 && row_begin == 0 && row_end == n && rows_per_block(n, sizeof(uint16_t)) >= n && !da::config().no_host_widen) {
    // the float64 matrix for a HOST caller: counts (uint16) leave the device, a quarter of the bytes over PCIe, and the host
    // widens them while copying -- count / n_hash is the same IEEE divide here as on the device (src/minHash.cpp:174)
    DevBuf dcnt;
    if ((rc = dcnt.alloc((size_t)n * (size_t)n * sizeof(uint16_t))) != DA_OK) return rc;
    if ((rc = launch_mh_compare(planes.as<uint32_t>(), n, n_hash, 0, n, true, DA_OUT_COMPACT, dcnt.p, n, nullptr, bits)) != DA_OK) return rc;
    tr.mark("compare (uint16 counts)");
    std::vector<double> table((size_t)n_hash + 1);
    for (int c = 0; c <= n_hash; ++c) table[(size_t)c] = (double)c / (double)n_hash;
    if ((rc = d2h_pipelined(out, dcnt.p, (size_t)n * (size_t)n * sizeof(uint16_t), table.data())) != DA_OK) return rc;
    tr.mark("device -> host + widen");
    return DA_OK;
  }
  const size_t esz = kind == DA_OUT_F64 ? sizeof(double) : sizeof(uint16_t);
  const int64_t rows_total = row_end - row_begin;
  const int64_t blk = rows_per_block(n, esz);
  const bool whole = (row_begin == 0 && row_end == n && blk >= n);
  DevBuf dout;
  if ((rc = dout.alloc((size_t)std::min(blk, rows_total) * (size_t)n * esz)) != DA_OK) return rc;
  tr.mark(whole ? "result allocation (whole)" : "result allocation (row blocks)");
  if (whole) {  // everything fits: compare only the upper triangle, store both halves
    rc = launch_mh_compare(planes.as<uint32_t>(), n, n_hash, 0, n, true, kind, dout.p, n, nullptr, bits);
    if (rc != DA_OK) return rc;
    tr.mark("compare");
    if ((rc = d2h_pipelined(out, dout.p, (size_t)n * (size_t)n * esz)) != DA_OK) return rc;
    tr.mark("device -> host");
    return DA_OK;
  }
  for (int64_t r0 = row_begin; r0 < row_end; r0 += blk) {
    const int64_t r1 = std::min(row_end, r0 + blk);
    rc = launch_mh_compare(planes.as<uint32_t>(), n, n_hash, r0, r1, false, kind, dout.p, n, nullptr, bits);
    if (rc != DA_OK) return rc;
    if ((rc = d2h_pipelined(static_cast<char *>(out) + (size_t)(r0 - row_begin) * (size_t)n * esz, dout.p,
                            (size_t)(r1 - r0) * (size_t)n * esz)) != DA_OK) return rc;
  }
  return DA_OK;
}

int da_similarity_mh(const uint8_t *residues, const int64_t *offsets, int64_t n, int k, int n_hash,
                     const uint32_t *seeds, double *out) {
  return mh_host_common(residues, offsets, n, k, n_hash, seeds, 0, n > 0 ? n : 0, DA_OUT_F64, out);
}

int da_mh_counts(const uint8_t *residues, const int64_t *offsets, int64_t n, int k, int n_hash,
                 const uint32_t *seeds, int64_t row_begin, int64_t row_end, uint16_t *counts_out) {
  return mh_host_common(residues, offsets, n, k, n_hash, seeds, row_begin, row_end, DA_OUT_COMPACT, counts_out);
}

// Residue validation with the reference's laziness (src/pairwiseSeqAlign.cpp:238-250):
// pairs are visited i ascending, j from i; inside a pair, sequence1[r] is checked at
// the start of DP row r+1, and all of sequence2 is scanned during DP row 1.  A pair
// with an empty sequence1 checks nothing.  Hence the first error raised is for the
// pair (i0, jb): i0 = first non-empty sequence, jb = first sequence holding an
// invalid byte (jb >= i0 necessarily).
static int nw_validate(const uint8_t *residues, const int64_t *offsets, int64_t n) {
  static const char order[] = "ARNDCQEGHILKMFPSTWYVBZX*";
  auto valid = [&](uint8_t c) { return c != 0 && strchr(order, c) != nullptr; };
  int64_t i0 = -1, jb = -1;
  for (int64_t i = 0; i < n && (i0 < 0 || jb < 0); ++i) {
    const int64_t b = offsets[i], e = offsets[i + 1];
    if (i0 < 0 && e > b) i0 = i;
    if (jb < 0)
      for (int64_t p = b; p < e; ++p)
        if (!valid(residues[p])) { jb = i; break; }
  }
  if (jb < 0) return DA_OK;
  const uint8_t *s2 = residues + offsets[jb];
  const int64_t n2 = offsets[jb + 1] - offsets[jb];
  if (jb == i0 && !valid(s2[0]))
    return fail(DA_ERR_BAD_RESIDUE_SEQ1, "Invalid amino acid in sequence1: %c", (char)s2[0]);
  for (int64_t p = 0; p < n2; ++p)
    if (!valid(s2[p])) return fail(DA_ERR_BAD_RESIDUE_SEQ2, "Invalid amino acid in sequence2: %c", (char)s2[p]);
  return DA_OK;  // unreachable
}

static int nw_host_common(const uint8_t *residues, const int64_t *offsets, int64_t n,
                          const char *matrix_name, int gap_open, int gap_ext, int64_t row_begin,
                          int64_t row_end, double *out_f64, int32_t *matches_out, int32_t *len_out,
                          int32_t *score_out) {
  const int mid = da_matrix_id(matrix_name);  // reference :338 -> :190-206, before anything else
  if (mid < 0) return DA_ERR_BAD_MATRIX;
  if (n <= 0) return DA_OK;                    // reference returns a 0x0 matrix
  if (!residues) return fail(DA_ERR_BAD_ARG, "NULL pointer");
  if (row_begin < 0 || row_end > n || row_begin > row_end) return fail(DA_ERR_BAD_ARG, "bad row range");
  int64_t total, max_len;
  int rc;
  if ((rc = check_offsets(offsets, n, &total, &max_len)) != DA_OK) return rc;
  if ((rc = nw_validate(residues, offsets, n)) != DA_OK) return rc;
  if ((rc = require_device()) != DA_OK) return rc;
  Trace tr("similarityNW host path");
  DeviceInput in;
  if ((rc = in.upload(residues, offsets, n, total, nullptr, 0)) != DA_OK) return rc;
  DevBuf codes, bad;
  if ((rc = codes.alloc((size_t)total)) != DA_OK) return rc;
  if ((rc = bad.alloc(sizeof(int32_t))) != DA_OK) return rc;
  DA_HIP_TRY(hipMemset(bad.p, 0, sizeof(int32_t)));
  if ((rc = launch_nw_encode(in.res.as<uint8_t>(), total, codes.as<uint8_t>(), bad.as<int32_t>(), nullptr)) != DA_OK)
    return rc;

  const int64_t rows_total = row_end - row_begin;
  if (out_f64 && row_begin == 0 && row_end == n && max_len <= 127 && rows_per_block(n, sizeof(uint16_t)) >= n &&
      !da::config().no_host_widen) {
    // as for similarityMH: the (matches << 8 | length) codes cross PCIe and the host divides (src/pairwiseSeqAlign.cpp:311)
    DevBuf dcode;
    if ((rc = dcode.alloc((size_t)n * (size_t)n * sizeof(uint16_t))) != DA_OK) return rc;
    tr.mark("validation + upload + allocations");
    if ((rc = nw_full_symmetric(codes.as<uint8_t>(), in.off.as<int64_t>(), n, total, max_len, mid, gap_open, gap_ext, DA_OUT_COMPACT,
                                dcode.p, n, nullptr)) != DA_OK) return rc;
    if (tr.on) {
      const NwRoute &r = nw_route();
      fprintf(stderr, "[dynaalign]   (plan %.2f ms, DP %.2f ms, expansion %.2f ms; %lld unique of %lld)\n", r.plan_ms, r.dp_ms, r.expand_ms,
                (long long)r.unique, (long long)r.n);
    }
    tr.mark("codes on the device (uint16)");
    std::vector<double> table(65536);
    const uint64_t nan_bits = 0xFFF8000000000000ULL;        // 0/0 as the reference's x86 host produces it
    for (uint32_t v = 0; v < 65536; ++v) {
      const uint32_t ln = v & 255u;
      if (ln == 0) memcpy(&table[v], &nan_bits, 8);
      else table[v] = (double)(v >> 8) / (double)ln;
    }
    rc = d2h_pipelined(out_f64, dcode.p, (size_t)n * (size_t)n * sizeof(uint16_t), table.data());
    tr.mark("device -> host + widen");
    return rc;
  }
  if (out_f64) {
    const int64_t blk = rows_per_block(n, sizeof(double));
    const bool whole = (row_begin == 0 && row_end == n && blk >= n);
    DevBuf dout;
    if ((rc = dout.alloc((size_t)std::min(blk, rows_total) * (size_t)n * sizeof(double))) != DA_OK) return rc;
    for (int64_t r0 = row_begin; r0 < row_end; r0 += blk) {
      const int64_t r1 = std::min(row_end, r0 + blk);
      if (whole)
        rc = nw_full_symmetric(codes.as<uint8_t>(), in.off.as<int64_t>(), n, total, max_len, mid, gap_open, gap_ext, DA_OUT_F64,
                               dout.p, n, nullptr);
      else
        rc = launch_nw(codes.as<uint8_t>(), in.off.as<int64_t>(), n, max_len, mid, gap_open, gap_ext, r0, r1,
                       false, DA_OUT_F64, dout.p, n, nullptr, 0, nullptr);
      if (rc != DA_OK) return rc;
      if ((rc = d2h_pipelined(out_f64 + (size_t)(r0 - row_begin) * (size_t)n, dout.p,
                              (size_t)(r1 - r0) * (size_t)n * sizeof(double))) != DA_OK) return rc;
    }
    return DA_OK;
  }
  // integer outputs: 32-bit packed (matches<<16|len) + score, unpacked on the host
  const int64_t blk = rows_per_block(n, sizeof(uint32_t) + sizeof(int32_t));
  DevBuf dpk, dsc;
  const int64_t brow = std::min(blk, rows_total);
  if ((rc = dpk.alloc((size_t)brow * (size_t)n * sizeof(uint32_t))) != DA_OK) return rc;
  if (score_out && (rc = dsc.alloc((size_t)brow * (size_t)n * sizeof(int32_t))) != DA_OK) return rc;
  std::vector<uint32_t> hpk((size_t)brow * (size_t)n);
  for (int64_t r0 = row_begin; r0 < row_end; r0 += blk) {
    const int64_t r1 = std::min(row_end, r0 + blk);
    const bool whole = (r0 == 0 && r1 == n);
    rc = launch_nw(codes.as<uint8_t>(), in.off.as<int64_t>(), n, max_len, mid, gap_open, gap_ext, r0, r1, whole,
                   DA_OUT_PACK32, dpk.p, n, score_out ? dsc.as<int32_t>() : nullptr, n, nullptr);
    if (rc != DA_OK) return rc;
    const size_t cnt = (size_t)(r1 - r0) * (size_t)n, base = (size_t)(r0 - row_begin) * (size_t)n;
    DA_HIP_TRY(hipMemcpy(hpk.data(), dpk.p, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (size_t e = 0; e < cnt; ++e) {
      if (matches_out) matches_out[base + e] = (int32_t)(hpk[e] >> 16);
      if (len_out) len_out[base + e] = (int32_t)(hpk[e] & 0xffffu);
    }
    if (score_out) DA_HIP_TRY(hipMemcpy(score_out + base, dsc.p, cnt * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  return DA_OK;
}

int da_similarity_nw(const uint8_t *residues, const int64_t *offsets, int64_t n, const char *matrix_name,
                     int gap_open, int gap_ext, double *out) {
  if (n > 0 && !out) return fail(DA_ERR_BAD_ARG, "NULL pointer");
  return nw_host_common(residues, offsets, n, matrix_name, gap_open, gap_ext, 0, n > 0 ? n : 0, out, nullptr,
                        nullptr, nullptr);
}

int da_nw_pairs(const uint8_t *residues, const int64_t *offsets, int64_t n, const char *matrix_name,
                int gap_open, int gap_ext, int64_t row_begin, int64_t row_end, int32_t *matches_out,
                int32_t *len_out, int32_t *score_out) {
  return nw_host_common(residues, offsets, n, matrix_name, gap_open, gap_ext, row_begin, row_end, nullptr,
                        matches_out, len_out, score_out);
}


// ------------------------------------------------- multi-device host entry points (SURVEY 8(b) da_opts, 8(e))
// ONE process drives several GPUs: a host thread per device (hipSetDevice is per thread), every device holds all
// sequences and rebuilds all signatures itself (2 MB in), the pair space is split, and every device copies ITS
// rows of the result straight into the caller's matrix -- P PCIe links instead of one, which is where more GPUs help
// the host-pointer boundary (the 80 GB result is PCIe-bound).  Exchange modes:
//   DA_EXCHANGE_ROWS      no device-to-device traffic: device p computes the contiguous row block it will copy out
//                         (full rows: twice the triangle's compare work, still far below the copy time);
//   DA_EXCHANGE_ALLGATHER the north-star's shape: cyclic upper-triangle shards -> ONE ncclAllGather over xGMI (RCCL,
//                         communicators from ncclCommInitAll) -> mirror + widen to the full matrix on every device;
//   DA_EXCHANGE_PEERCOPY  the same shards exchanged by direct hipMemcpyPeerAsync reads of every peer's block: xGMI is
//                         point-to-point, so P - 1 concurrent peer copies use all links at once where a ring is
//                         per-link bound; needs no RCCL.
// RCCL is bound at run time (dlopen of librccl.so.1): a process that already carries an RCCL (PyTorch's) keeps one copy.
}  // extern "C"  (reopened below)

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <array>
#include <chrono>

namespace da {
namespace {

struct Rccl {
  void *h = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;              // optional: tears down a communicator whose collective may be stuck
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
  bool load() {
    if (h) return true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
      why = dlerror();
    }
    if (!h) return false;
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(h, "ncclCommAbort"));
    AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(h, "ncclAllGather"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!CommInitAll || !CommDestroy || !AllGather || !GetErrorString) { why = "librccl lacks ncclCommInitAll / ncclAllGather"; h = nullptr; return false; }
    return true;
  }
};
Rccl &rccl() { static Rccl r; return r; }
std::mutex &rccl_mutex() { static std::mutex m; return m; }

// all ranks arrive with their status; everyone learns whether all were fine
class StatusBarrier {
 public:
  explicit StatusBarrier(int n) : n_(n) {}
  bool arrive(bool ok) {
    std::unique_lock<std::mutex> g(m_);
    if (!ok) all_ok_ = false;
    const uint64_t gen = gen_;
    if (++count_ == n_) { count_ = 0; result_ = all_ok_; ++gen_; cv_.notify_all(); return result_; }
    cv_.wait(g, [&]() { return gen_ != gen; });
    return result_;
  }
 private:
  std::mutex m_;
  std::condition_variable cv_;
  int n_, count_ = 0;
  uint64_t gen_ = 0;
  bool all_ok_ = true, result_ = true;
};

enum { PH_SETUP = 0, PH_COMPUTE, PH_EXCHANGE, PH_FINALIZE, PH_D2H, PH_TOTAL, PH_COUNT };

struct Multi {
  int P = 1, exchange = DA_EXCHANGE_ROWS;
  std::vector<int> devs;
  std::vector<ncclComm_t> comms;
  std::vector<const void *> block;                 // rank p's block to exchange (device pointer), PEERCOPY
  std::vector<int> rc;
  std::vector<std::string> msg;
  std::vector<std::array<double, PH_COUNT>> ms;
  std::unique_ptr<StatusBarrier> bar;
  double t_entry = 0.0, comm_setup_ms = 0.0;       // entry of the C call; ncclCommInitAll (first call with this device list)
  std::unique_lock<std::mutex> comm_use;           // held while this call uses the cached communicators
};

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// RCCL communicators are kept per device list: ncclCommInitAll costs far more than the exchange it serves, and the glue calls
// once per recursion level.  One call at a time uses them (comm_use_mutex); da_release_device_memory() destroys them.
std::map<std::vector<int>, std::vector<ncclComm_t>> &comm_cache() { static auto *c = new std::map<std::vector<int>, std::vector<ncclComm_t>>; return *c; }
std::mutex &comm_use_mutex() { static std::mutex m; return m; }

// rows rank p copies to the host: contiguous, boundaries on 128-row tiles
void row_split(int64_t n, int P, int p, int64_t *r0, int64_t *r1) {
  const int64_t T = ceil_div(n, 128);
  *r0 = std::min(n, ceil_div(T * p, P) * 128);
  *r1 = std::min(n, ceil_div(T * (p + 1), P) * 128);
}

int parse_opts(const da_opts *o, Multi &m) {
  m.t_entry = now_ms();                              // phase_ms[5] is the whole call: validation and communicator setup included
  m.P = 1; m.exchange = DA_EXCHANGE_ROWS; m.devs.clear();
  if (o) {
    if (o->struct_size < (uint32_t)offsetof(da_opts, phase_ms)) return fail(DA_ERR_BAD_ARG, "da_opts.struct_size is not set (sizeof(da_opts))");
    if (o->n_devices < 0 || (o->n_devices > 0 && !o->devices)) return fail(DA_ERR_BAD_ARG, "bad device list");
    if (o->exchange != DA_EXCHANGE_ROWS && o->exchange != DA_EXCHANGE_ALLGATHER && o->exchange != DA_EXCHANGE_PEERCOPY)
      return fail(DA_ERR_BAD_ARG, "unknown exchange mode %d", o->exchange);
    m.exchange = o->exchange;
    for (int i = 0; i < o->n_devices; ++i) m.devs.push_back(o->devices[i]);
  }
  int rc = require_device();
  if (rc != DA_OK) return rc;
  int cnt = 0;
  DA_HIP_TRY(hipGetDeviceCount(&cnt));
  if (m.devs.empty()) { int cur = 0; DA_HIP_TRY(hipGetDevice(&cur)); m.devs.push_back(cur); }
  for (size_t i = 0; i < m.devs.size(); ++i) {
    if (m.devs[i] < 0 || m.devs[i] >= cnt) return fail(DA_ERR_BAD_ARG, "device %d not present (%d visible)", m.devs[i], cnt);
    if (m.exchange == DA_EXCHANGE_ALLGATHER)          // RCCL refuses two ranks on one GPU; the other modes allow it (tests on a 1-GPU box)
      for (size_t j = 0; j < i; ++j)
        if (m.devs[j] == m.devs[i]) return fail(DA_ERR_BAD_ARG, "device %d listed twice (not allowed with DA_EXCHANGE_ALLGATHER)", m.devs[i]);
  }
  m.P = (int)m.devs.size();
  m.rc.assign(m.P, DA_OK); m.msg.assign(m.P, ""); m.block.assign(m.P, nullptr);
  m.ms.assign(m.P, std::array<double, PH_COUNT>{});
  m.bar.reset(new StatusBarrier(m.P));
  if (m.exchange == DA_EXCHANGE_ALLGATHER) {
    m.comm_use = std::unique_lock<std::mutex>(comm_use_mutex());
    std::lock_guard<std::mutex> g(rccl_mutex());
    if (!rccl().load()) { m.comm_use.unlock(); return fail(DA_ERR_UNSUPPORTED, "DA_EXCHANGE_ALLGATHER needs RCCL: %s", rccl().why.c_str()); }
    auto it = comm_cache().find(m.devs);
    if (it == comm_cache().end()) {
      const double t0 = now_ms();
      std::vector<ncclComm_t> comms((size_t)m.P, nullptr);
      const ncclResult_t r = rccl().CommInitAll(comms.data(), m.P, m.devs.data());
      if (r != ncclSuccess) { m.comm_use.unlock(); return fail(DA_ERR_HIP, "ncclCommInitAll failed: %s", rccl().GetErrorString(r)); }
      m.comm_setup_ms = now_ms() - t0;
      it = comm_cache().emplace(m.devs, std::move(comms)).first;
    }
    m.comms = it->second;
  }
  return DA_OK;
}

// the communicators stay cached (destroy_cached_comms); the call only gives up its right to use them.  `failed`: some rank of this
// call returned an error -- a communicator that has seen a failed or abandoned collective may hang the next one, so the cache entry of
// this device list is torn down (ncclCommAbort where the library has it) before the lock is released; the next call makes new ones.
size_t g_comm_evictions = 0;                        // (tests: da_debug_comm_cache_state)
void destroy_comms(Multi &m, bool failed = false) {
  if (failed && m.exchange == DA_EXCHANGE_ALLGATHER && m.comm_use.owns_lock()) {
    std::lock_guard<std::mutex> g(rccl_mutex());
    auto it = comm_cache().find(m.devs);
    if (it != comm_cache().end()) {
      for (auto c : it->second)
        if (c) (void)(rccl().CommAbort ? rccl().CommAbort(c) : rccl().CommDestroy(c));
      comm_cache().erase(it);
      ++g_comm_evictions;
    }
  }
  m.comms.clear();
  if (m.comm_use.owns_lock()) m.comm_use.unlock();
}

// direct xGMI reads of the peers' blocks (PEERCOPY); without it hipMemcpyPeer stages through the host
void enable_peer_access(const Multi &m, int p) {
  for (int q = 0; q < m.P; ++q)
    if (m.devs[q] != m.devs[p]) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, m.devs[p], m.devs[q]) == hipSuccess && can) {
        const hipError_t e = hipDeviceEnablePeerAccess(m.devs[q], 0);
        if (e != hipSuccess) (void)hipGetLastError();   // already enabled: fine
      }
    }
}

// run body(p) on one host thread per rank; the first failing rank's status / message become the caller's
template <typename F> int run_ranks(Multi &m, const da_opts *o, F body) {
  const double t0 = m.t_entry > 0.0 ? m.t_entry : now_ms();
  auto one = [&](int p) {
    int rc = DA_OK;
    if (hipSetDevice(m.devs[p]) != hipSuccess) rc = fail(DA_ERR_HIP, "hipSetDevice(%d) failed", m.devs[p]);
    if (rc == DA_OK && m.exchange == DA_EXCHANGE_PEERCOPY) enable_peer_access(m, p);
    if (rc == DA_OK) rc = body(p);
    m.rc[p] = rc;
    if (rc != DA_OK) m.msg[p] = last_error_ref();
  };
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (m.P == 1) one(0);
  else {
    std::vector<std::thread> th;
    for (int p = 0; p < m.P; ++p) th.emplace_back(one, p);
    for (auto &t : th) t.join();
  }
  (void)hipSetDevice(cur);
  bool any_failed = false;
  for (int p = 0; p < m.P; ++p) any_failed = any_failed || m.rc[p] != DA_OK;
  destroy_comms(m, any_failed);
  if (o && o->struct_size >= sizeof(da_opts) && o->phase_ms) {
    for (int k = 0; k < PH_COUNT; ++k) {
      double mx = 0.0;
      for (int p = 0; p < m.P; ++p) mx = std::max(mx, m.ms[p][k]);
      o->phase_ms[k] = mx;
    }
    o->phase_ms[PH_SETUP] += m.comm_setup_ms;          // ncclCommInitAll, paid by the first call with this device list
    o->phase_ms[PH_TOTAL] = now_ms() - t0;
  }
  for (int p = 0; p < m.P; ++p)
    if (m.rc[p] != DA_OK) { last_error_ref() = m.msg[p]; return m.rc[p]; }
  return DA_OK;
}

// the exchange step of the sharded modes: my block (bytes) -> gathered[P][bytes] on this rank's device
int exchange_blocks(Multi &m, int p, const void *mine, size_t bytes, void *gathered) {
  if (m.exchange == DA_EXCHANGE_ALLGATHER) {
    const ncclResult_t r = rccl().AllGather(mine, gathered, bytes, ncclUint8, m.comms[p], nullptr);
    if (r != ncclSuccess) return fail(DA_ERR_HIP, "ncclAllGather failed: %s", rccl().GetErrorString(r));
    DA_HIP_TRY(hipStreamSynchronize(nullptr));
    return DA_OK;
  }
  // PEERCOPY: every peer's block is complete (the barrier before this call); read them all, concurrently
  for (int q = 0; q < m.P; ++q) {
    char *dst = static_cast<char *>(gathered) + (size_t)q * bytes;
    if (m.devs[q] == m.devs[p]) DA_HIP_TRY(hipMemcpyAsync(dst, m.block[q], bytes, hipMemcpyDeviceToDevice, nullptr));
    else DA_HIP_TRY(hipMemcpyPeerAsync(dst, m.devs[p], m.block[q], m.devs[q], bytes, nullptr));
  }
  DA_HIP_TRY(hipStreamSynchronize(nullptr));
  return DA_OK;
}

}  // namespace

size_t destroy_cached_comms() {
  std::lock_guard<std::mutex> u(comm_use_mutex());
  std::lock_guard<std::mutex> g(rccl_mutex());
  size_t k = 0;
  for (auto &e : comm_cache())
    for (auto c : e.second) if (c) { (void)rccl().CommDestroy(c); ++k; }
  comm_cache().clear();
  return k;
}
}  // namespace da

extern "C" {

/* tests: {cached device lists, evictions after a failed ALLGATHER call}; with force_fail != 0 the NEXT DA_EXCHANGE_ALLGATHER call's rank 0
 * fails right after its compute phase (before the collective) -- every rank then skips the exchange and the cache entry must go */
static std::atomic<int> g_force_exchange_error{0};
int da_debug_comm_cache_state(int force_fail, size_t *out2) {
  std::lock_guard<std::mutex> u(da::comm_use_mutex());
  if (out2) { out2[0] = da::comm_cache().size(); out2[1] = da::g_comm_evictions; }
  if (force_fail) g_force_exchange_error.store(1);
  return DA_OK;
}

int da_rccl_available(void) {
  std::lock_guard<std::mutex> g(rccl_mutex());
  return rccl().load() ? 1 : 0;
}

int da_similarity_mh_opts(const uint8_t *residues, const int64_t *offsets, int64_t n, int k, int n_hash,
                          const uint32_t *seeds, double *out, const da_opts *opts) {
  int rc = validate_mh(n, k, n_hash);
  if (rc != DA_OK) return rc;
  if (!residues || !seeds || !out) return fail(DA_ERR_BAD_ARG, "NULL pointer");
  int64_t total, max_len;
  if ((rc = check_offsets(offsets, n, &total, &max_len)) != DA_OK) return rc;
  Multi m;
  if ((rc = parse_opts(opts, m)) != DA_OK) return rc;
  if (n_hash > 65535 && !(m.P == 1 && m.exchange == DA_EXCHANGE_ROWS)) {
    destroy_comms(m);
    return fail(DA_ERR_UNSUPPORTED, "n_hash = %d (> 65535) runs on one device only (16-bit counters in the sharded kernels)", n_hash);
  }
  if (m.P == 1 && m.exchange == DA_EXCHANGE_ROWS)      // one device, nothing to split: the symmetric single-device path
    return run_ranks(m, opts, [&](int) -> int {
      return mh_host_common(residues, offsets, n, k, n_hash, seeds, 0, n, DA_OUT_F64, out);
    });
  const int vbits = std::max(8, 32 - __builtin_clz((unsigned)n_hash));
  const ShardGeom sg = shard_geom(n, m.P, 128);
  const size_t packed = (size_t)shard_packed_bytes(sg, vbits);
  return run_ranks(m, opts, [&](int p) -> int {
    int rc = DA_OK;
    bool ok;
    double t = now_ms();
    DeviceInput in;
    DevBuf sig, planes, local, pk, gathered, dout;
    int bits = 32;
    const int64_t lds = sig_ld_for(n_hash);
    int64_t r0, r1;
    row_split(n, m.P, p, &r0, &r1);
    do {   // setup: upload + K1 + K1b on this device
      if ((rc = in.upload(residues, offsets, n, total, seeds, n_hash)) != DA_OK) break;
      if ((rc = sig.alloc((size_t)n * lds * 4)) != DA_OK) break;
      if ((rc = planes.alloc((size_t)mh_planes_words(n, n_hash) * 4)) != DA_OK) break;
      if ((rc = launch_minhash_signatures(in.res.as<uint8_t>(), in.off.as<int64_t>(), n, k, n_hash, in.seeds.as<uint32_t>(),
                                          sig.as<uint32_t>(), lds, nullptr)) != DA_OK) break;
      DevBuf work;
      const size_t wb = mh_planes_workspace_bytes(n, n_hash);
      if ((rc = work.alloc(wb)) != DA_OK) break;
      if ((rc = build_planes(sig.as<uint32_t>(), lds, n, n_hash, 0, work.p, wb, planes.as<uint32_t>(), &bits, nullptr)) != DA_OK) break;
      if (hipStreamSynchronize(nullptr) != hipSuccess) rc = fail(DA_ERR_HIP, "stream synchronisation failed");
    } while (0);
    m.ms[p][PH_SETUP] = now_ms() - t;
    if (m.exchange == DA_EXCHANGE_ROWS) {
      if (rc != DA_OK || r1 <= r0) return rc;
      const int64_t blk = rows_per_block(n, sizeof(double));
      if ((rc = dout.alloc((size_t)std::min(blk, r1 - r0) * (size_t)n * sizeof(double))) != DA_OK) return rc;
      for (int64_t b0 = r0; b0 < r1 && rc == DA_OK; b0 += blk) {
        const int64_t b1 = std::min(r1, b0 + blk);
        t = now_ms();
        if ((rc = launch_mh_compare(planes.as<uint32_t>(), n, n_hash, b0, b1, false, DA_OUT_F64, dout.p, n, nullptr, bits)) != DA_OK) break;
        if (hipStreamSynchronize(nullptr) != hipSuccess) { rc = fail(DA_ERR_HIP, "compare kernel failed"); break; }
        m.ms[p][PH_COMPUTE] += now_ms() - t;
        t = now_ms();
        rc = d2h_pipelined(out + (size_t)b0 * (size_t)n, dout.p, (size_t)(b1 - b0) * (size_t)n * sizeof(double));
        m.ms[p][PH_D2H] += now_ms() - t;
      }
      return rc;
    }
    // sharded modes: cyclic upper-triangle tiles -> packed block -> exchange -> full matrix on this device
    t = now_ms();
    if (rc == DA_OK) do {
      if ((rc = local.alloc((size_t)sg.rows * (size_t)sg.W * 2)) != DA_OK) break;
      if ((rc = pk.alloc(packed)) != DA_OK) break;
      if ((rc = gathered.alloc(packed * (size_t)m.P)) != DA_OK) break;
      if ((rc = dout.alloc((size_t)n * (size_t)n * sizeof(double))) != DA_OK) break;
      if (hipMemsetAsync(local.p, 0, (size_t)sg.rows * (size_t)sg.W * 2, nullptr) != hipSuccess) { rc = fail(DA_ERR_HIP, "hipMemsetAsync failed"); break; }
      if ((int64_t)p * 128 < n)
        if ((rc = launch_mh_compare(planes.as<uint32_t>(), n, n_hash, (int64_t)p * 128, n, false, DA_OUT_COMPACT, local.p, sg.W, nullptr,
                                    bits, m.P, true, sg.Q, sg.W)) != DA_OK) break;
      if ((rc = launch_pack_shard(local.as<uint16_t>(), sg.W, sg, vbits, pk.as<uint8_t>(), nullptr)) != DA_OK) break;
      if (hipStreamSynchronize(nullptr) != hipSuccess) rc = fail(DA_ERR_HIP, "shard compare failed");
    } while (0);
    m.ms[p][PH_COMPUTE] = now_ms() - t;
    m.block[p] = pk.p;
    if (p == 0 && rc == DA_OK && m.exchange == DA_EXCHANGE_ALLGATHER && g_force_exchange_error.exchange(0))   // (test hook, da_debug_comm_cache_state)
      rc = fail(DA_ERR_HIP, "forced exchange failure (test hook)");
    ok = m.bar->arrive(rc == DA_OK);                 // every block is complete (and nobody failed) before anyone exchanges
    if (!ok) return rc;
    t = now_ms();
    rc = exchange_blocks(m, p, pk.p, packed, gathered.p);
    m.ms[p][PH_EXCHANGE] = now_ms() - t;
    ok = m.bar->arrive(rc == DA_OK);                 // peers have finished reading my block before it is freed
    if (!ok) return rc;
    t = now_ms();
    if ((rc = launch_finalize_packed(gathered.as<uint8_t>(), sg, vbits, n_hash, dout.as<double>(), n, nullptr)) != DA_OK) return rc;
    if (hipStreamSynchronize(nullptr) != hipSuccess) return fail(DA_ERR_HIP, "finalize kernel failed");
    m.ms[p][PH_FINALIZE] = now_ms() - t;
    t = now_ms();
    if (r1 > r0) rc = d2h_pipelined(out + (size_t)r0 * (size_t)n, dout.as<double>() + (size_t)r0 * (size_t)n, (size_t)(r1 - r0) * (size_t)n * sizeof(double));
    m.ms[p][PH_D2H] = now_ms() - t;
    return rc;
  });
}

int da_similarity_nw_opts(const uint8_t *residues, const int64_t *offsets, int64_t n, const char *matrix_name, int gap_open,
                          int gap_ext, double *out, const da_opts *opts) {
  const int mid = da_matrix_id(matrix_name);          // reference :338 -> :190-206, before anything else
  if (mid < 0) return DA_ERR_BAD_MATRIX;
  if (n <= 0) return DA_OK;                            // reference returns a 0x0 matrix
  if (!residues || !out) return fail(DA_ERR_BAD_ARG, "NULL pointer");
  int64_t total, max_len;
  int rc;
  if ((rc = check_offsets(offsets, n, &total, &max_len)) != DA_OK) return rc;
  if ((rc = nw_validate(residues, offsets, n)) != DA_OK) return rc;
  Multi m;
  if ((rc = parse_opts(opts, m)) != DA_OK) return rc;
  if (m.exchange != DA_EXCHANGE_ROWS && max_len > 64) {
    destroy_comms(m);
    return fail(DA_ERR_UNSUPPORTED, "the sharded NW exchange works on uint16 codes of sequences up to 64 residues (longest here: %lld); "
                                    "use DA_EXCHANGE_ROWS", (long long)max_len);
  }
  if (m.P == 1 && m.exchange == DA_EXCHANGE_ROWS)
    return run_ranks(m, opts, [&](int) -> int {
      return nw_host_common(residues, offsets, n, matrix_name, gap_open, gap_ext, 0, n, out, nullptr, nullptr, nullptr);
    });
  const ShardGeom sg = shard_geom(n, m.P, 128);
  const size_t blk_bytes = (size_t)sg.rows * (size_t)sg.W * 2;
  return run_ranks(m, opts, [&](int p) -> int {
    int rc = DA_OK;
    bool ok;
    double t = now_ms();
    DeviceInput in;
    DevBuf codes, bad, local, gathered, dout;
    int64_t r0, r1;
    row_split(n, m.P, p, &r0, &r1);
    do {
      if ((rc = in.upload(residues, offsets, n, total, nullptr, 0)) != DA_OK) break;
      if ((rc = codes.alloc((size_t)total)) != DA_OK) break;
      if ((rc = bad.alloc(sizeof(int32_t))) != DA_OK) break;
      if (hipMemsetAsync(bad.p, 0, sizeof(int32_t), nullptr) != hipSuccess) { rc = fail(DA_ERR_HIP, "hipMemsetAsync failed"); break; }
      rc = launch_nw_encode(in.res.as<uint8_t>(), total, codes.as<uint8_t>(), bad.as<int32_t>(), nullptr);
    } while (0);
    m.ms[p][PH_SETUP] = now_ms() - t;
    if (m.exchange == DA_EXCHANGE_ROWS) {
      if (rc != DA_OK || r1 <= r0) return rc;
      const int64_t blk = rows_per_block(n, sizeof(double));
      if ((rc = dout.alloc((size_t)std::min(blk, r1 - r0) * (size_t)n * sizeof(double))) != DA_OK) return rc;
      for (int64_t b0 = r0; b0 < r1 && rc == DA_OK; b0 += blk) {
        const int64_t b1 = std::min(r1, b0 + blk);
        t = now_ms();
        if ((rc = launch_nw(codes.as<uint8_t>(), in.off.as<int64_t>(), n, max_len, mid, gap_open, gap_ext, b0, b1, false, DA_OUT_F64,
                            dout.p, n, nullptr, 0, nullptr)) != DA_OK) break;
        if (hipStreamSynchronize(nullptr) != hipSuccess) { rc = fail(DA_ERR_HIP, "NW kernel failed"); break; }
        m.ms[p][PH_COMPUTE] += now_ms() - t;
        t = now_ms();
        rc = d2h_pipelined(out + (size_t)b0 * (size_t)n, dout.p, (size_t)(b1 - b0) * (size_t)n * sizeof(double));
        m.ms[p][PH_D2H] += now_ms() - t;
      }
      return rc;
    }
    t = now_ms();
    if (rc == DA_OK) do {
      if ((rc = local.alloc(blk_bytes)) != DA_OK) break;
      if ((rc = gathered.alloc(blk_bytes * (size_t)m.P)) != DA_OK) break;
      if ((rc = dout.alloc((size_t)n * (size_t)n * sizeof(double))) != DA_OK) break;
      if (hipMemsetAsync(local.p, 0, blk_bytes, nullptr) != hipSuccess) { rc = fail(DA_ERR_HIP, "hipMemsetAsync failed"); break; }
      if ((rc = launch_nw(codes.as<uint8_t>(), in.off.as<int64_t>(), n, max_len, mid, gap_open, gap_ext, 0, n, false, DA_OUT_COMPACT,
                          local.p, sg.W, nullptr, 0, nullptr, p, m.P)) != DA_OK) break;
      if (hipStreamSynchronize(nullptr) != hipSuccess) rc = fail(DA_ERR_HIP, "NW shard kernel failed");
    } while (0);
    m.ms[p][PH_COMPUTE] = now_ms() - t;
    m.block[p] = local.p;
    ok = m.bar->arrive(rc == DA_OK);
    if (!ok) return rc;
    t = now_ms();
    rc = exchange_blocks(m, p, local.p, blk_bytes, gathered.p);
    m.ms[p][PH_EXCHANGE] = now_ms() - t;
    ok = m.bar->arrive(rc == DA_OK);
    if (!ok) return rc;
    t = now_ms();
    if ((rc = launch_finalize_sharded(gathered.as<uint16_t>(), sg.W, sg, true, 0, dout.as<double>(), n, nullptr)) != DA_OK) return rc;
    if (hipStreamSynchronize(nullptr) != hipSuccess) return fail(DA_ERR_HIP, "finalize kernel failed");
    m.ms[p][PH_FINALIZE] = now_ms() - t;
    t = now_ms();
    if (r1 > r0) rc = d2h_pipelined(out + (size_t)r0 * (size_t)n, dout.as<double>() + (size_t)r0 * (size_t)n, (size_t)(r1 - r0) * (size_t)n * sizeof(double));
    m.ms[p][PH_D2H] = now_ms() - t;
    return rc;
  });
}

}  // extern "C"
