// dict_kernels.hip -- exact 16-bit re-coding of the MinHash signatures before the compare.
//
// The compare (reference src/minHash.cpp:168-173) only asks whether sig[i][h] == sig[j][h].
// Any per-column injective re-coding of the values keeps every answer, and a value that occurs
// in ONE sequence of column h can never match anything.  So per hash function h:
//     values seen >= 2 times  ->  dense ids 0 .. D-1               (D <= n/2)
//     values seen once        ->  0xFFFE on the row side, 0xFFFF on the column side
// (row and column operands of the compare kernel are separate arrays, so two different
// singletons -- and a singleton against itself -- always read as "different"; the compare
// kernel forces the diagonal, which the reference sets to 1.0 anyway, src/minHash.cpp:161).
// D + 2 <= 65536 holds for every input with n <= DA_DICT_MAX_N = 131068, so 16 bit planes per
// 32 hash functions carry the same information as the 32 planes of the raw values and the
// bit-sliced compare does half the work.  Bit-exact by construction, not a sketch.
//
//   k_sig_transpose : sig[n][ld] -> sigT[n_hash][ldT]          (coalesced column access)
//   k_dictionary    : one workgroup per hash function; LDS hash table over key partitions
//   k_ids_to_planes : ids[n_hash][ldT] (uint16) -> bit planes, 8 / 12 / 16 per 32-hash group (as
//                     many as the largest column dictionary needs), row copy + pair-swapped column
//                     copy, in the blocked order k_mh_compare stages them in (da_common.hpp)
//   k_sig_to_planes : raw 32-plane operand from sig (n > DA_DICT_MAX_N, or dictionary overflow)
#include "da_common.hpp"
#include <atomic>

namespace da {
namespace {

// ------------------------------------------------------------- transpose --
__global__ __launch_bounds__(256) void k_sig_transpose(const uint32_t *__restrict__ sig, int64_t ld_sig, int64_t n,
                                                       int n_hash, uint32_t *__restrict__ sigT, int64_t ldT) {
  __shared__ uint32_t t[64][65];
  const int64_t i0 = (int64_t)blockIdx.x * 64;
  const int h0 = blockIdx.y * 64;
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;   // 64 x 4
  for (int r = ly; r < 64; r += 4) {
    const int64_t i = i0 + r;
    const int h = h0 + lx;
    t[r][lx] = (i < n && h < n_hash) ? sig[i * ld_sig + h] : 0u;
  }
  __syncthreads();
  for (int r = ly; r < 64; r += 4) {
    const int h = h0 + r;
    const int64_t i = i0 + lx;
    if (h < n_hash && i < ldT) sigT[(int64_t)h * ldT + i] = t[lx][r];
  }
}

// ------------------------------------------------------------ dictionary --
constexpr int DK_THREADS = 1024;
constexpr int DK_SLOTS = 16384;        // LDS table slots per key partition (power of two)
constexpr int DK_PART_KEYS = 8192;     // partitions are sized for <= 50 % load even if every key is distinct
constexpr int DK_MAX_PARTS = 16;       // partitions: n <= DA_DICT_MAX_N = 131068 -> ceil(n / 8192) <= 16

// a bijection of uint32 (murmur3's finaliser): partition and slot are taken from the mixed key,
// so the skew of min-hash VALUES (they crowd near 0) does not reach the table
__device__ __forceinline__ uint32_t dk_mix(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ int dk_part(uint32_t z, int R) { return (int)__umulhi(z, (uint32_t)R); }
__device__ __forceinline__ uint32_t dk_slot(uint32_t z) { return (z * 0x9e3779b1u) >> (32 - 14); }
static_assert(DK_SLOTS == (1 << 14), "dk_slot assumes 2^14 slots");

// status[0]: 0 ok, 1 a partition overflowed its table, 2 more than 65534 repeated values
// status[1]: the largest number of repeated values (= ids handed out) in any column
//
// One workgroup per hash function (column of sigT).
//   1. the column's mixed keys are counting-sorted by partition into the scratch arrays
//      (sz = mixed key, si = sequence index; order inside a partition is arbitrary);
//   2. per partition: LDS hash table -- first arrival claims a slot, later arrivals of the same
//      key flag it; flagged slots are numbered (base + scan); every key is looked up again and
//      its code written to idsT[si].
// Each key is read four times in all, whatever the number of partitions.
__global__ __launch_bounds__(DK_THREADS) void k_dictionary(const uint32_t *__restrict__ sigT, int64_t ldT, int64_t n,
                                                           int R, uint32_t *__restrict__ szT, uint32_t *__restrict__ siT,
                                                           uint16_t *__restrict__ idsT, int64_t ld_ids,
                                                           int *__restrict__ status) {
  constexpr int NW = DK_THREADS / 64;
  __shared__ uint32_t keys[DK_SLOTS];
  __shared__ uint16_t vals[DK_SLOTS];
  __shared__ uint32_t wsum[NW];
  __shared__ uint32_t cnt[NW][DK_MAX_PARTS];      // keys of partition p seen by wave w, then its write cursor
  __shared__ uint32_t pbase[DK_MAX_PARTS + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t *col = sigT + (int64_t)blockIdx.x * ldT;
  uint32_t *sz = szT + (int64_t)blockIdx.x * ldT;
  uint32_t *si = siT + (int64_t)blockIdx.x * ldT;
  uint16_t *ids = idsT + (int64_t)blockIdx.x * ld_ids;

  // ---- 1. counting sort by partition (per-wave counters keep the LDS atomics apart)
  for (int t = tid; t < NW * DK_MAX_PARTS; t += DK_THREADS) (&cnt[0][0])[t] = 0;
  __syncthreads();
  for (int64_t i = tid; i < n; i += DK_THREADS) atomicAdd(&cnt[wave][dk_part(dk_mix(col[i]), R)], 1u);
  __syncthreads();
  if (tid < R) {                     // cnt[w][p] -> first slot of wave w inside partition p (relative)
    uint32_t run = 0;
    for (int w = 0; w < NW; ++w) { const uint32_t c = cnt[w][tid]; cnt[w][tid] = run; run += c; }
    pbase[tid + 1] = run;            // size of partition tid for now
  }
  __syncthreads();
  if (tid == 0) {
    pbase[0] = 0;
    for (int p = 0; p < R; ++p) pbase[p + 1] += pbase[p];
  }
  __syncthreads();
  for (int64_t i = tid; i < n; i += DK_THREADS) {
    const uint32_t z = dk_mix(col[i]);
    const int p = dk_part(z, R);
    const uint32_t pos = pbase[p] + atomicAdd(&cnt[wave][p], 1u);
    sz[pos] = z;
    si[pos] = (uint32_t)i;
  }
  __syncthreads();                   // the workgroup's own global writes are visible to it from here on

  // ---- 2. one LDS table per partition
  uint32_t base = 0;                 // dense ids handed out so far (uniform over the workgroup)
  bool overflow = false;
  for (int r = 0; r < R; ++r) {
    const uint32_t lo = pbase[r], hi = pbase[r + 1];
    // a mixed key that can never be inserted in this pass marks an empty slot: the smallest z
    // of partition (r+1) % R  (R >= 2)
    const uint32_t q = (uint32_t)((r + 1) % R);
    const uint32_t EMPTY = (uint32_t)((((uint64_t)q << 32) + (uint32_t)R - 1) / (uint32_t)R);
    for (int s = tid; s < DK_SLOTS; s += DK_THREADS) { keys[s] = EMPTY; vals[s] = 0; }
    __syncthreads();
    for (uint32_t j = lo + tid; j < hi; j += DK_THREADS) {
      const uint32_t z = sz[j];
      uint32_t s = dk_slot(z);
      for (int probes = 0;; ++probes) {
        const uint32_t old = atomicCAS(&keys[s], EMPTY, z);
        if (old == EMPTY) break;
        if (old == z) { vals[s] = 1; break; }
        s = (s + 1) & (DK_SLOTS - 1);
        if (probes >= DK_SLOTS) { overflow = true; break; }
      }
    }
    __syncthreads();

    // number the repeated keys of this partition: base + exclusive scan over the slots
    uint32_t mine = 0;
    for (int s = tid; s < DK_SLOTS; s += DK_THREADS) mine += vals[s];
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int w = 0; w < NW; ++w) {
      const uint32_t v = wsum[w];
      if (w < wave) before += v;
      total += v;
    }
    uint32_t next = base + before + incl - mine;
    for (int s = tid; s < DK_SLOTS; s += DK_THREADS) {
      if (keys[s] == EMPTY) continue;
      vals[s] = vals[s] ? (uint16_t)(next++) : (uint16_t)0xFFFFu;
    }
    base += total;
    __syncthreads();

    // look every key of the partition up again and emit its code
    for (uint32_t j = lo + tid; j < hi; j += DK_THREADS) {
      const uint32_t z = sz[j];
      uint32_t s = dk_slot(z);
      uint16_t id = 0xFFFFu;
      for (int probes = 0; probes <= DK_SLOTS; ++probes) {
        const uint32_t cur = keys[s];
        if (cur == z) { id = vals[s]; break; }
        if (cur == EMPTY) break;                       // only after an overflow
        s = (s + 1) & (DK_SLOTS - 1);
      }
      ids[si[j]] = id;
    }
    __syncthreads();
  }
  if (overflow) atomicMax(status, 1);
  if (tid == 0 && base > 65534u) atomicMax(status, 2);
  if (tid == 0) atomicMax(status + 1, (int)base);
}

// -------------------------------------------------- heavy / rare split (round 4) --
// The bit-sliced compare pays per code PLANE, and a column dictionary of clustered data is very skewed: on the h3n2-like 100k set a column has
// ~4 000 repeated values, but the 254 most frequent ones carry all but 7*10^6 of the 2.2*10^10 matching (pair, hash function) incidences.  So,
// exactly:  matches(i, j) = #{h : code equal, code HEAVY in column h} + #{h : code equal, code RARE in column h}
//   idsD: heavy values -> ranks 0 .. keep-1, everything else 0xFFFF (= the two "never equal" codes of k_ids_to_planes): the dense compare runs on
//         8 planes instead of 12 - 16;
//   idsS: rare repeated values keep their dictionary code, everything else 0xFFFF: their incidences are enumerated by the sparse route's list
//         kernels (minhash_kernels.hip k_sp_classes .. k_sp_band) and added to the dense result by k_hy_fixup.
// One workgroup per column: LDS histogram of the codes, the count threshold of the `keep` most frequent ones by bisection, ties at the threshold
// taken in code order until the quota is used up; stats[0] += incidences among rare values, stats[1] = max(largest rare class), stats[2] / [3]: the same
// over ALL repeated values (what k_sp_count reports: the sparse route's admission test).
__global__ __launch_bounds__(1024) void k_hy_split(const uint16_t *__restrict__ idsT, int64_t ld_ids, int n, int max_ids, int keep,
                                                   uint16_t *__restrict__ idsD, uint16_t *__restrict__ idsS, unsigned long long *__restrict__ stats) {
  extern __shared__ uint32_t hy_hist[];        // max_ids counters (later: ranks), then 1024 partial sums
  uint32_t *part = hy_hist + max_ids;
  __shared__ uint32_t red[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint16_t *ids = idsT + (int64_t)blockIdx.x * ld_ids;
  uint16_t *outD = idsD + (int64_t)blockIdx.x * ld_ids, *outS = idsS + (int64_t)blockIdx.x * ld_ids;
  for (int c = tid; c < max_ids; c += 1024) hy_hist[c] = 0;
  __syncthreads();
  // (eight codes per 16-byte load: the rows are 128-byte aligned and padded to a multiple of 64 codes -- both passes over the column are latency-bound otherwise)
  const uint4 *ids8 = reinterpret_cast<const uint4 *>(ids);
  const int n8 = (n + 7) / 8;
  for (int q = tid; q < n8; q += 1024) {
    const uint4 v = ids8[q];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t c = (w[u >> 1] >> (16 * (u & 1))) & 0xffffu;
      if (q * 8 + u < n && c != 0xFFFFu) atomicAdd(&hy_hist[c], 1u);
    }
  }
  __syncthreads();
  auto count_ge = [&](uint32_t t) -> uint32_t {      // how many codes occur at least t times (uniform result)
    uint32_t k = 0;
    for (int c = tid; c < max_ids; c += 1024) k += hy_hist[c] >= t;
    for (int o = 32; o > 0; o >>= 1) k += __shfl_down(k, o);
    __syncthreads();                                 // (red[] of the previous round has been read by everyone)
    if (lane == 0) red[wave] = k;
    __syncthreads();
    uint32_t tot = 0;
    for (int w = 0; w < 16; ++w) tot += red[w];
    return tot;
  };
  uint32_t lo = 1, hi = (uint32_t)n + 1u;            // smallest t with count_ge(t) <= keep  (count_ge(n + 1) = 0)
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (count_ge(mid) <= (uint32_t)keep) hi = mid; else lo = mid + 1;
  }
  const uint32_t t = lo, above = count_ge(t), quota = (uint32_t)keep - above;    // ties: values seen exactly t - 1 times (>= 2: a repeated value)
  const bool ties = t >= 3;
  const int per = (max_ids + 1023) / 1024, c_lo = tid * per, c_hi = c_lo + per < max_ids ? c_lo + per : max_ids;
  uint32_t f1 = 0, f2 = 0;
  for (int c = c_lo; c < c_hi; ++c) { const uint32_t m = hy_hist[c]; f1 += m >= t; f2 += ties && m == t - 1; }
  part[tid] = (f1 << 16) | f2;                       // f1 sums stay <= keep < 2^16, f2 sums <= max_ids <= 2^15
  __syncthreads();
  if (tid < 64) {                                    // exclusive scan of the 1024 packed partial sums by one wave (16 each)
    uint32_t v[16], tot = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { v[q] = part[tid * 16 + q]; tot += v[q]; }
    uint32_t incl = tot;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if (tid >= o) incl += up; }
    uint32_t run = incl - tot;
#pragma unroll
    for (int q = 0; q < 16; ++q) { part[tid * 16 + q] = run; run += v[q]; }
  }
  __syncthreads();
  uint32_t p1 = part[tid] >> 16, p2 = part[tid] & 0xffffu;
  unsigned long long e = 0, mx = 0, e_all = 0, mx_all = 0;
  for (int c = c_lo; c < c_hi; ++c) {
    const uint32_t m = hy_hist[c];
    uint32_t rank = 0xFFFFu;
    if (m >= t) rank = p1++;
    else if (ties && m == t - 1) { if (p2 < quota) rank = above + p2; ++p2; }
    if (m >= 2) {
      const unsigned long long inc = (unsigned long long)m * (m - 1) / 2;
      e_all += inc; mx_all = m > mx_all ? m : mx_all;
      if (rank == 0xFFFFu) { e += inc; mx = m > mx ? m : mx; }
    }
    hy_hist[c] = rank;
  }
  for (int o = 32; o > 0; o >>= 1) {
    e += __shfl_down(e, o);
    e_all += __shfl_down(e_all, o);
    const unsigned long long other = __shfl_down(mx, o), other_all = __shfl_down(mx_all, o);
    mx = other > mx ? other : mx;
    mx_all = other_all > mx_all ? other_all : mx_all;
  }
  if (lane == 0) { atomicAdd(&stats[0], e); atomicMax(&stats[1], mx); atomicAdd(&stats[2], e_all); atomicMax(&stats[3], mx_all); }
  __syncthreads();
  uint4 *outD8 = reinterpret_cast<uint4 *>(outD), *outS8 = reinterpret_cast<uint4 *>(outS);
  for (int q = tid; q < n8; q += 1024) {                     // (codes past n inside the last 16 bytes: written as 0xFFFF, never read)
    const uint4 v = ids8[q];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t d[4] = {0, 0, 0, 0}, sp[4] = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t c = (w[u >> 1] >> (16 * (u & 1))) & 0xffffu;
      const bool live = q * 8 + u < n && c != 0xFFFFu;
      const uint32_t r = live ? hy_hist[c] : 0xFFFFu;
      d[u >> 1] |= r << (16 * (u & 1));
      sp[u >> 1] |= ((live && r == 0xFFFFu) ? c : 0xFFFFu) << (16 * (u & 1));
    }
    outD8[q] = make_uint4(d[0], d[1], d[2], d[3]);
    outS8[q] = make_uint4(sp[0], sp[1], sp[2], sp[3]);
  }
}

// ----------------------------------------------------------- bit planes --
// 64 sequences x all hash functions per workgroup.  Per chunk of 64 hash functions the 64 x 64
// uint16 code tile is staged in LDS; a wave then owns a sequence and its lane l holds the code of
// hash function h0 + l, so one 64-bit ballot per bit is plane p of two adjacent 32-hash groups.
// pl = planes kept per group (8, 12 or 16: every id is < 2^pl - 2).  Singletons (0xFFFF in idsT):
// row copy gets 2^pl - 2, column copy 2^pl - 1 -- they differ in plane 0 only.
// code_bits <= pl: width of the codes (the singleton codes are 2^code_bits - 2 / - 1; planes code_bits .. pl - 1 come out zero).
__global__ __launch_bounds__(256) void k_ids_to_planes(const uint16_t *__restrict__ idsT, int64_t ld_ids, int64_t n,
                                                       int n_hash, int pl, int code_bits, uint32_t *__restrict__ planes) {
  __shared__ uint16_t tile[64][66];
  const int64_t i0 = (int64_t)blockIdx.x * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const PlaneGeom pg = plane_geom(n, n_hash, pl);
  const int ngroup = (n_hash + 31) / 32;
  for (int h0 = 0; h0 < ngroup * 32; h0 += 64) {
    __syncthreads();
    for (int r = wave; r < 64; r += 4) {       // row r of the tile = hash function h0 + r, 64 sequences (128 B)
      const int h = h0 + r;
      const int64_t i = i0 + lane;
      tile[r][lane] = (h < n_hash && i < n) ? idsT[(int64_t)h * ld_ids + i] : (uint16_t)0;
    }
    __syncthreads();
    for (int sq = wave; sq < 64; sq += 4) {
      const int64_t i = i0 + sq;
      if (i >= n) break;                         // wave-uniform
      const uint32_t v = tile[lane][sq];
      const bool single = (v == 0xFFFFu) && (h0 + lane < n_hash);
      const uint32_t va = single ? (1u << code_bits) - 2u : v;
      const uint32_t vb = single ? (1u << code_bits) - 1u : v;
      uint32_t wa = 0, wb = 0;
      for (int p = 0; p < pl; ++p) {                       // pl is uniform
        const unsigned long long m = __ballot((va >> p) & 1u);
        const uint32_t half = (lane < 32) ? (uint32_t)m : (uint32_t)(m >> 32);
        wa = ((lane & 31) == p) ? half : wa;
      }
      {
        const unsigned long long m = __ballot(vb & 1u);    // plane 0 of the column copy
        const uint32_t half = (lane < 32) ? (uint32_t)m : (uint32_t)(m >> 32);
        wb = ((lane & 31) == 0) ? half : wa;
      }
      const int g = (h0 >> 5) + (lane >> 5);               // the group is the stage (pl <= 16)
      const int p = lane & 31;
      if (p < pl && g < ngroup) {
        const int64_t w = plane_unit_word(pg, i, g, p >> 2);
        planes[w + (p & 3)] = wa;
        planes[pg.copy_words + w + ((p & 3) ^ 1)] = wb;    // pair-swapped copy (see k_mh_compare)
        if (pl == 16) {                                    // padded twin for k_mh_compare_a16 (da_common.hpp)
          const int64_t ws = pad16_base_words(n, n_hash) + pad16_slot_word(n, n_hash, i, g);
          planes[ws + p] = wa;
          planes[ws + pad16_copy_words(n, n_hash) + (p ^ 1)] = wb;
        }
      }
    }
  }
}

// raw values, 32 planes per group = two 16-plane stages
__global__ __launch_bounds__(256) void k_sig_to_planes(const uint32_t *__restrict__ sig, int64_t ld_sig, int64_t n,
                                                       int n_hash, uint32_t *__restrict__ planes) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= n) return;
  const PlaneGeom pg = plane_geom(n, n_hash, 32);
  const int ngroup = (n_hash + 31) / 32;
  const uint32_t *row = sig + i * ld_sig;
  for (int h0 = 0; h0 < ngroup * 32; h0 += 64) {
    const int h = h0 + lane;
    const uint32_t v = (h < n_hash) ? row[h] : 0u;
    uint32_t w = 0;
#pragma unroll
    for (int p = 0; p < 32; ++p) {
      const unsigned long long m = __ballot((v >> p) & 1u);
      const uint32_t half = (lane < 32) ? (uint32_t)m : (uint32_t)(m >> 32);
      w = ((lane & 31) == p) ? half : w;
    }
    const int g = (h0 >> 5) + (lane >> 5);
    const int p = lane & 31;
    if (g < ngroup) {
      const int64_t u = plane_unit_word(pg, i, 2 * g + (p >> 4), (p & 15) >> 2);
      planes[u + (p & 3)] = w;
      planes[pg.copy_words + u + ((p & 3) ^ 1)] = w;
    }
  }
}

}  // namespace

static int64_t dict_ldT(int64_t n) { return ceil_div(n, 64) * 64; }

// carve-up of the workspace: all [n_hash][ldT]
struct DictWork { uint32_t *sigT, *sz, *si; uint16_t *idsT; int *status; };
static DictWork dict_work(void *d_work, int64_t n, int n_hash) {
  const size_t m = (size_t)n_hash * (size_t)dict_ldT(n);
  DictWork w;
  w.sigT = static_cast<uint32_t *>(d_work);
  w.sz = w.sigT + m;
  w.si = w.sz + m;
  w.idsT = reinterpret_cast<uint16_t *>(w.si + m);
  w.status = reinterpret_cast<int *>((reinterpret_cast<uintptr_t>(w.idsT + m) + 63) & ~(uintptr_t)63);
  return w;
}

size_t mh_planes_workspace_bytes(int64_t n, int n_hash) {
  if (n <= 0 || n_hash <= 0 || n > DA_DICT_MAX_N) return 256;
  const size_t ldT = (size_t)dict_ldT(n);
  return (size_t)n_hash * ldT * (4 + 4 + 4 + 2) + 256;   // sigT, sorted keys, their indices, codes, status
}

const uint16_t *mh_dictionary_codes(const void *d_work, int64_t n, int n_hash, int64_t *ld_ids) {
  *ld_ids = dict_ldT(n);
  return dict_work(const_cast<void *>(d_work), n, n_hash).idsT;
}

// Step 1 (transpose + dictionary): leaves the codes and two status words in the workspace.
// *d_status_out points at them; read them once the stream has drained, pick the plane count
// (mh_plane_bits_for) and run step 2.
int launch_mh_dictionary(const uint32_t *d_sig, int64_t ld_sig, int64_t n, int n_hash, void *d_work,
                         int **d_status_out, hipStream_t stream) {
  const int64_t ldT = dict_ldT(n);
  const DictWork w = dict_work(d_work, n, n_hash);
  uint32_t *sigT = w.sigT;
  uint16_t *idsT = w.idsT;
  int *status = w.status;
  *d_status_out = status;
  DA_HIP_TRY(hipMemsetAsync(status, 0, 2 * sizeof(int), stream));
  hipLaunchKernelGGL(k_sig_transpose, dim3((unsigned)ceil_div(n, 64), (unsigned)ceil_div(n_hash, 64)), dim3(256), 0,
                     stream, d_sig, ld_sig, n, n_hash, sigT, ldT);
  int R = (int)ceil_div(n, DK_PART_KEYS);
  if (R < 2) R = 2;
  hipLaunchKernelGGL(k_dictionary, dim3((unsigned)n_hash), dim3(DK_THREADS), 0, stream, sigT, ldT, n, R, w.sz, w.si, idsT, ldT,
                     status);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

// Heavy / rare split of the dictionary codes (k_hy_split): leaves the dense codes (ranks of the `keep` most frequent values per column) and the
// sparse codes (the other repeated values) in the workspace's sort scratch -- free once k_dictionary has run -- and two statistics in d_stats
// (four words, zeroed here): incidences among rare values, largest rare class, incidences among all repeated values, largest class.  The codes of
// launch_mh_dictionary stay as they are.
constexpr int HY_MAX_IDS = 32768;
int launch_mh_heavy_split(void *d_work, int64_t n, int n_hash, int max_ids, int keep, const uint16_t **d_dense_out, const uint16_t **d_sparse_out,
                          unsigned long long *d_stats, hipStream_t stream) {
  if (max_ids < 1 || max_ids > HY_MAX_IDS || keep < 1 || keep > 65534 || n > 0x7fffffffLL)
    return fail(DA_ERR_UNSUPPORTED, "heavy / rare split: too many repeated values per column");
  const int64_t ldT = dict_ldT(n);
  const DictWork w = dict_work(d_work, n, n_hash);
  uint16_t *dense = reinterpret_cast<uint16_t *>(w.sz), *sparse = reinterpret_cast<uint16_t *>(w.si);
  static std::atomic<uint64_t> attr_done;
  int dev = 0;
  DA_HIP_TRY(hipGetDevice(&dev));
  if (!((attr_done.load() >> (dev & 63)) & 1u)) {
    DA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_hy_split), hipFuncAttributeMaxDynamicSharedMemorySize, (HY_MAX_IDS + 1024) * 4));
    attr_done.fetch_or(1ull << (dev & 63));
  }
  DA_HIP_TRY(hipMemsetAsync(d_stats, 0, 32, stream));
  hipLaunchKernelGGL(k_hy_split, dim3((unsigned)n_hash), dim3(1024), (size_t)(max_ids + 1024) * 4, stream, w.idsT, ldT, (int)n, max_ids, keep, dense, sparse,
                     d_stats);
  DA_HIP_TRY(hipGetLastError());
  *d_dense_out = dense;
  *d_sparse_out = sparse;
  return DA_OK;
}

// planes per group that hold `max_ids` dense ids plus the two singleton codes
// 14 = the 16-plane operand with 14-bit codes (planes 14 and 15 zero): the hand-scheduled kernel then runs seven two-plane steps
int mh_plane_bits_for(int max_ids) {
  if (max_ids + 2 <= (1 << 8)) return 8;
  if (max_ids + 2 <= (1 << 12)) return 12;
  if (max_ids + 2 <= (1 << 14)) return 14;
  if (max_ids + 2 <= (1 << 15)) return 15;                    // (15: plane 15 zero -- the eighth step drops its second half)
  return 16;
}

// Step 2: codes -> bit planes, `plane_bits` (8 / 12 / 16) planes per group of 32 hash functions
int launch_ids_to_planes(const void *d_work, int64_t n, int n_hash, int plane_bits, uint32_t *d_planes,
                         hipStream_t stream, const uint16_t *d_codes) {
  const int64_t ldT = dict_ldT(n);
  const uint16_t *idsT = d_codes ? d_codes : dict_work(const_cast<void *>(d_work), n, n_hash).idsT;   // d_codes: the dense codes of launch_mh_heavy_split
  hipLaunchKernelGGL(k_ids_to_planes, dim3((unsigned)ceil_div(n, 64)), dim3(256), 0, stream, idsT, ldT, n, n_hash,
                     (plane_bits == 14 || plane_bits == 15) ? 16 : plane_bits, plane_bits, d_planes);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

int launch_sig_to_planes(const uint32_t *d_sig, int64_t ld_sig, int64_t n, int n_hash, uint32_t *d_planes,
                         hipStream_t stream) {
  hipLaunchKernelGGL(k_sig_to_planes, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, stream, d_sig, ld_sig, n, n_hash,
                     d_planes);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

}  // namespace da
