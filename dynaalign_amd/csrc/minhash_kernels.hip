// minhash_kernels.hip -- gfx950 kernels for the similarityMH hot path.
//
//   K1  k_minhash_signatures : k-shingle MurmurHash3 + per-sequence running min
//                              (reference src/minHash.cpp:140-157, hash :21-64)
//   K2  k_mh_compare         : all-pairs signature equality count + divide
//                              (reference src/minHash.cpp:160-178)
//
// Written for CDNA4 only: 64-wide wavefronts, 160 KiB LDS, 8 XCDs.  Integer
// work; no MFMA (an equality count is not a contraction).
#include "da_common.hpp"

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>
#include <type_traits>

namespace da {
namespace {

// ---------------------------------------------------------------- murmur3 --
// MurmurHash3_x86_32 split so that the seed-independent part of every 4-byte
// block ("scramble") is computed once per window position and shared by all
// n_hash seeds.  Reference: src/minHash.cpp:22-27 (constants), :34-41 (body),
// :43-54 (tail), :56-61 (finaliser).
__device__ __forceinline__ uint32_t rotl32(uint32_t v, int r) {
  return __builtin_rotateleft32(v, r);
}
__device__ __forceinline__ uint32_t mm3_scramble(uint32_t w) {
  w *= 0xcc9e2d51u;
  w = rotl32(w, 15);
  w *= 0x1b873593u;
  return w;
}
__device__ __forceinline__ uint32_t mm3_mix(uint32_t h, uint32_t kk) {
  h ^= kk;
  return rotl32(h, 13) * 5u + 0xe6546b64u;
}
__device__ __forceinline__ uint32_t mm3_final(uint32_t h, uint32_t len) {
  h ^= len;
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}

constexpr int K1_THREADS = 256;
constexpr int K1_CHUNK = 1024;  // window positions staged per pass

// One workgroup per sequence.  Pass structure for a chunk of window positions
// [c0, c0+CH):
//   1. all threads cooperatively read the residue bytes the chunk needs
//      (coalesced byte reads of the packed buffer) and write, per byte
//      position q, the scrambled 4-byte block starting at q (KK[q]) and the
//      scrambled (k&3)-byte tail starting at q (TT[q]) into LDS;
//   2. thread t owns hash functions h = t, t+256, ...: for each it walks the
//      chunk's windows, rebuilding murmur3(window, seed_h) from KK/TT (LDS
//      broadcast reads: every lane reads the same address) and keeps the
//      running min in a register -- no cross-lane reduction is needed because
//      lanes are spread over hash functions, not over windows.
// Window p of a k-byte shingle uses blocks KK[p], KK[p+4], ... (k/4 of them)
// and tail TT[p + 4*(k/4)].
template <bool K_IS_4>
__global__ __launch_bounds__(K1_THREADS) void k_minhash_signatures(
    const uint8_t *__restrict__ residues, const int64_t *__restrict__ offsets, int k, int n_hash,
    const uint32_t *__restrict__ seeds, uint32_t *__restrict__ sig, int64_t ld_sig) {
  extern __shared__ uint32_t lds_k1[];
  const int64_t seq = blockIdx.x;
  const int64_t beg = offsets[seq];
  const int64_t len = offsets[seq + 1] - beg;
  const int64_t nwin = (len >= k) ? (len - k + 1) : 0;  // src/minHash.cpp:98-103
  const int nblk = k >> 2, rem = k & 3;
  const int span = K1_CHUNK + k;  // byte positions whose KK/TT a chunk may touch
  uint32_t *KK = lds_k1;
  uint32_t *TT = lds_k1 + span;
  uint32_t *row = sig + seq * ld_sig;
  const uint8_t *s = residues + beg;

  if (nwin == 0)  // identity of min: UINT32_MAX (src/minHash.cpp:140)
    for (int h = threadIdx.x; h < n_hash; h += K1_THREADS) row[h] = 0xffffffffu;
  for (int64_t c0 = 0; c0 < nwin; c0 += K1_CHUNK) {
    const int cw = (int)((nwin - c0 < K1_CHUNK) ? (nwin - c0) : K1_CHUNK);
    const int need = cw + k - 1;  // byte positions c0 .. c0+need-1 are inside the sequence
    __syncthreads();
    for (int q = threadIdx.x; q < need; q += K1_THREADS) {
      const int64_t g = c0 + q;
      uint32_t b0 = s[g];
      uint32_t b1 = (g + 1 < len) ? s[g + 1] : 0u;
      uint32_t b2 = (g + 2 < len) ? s[g + 2] : 0u;
      uint32_t b3 = (g + 3 < len) ? s[g + 3] : 0u;
      KK[q] = mm3_scramble(b0 | (b1 << 8) | (b2 << 16) | (b3 << 24));
      if (!K_IS_4) {
        uint32_t t = b0;
        if (rem >= 2) t |= b1 << 8;
        if (rem == 3) t |= b2 << 16;
        TT[q] = mm3_scramble(t);
      }
    }
    __syncthreads();
    for (int h = threadIdx.x; h < n_hash; h += K1_THREADS) {
      const uint32_t seed = seeds[h];
      uint32_t best = (c0 == 0) ? 0xffffffffu : row[h];
      if (K_IS_4) {
#pragma unroll 4
        for (int p = 0; p < cw; ++p) {
          uint32_t v = mm3_final(mm3_mix(seed, KK[p]), 4u);
          best = v < best ? v : best;
        }
      } else {
        for (int p = 0; p < cw; ++p) {
          uint32_t hh = seed;
          for (int b = 0; b < nblk; ++b) hh = mm3_mix(hh, KK[p + 4 * b]);
          if (rem) hh ^= TT[p + 4 * nblk];
          uint32_t v = mm3_final(hh, (uint32_t)k);
          best = v < best ? v : best;
        }
      }
      row[h] = best;
    }
  }
}

// ---------------------------------------------------------------- compare --
// Bit-sliced equality.  On gfx950 integer compares, v_cndmask, v_addc and
// v_bcnt issue at half rate (16 lanes/clk/SIMD) while v_bitop3_b32 -- any
// 3-input boolean function -- issues at the full 32 lanes/clk (measured:
// profiles/r01_ubench_inst_rate.txt).  So instead of "compare two u32, add the
// carry" (2 half-rate ops per hash function) the kernel works on the 32x32
// bit-transposed signatures K1 emits: for a group of 32 hash functions,
//     d = OR over the 32 bit planes p of (A_p XOR B_p)
// has bit t set iff the two sequences DIFFER at hash function 32g+t, and costs
// one full-rate  v_bitop3 d, d, a, b  (d | (a ^ b))  per plane; one v_bcnt per
// group then adds popcount(d) to the pair's mismatch count.  That is ~1.03
// instructions per hash function instead of 2, all but 1/33 of them full rate.
// matches = n_hash - mismatches; padding hash functions are all-zero planes on
// both sides and never count as mismatches.
//
// Pair-space tile of 128 x 128 per 256-thread workgroup; each lane keeps an
// 8 x 8 block of pairs in registers (64 OR-accumulators + 64 mismatch counters).
// Plane words of HC = 32 hash functions (one group, 128 B per sequence) are
// staged through LDS per step.
constexpr int K2_TILE = 128;
constexpr int K2_GROUP = 32;         // hash functions per bit-plane group (= planes per group)
// planes per LDS stage and 16-byte segments per row per stage depend on the plane count PL of the
// kernel instance: PL = 32 -> 16 planes (half a group), PL = 16 / 12 / 8 -> the whole group
constexpr int K2_BAND = 8;           // tile rows per L2-resident band
constexpr int K2_THREADS = 256;

// Rows/cols owned by lane coordinate t (0..15):  32*g + 2*t + e, g=0..3, e=0..1.
// LDS slot order (k2_slot / k2_row_of_slot) and the operand's memory layout: da_common.hpp.
// A ds_read_b128 of the compute loop touches <= 8 consecutive slots at one segment.  With 2 or 3
// segments per slot (32- / 48-byte slots) those land in 8 different 16-byte bank groups by
// themselves; with 4 (64-byte slots) the segment index is XOR-swizzled with (slot >> 2) & 3.

// d | (a ^ b) as ONE full-rate v_bitop3_b32 (truth table over S0=0xF0,S1=0xCC,S2=0xAA:
// 0xF0 | (0xCC ^ 0xAA) = 0xF6).  Left to itself hipcc picks v_xor + v_or3 (half rate).
__device__ __forceinline__ uint32_t or_xor(uint32_t d, uint32_t a, uint32_t b) {
  return __builtin_amdgcn_bitop3_b32(d, a, b, 0xF6);
}

// 16-byte streaming store: the result matrix is written once and never read back by the kernel
typedef double da_double2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nt_store2(double *p, double a, double b) {
  da_double2_t v = {a, b};
  __builtin_nontemporal_store(v, reinterpret_cast<da_double2_t *>(p));   // (write-back stores wash the planes out of L2: +0.4 ms)
}

#ifndef K2_RING_DEPTH
#define K2_RING_DEPTH(PL) 3
#endif

struct TileId { int ti, tj; bool valid; };

// Enumeration of pair-space tiles.  Tile rows are grouped into bands of
// K2_BAND; inside a band tiles are visited column-major, so ~64 consecutive
// tile ids share 8 a-tiles (kept in the XCD's L2 for the whole band) and 8
// b-tiles.  Consecutive ids go to one XCD (the caller's blockIdx remap).
//   symmetric: only tiles with tj >= ti.   T = tiles per side (columns),
//   rect     : tile rows [0,TR) x tile cols [0,T).
// Everything here is wave-uniform and runs while the other workgroups of the CU are in their
// plane loops, so it is written to stay cheap: 32-bit integers, float estimates corrected by
// exact integer tests, no 64-bit division, no double sqrt.  Needs ntiles < 2^31, T < 2^20.
__host__ __device__ __forceinline__ int div_small(int l, int h) {   // l / h for 0 <= l < 2^24, 1 <= h <= 8
#ifdef __HIP_DEVICE_COMPILE__
  int q = (int)__fdividef((float)l, (float)h);
#else
  int q = (int)((float)l / (float)h);                                 // (host twin for tests: the estimate is corrected exactly either way)
#endif
  while (q * h > l) --q;
  while ((q + 1) * h <= l) ++q;
  return q;
}
__host__ __device__ __forceinline__ TileId decode_tile(int64_t L, int TR, int T, bool symmetric) {
  TileId o{0, 0, true};
  const int S = K2_BAND;
  if (!symmetric) {
    const int64_t per_band = (int64_t)S * T;
    int B = (int)(L / per_band);
    int64_t l = L - (int64_t)B * per_band;
    int r0 = B * S;
    int h = (TR - r0 < S) ? (TR - r0) : S;
    if (h <= 0) { o.valid = false; return o; }
    o.tj = (int)(l / h);
    o.ti = r0 + (int)(l - (int64_t)o.tj * h);
    o.valid = o.tj < T;
    return o;
  }
  // full bands: count(B) = c0 - S*S*B, c0 = S(S+1)/2 + (T-S)*S ; prefix(B) = B*c0 - S*S*B(B-1)/2
  const int nfull = T / S;  // bands with S rows
  const int c0 = S * (S + 1) / 2 + (T - S) * S;
  auto prefix = [&](int B) { return (int64_t)B * c0 - (int64_t)(S * S / 2) * B * (B - 1); };
  int B;
  if (nfull > 0 && L < prefix(nfull)) {
    // solve prefix(B) <= L: (S*S/2) B^2 - (c0 + S*S/2) B + L >= 0 -- float estimate, exact correction
    const float a = 0.5f * S * S, b = (float)c0 + a;
    const float disc = b * b - 4.0f * a * (float)L;
    B = (int)((b - sqrtf(disc > 0.0f ? disc : 0.0f)) / (2.0f * a));
    if (B < 0) B = 0;
    if (B > nfull - 1) B = nfull - 1;
    while (B > 0 && prefix(B) > L) --B;
    while (B + 1 <= nfull - 1 && prefix(B + 1) <= L) ++B;
  } else {
    B = nfull;  // the partial last band (or invalid)
  }
  const int r0 = B * S;
  const int h = (T - r0 < S) ? (T - r0) : S;
  if (h <= 0) { o.valid = false; return o; }
  int l = (int)(L - prefix(B < nfull ? B : nfull));   // < S * T
  const int tri = h * (h + 1) / 2;
  if (l < tri) {  // the diagonal super-tile: column q holds q+1 tiles
    int q = 0;
    while ((q + 1) * (q + 2) / 2 <= l) ++q;
    o.tj = r0 + q;
    o.ti = r0 + (l - q * (q + 1) / 2);
  } else {
    l -= tri;
    const int cq = div_small(l, h);
    o.tj = r0 + h + cq;
    o.ti = r0 + (l - cq * h);
  }
  o.valid = o.tj < T;
  return o;
}

__host__ __device__ inline int64_t count_tiles(int TR, int T, bool symmetric) {
  if (!symmetric) return (int64_t)TR * T;
  return (int64_t)T * (T + 1) / 2;
}

// Phase stamps for tools/k2_timeline.py; only in a -DDA_K2_TIMING build (never the shipped library).
#ifdef DA_K2_TIMING
__device__ unsigned long long *g_k2_timing = nullptr;   // [blocks][8]: t_entry, t_loop, t_loop_end, t_exit, hw_id, xcc_id
#define K2_STAMP(i) do { if (g_k2_timing && threadIdx.x == 0) g_k2_timing[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#define K2_STAMP_HW() do { if (g_k2_timing && threadIdx.x == 0) { \
    unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); \
    g_k2_timing[(size_t)blockIdx.x * 8 + 4] = hw; g_k2_timing[(size_t)blockIdx.x * 8 + 5] = xcc; } } while (0)
#else
#define K2_STAMP(i) do { } while (0)
#define K2_STAMP_HW() do { } while (0)
#endif

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

#ifdef DA_K2_DEBUG
__device__ uint32_t *g_k2_debug = nullptr;
#endif
// Which tiles the hand-scheduled kernel below computes: strictly above the diagonal, wholly inside the matrix,
// output aligned for its wide stores.  Everything else stays with k_mh_compare (only_edge = 1).
__device__ __forceinline__ bool a12_takes(int ti, int tj, int64_t n, int64_t ld, const void *out, bool f64) {
  return ti != tj && (int64_t)(ti + 1) * K2_TILE <= n && (int64_t)(tj + 1) * K2_TILE <= n && (ld & 1) == 0 &&
         (reinterpret_cast<uintptr_t>(out) & (f64 ? 15 : 3)) == 0;
}

// ---- 12-plane compare with a hand-allocated stage loop ------------------------------------------
// hipcc needs ~30 VGPRs more than the loop strictly does, which pins k_mh_compare at 168 VGPRs = 3 waves per SIMD
// -- an odd wave count, which costs the gfx950 VALU a quarter of its issue slots (tools/ubench/k2_inner2).  Here the
// stage loop (DMA issue, counted waits, barriers, plane loop with 8-byte operands, popcounts) is ONE asm statement
// with a fixed register map (tools/gen_k2_asm.py -> k2_loop_p12.inc): 116 + 10 VGPRs, so 4 workgroups per CU =
// 4 waves per SIMD.  The C++ around it decodes the tile, hands five per-lane values over in v120..v124, reads the
// 32 packed mismatch counters back from LDS and stores the tile like k_mh_compare's straight-line epilogue.
// Symmetric mode, interior off-diagonal tiles only (a12_takes).
#ifndef K2_LOOP_INC
#define K2_LOOP_INC "k2_loop_p12.inc"
#endif
#ifndef K2_PRO_PRIO
#define K2_PRO_PRIO 0     // wave priority of the tile prologue (decode + address arithmetic before the stage loop)
#endif
constexpr int K2_A12_TABLE_MAX = 3 * 2 * K2_TILE * 3 * 16 / 8;   // doubles that fit the 12-plane kernel's ring (3 stages x 256 rows x 48 B)
template <bool F64>
__global__ __launch_bounds__(K2_THREADS, 4) void k_mh_compare_a12(const uint32_t *__restrict__ planes, int64_t n, int n_hash,
                                                                  void *__restrict__ out_v, int64_t ld, int64_t ntiles,
                                                                  int64_t per_xcd) {
  constexpr int PL = 12, SEGS = 3, STAGE_UNITS = 2 * K2_TILE * SEGS;
  __shared__ __attribute__((aligned(16))) uint4 lds_ab[3 * STAGE_UNITS];   // 36 KiB ring; afterwards counters, then the ratio table
  static_assert(sizeof(lds_ab) / (sizeof(double)) == K2_A12_TABLE_MAX, "launch_mh_compare's table guard must match the ring size");
  if (K2_PRO_PRIO) __builtin_amdgcn_s_setprio(K2_PRO_PRIO);
  K2_STAMP(0);
  K2_STAMP_HW();
  const int64_t bid = blockIdx.x;
  const int T = (int)((n + K2_TILE - 1) / K2_TILE);
  const int64_t L = (bid & 7) * per_xcd + (bid >> 3);
  if (L >= ntiles) return;
  const TileId tl = decode_tile(L, T, T, true);
  if (!tl.valid || !a12_takes(tl.ti, tl.tj, n, ld, out_v, F64)) return;
  if (!F64) K2_STAMP(6);                                      // (timing build, uint16 kind: prologue split)
  const int64_t I0 = (int64_t)tl.ti * K2_TILE, J0 = (int64_t)tl.tj * K2_TILE;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tx = ((wave & 1) << 3) + (lane & 7), ty = ((wave >> 1) << 3) + (lane >> 3);
  const PlaneGeom pg = plane_geom(n, n_hash, PL);
  // DMA source of this lane in the wave's first instruction of stage 0; instructions q = 1, 2 read 1 KiB and 2 KiB
  // further (the operand is stored in staging order), stage s reads 128 * 12 words further
  const int u = wave * SEGS * 64 + lane, sl0 = u / SEGS, sl = sl0 & 127;
  const uint32_t *src = planes + (sl0 < 128 ? 0 : pg.copy_words) +
                        plane_unit_word(pg, (sl0 < 128 ? I0 : J0) + k2_row_of_slot(sl), 0, u - sl0 * SEGS);
  if (!F64) K2_STAMP(7);
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_t *)lds_ab);
  const uint32_t a_off = lds_base + (uint32_t)(ty * SEGS * 16);
  const uint32_t b_off = lds_base + (uint32_t)((K2_TILE * SEGS + tx * SEGS) * 16);
  const uint32_t src_lo = (uint32_t)reinterpret_cast<uintptr_t>(src), src_hi = (uint32_t)(reinterpret_cast<uintptr_t>(src) >> 32);
  const uint32_t wb = (uint32_t)tid * 4u;
  const uint32_t nstage = (uint32_t)((n_hash + K2_GROUP - 1) / K2_GROUP), stage_bytes = 128u * PL * 4u;
  const uint32_t wave_id = __builtin_amdgcn_readfirstlane((uint32_t)wave);
  const uint32_t src_lo_u = __builtin_amdgcn_readfirstlane(src_lo), src_hi_u = __builtin_amdgcn_readfirstlane(src_hi);   // lane 0's source
  K2_STAMP(1);
  // the lane's 32 packed mismatch counters (columns 2j, 2j+1 per register) leave the block in v64..v95
  uint32_t mis[8][4];
  uint32_t tid_after;
  {
    register uint32_t r120 asm("v120") = a_off;
    register uint32_t r121 asm("v121") = b_off;
    register uint32_t r122 asm("v122") = src_lo;
    register uint32_t r123 asm("v123") = src_hi;
    register uint32_t r124 asm("v124") = wb;
#define K2_CNT(i) register uint32_t c##i asm("v" #i);
    K2_CNT(64) K2_CNT(65) K2_CNT(66) K2_CNT(67) K2_CNT(68) K2_CNT(69) K2_CNT(70) K2_CNT(71) K2_CNT(72) K2_CNT(73) K2_CNT(74)
    K2_CNT(75) K2_CNT(76) K2_CNT(77) K2_CNT(78) K2_CNT(79) K2_CNT(80) K2_CNT(81) K2_CNT(82) K2_CNT(83) K2_CNT(84) K2_CNT(85)
    K2_CNT(86) K2_CNT(87) K2_CNT(88) K2_CNT(89) K2_CNT(90) K2_CNT(91) K2_CNT(92) K2_CNT(93) K2_CNT(94) K2_CNT(95)
#undef K2_CNT
    asm volatile(
#include K2_LOOP_INC
        : "+v"(r122), "+v"(r123),                                // the block reuses them as an operand buffer
          "=v"(c64), "=v"(c65), "=v"(c66), "=v"(c67), "=v"(c68), "=v"(c69), "=v"(c70), "=v"(c71), "=v"(c72), "=v"(c73), "=v"(c74),
          "=v"(c75), "=v"(c76), "=v"(c77), "=v"(c78), "=v"(c79), "=v"(c80), "=v"(c81), "=v"(c82), "=v"(c83), "=v"(c84), "=v"(c85),
          "=v"(c86), "=v"(c87), "=v"(c88), "=v"(c89), "=v"(c90), "=v"(c91), "=v"(c92), "=v"(c93), "=v"(c94), "=v"(c95)
        : [lb] "s"(lds_base), [ns] "s"(nstage), [st] "s"(stage_bytes), [wv] "s"(wave_id), [sl] "s"(src_lo_u), [sh] "s"(src_hi_u),
          "v"(r120), "v"(r121), "v"(r124)
        : "memory", "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "v125",   // m0 is saved in s47 and restored by the block
          "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",
          "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39",
          "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59",
          "v60", "v61", "v62", "v63", "v96", "v97", "v98", "v99",
          "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116",
          "v117", "v118", "v119");
    // everything lane-dependent the epilogue needs is re-derived from v124 (4 * thread id), which survives the block:
    // nothing per-lane has to live across it (the block clobbers all but four VGPRs; hipcc spilled to scratch otherwise)
    asm volatile("" : "+v"(r124));
    tid_after = r124 >> 2;
    const uint32_t cnt[32] = {c64, c65, c66, c67, c68, c69, c70, c71, c72, c73, c74, c75, c76, c77, c78, c79,
                              c80, c81, c82, c83, c84, c85, c86, c87, c88, c89, c90, c91, c92, c93, c94, c95};
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) mis[r][c] = cnt[4 * r + c];
  }
  K2_STAMP(2);
#ifdef DA_K2_DEBUG
  if (g_k2_debug && tl.ti == 0 && tl.tj == 1) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(lds_ab);
    for (int i = tid; i < 3 * STAGE_UNITS * 4; i += K2_THREADS) g_k2_debug[i] = w[i];
  }
#endif
#ifdef K2_NO_STORES   // experiment: how long does the kernel take without its epilogue (tools/k2_variants.sh)?
  if (n_hash > 0) { if (mis[0][0] == 0xdeadbeefu) reinterpret_cast<uint32_t *>(out_v)[0] = tid_after; return; }
#endif
  const uint32_t nn = (uint32_t)n_hash * 0x10001u;             // two match counts per register (no borrow: each <= n_hash)
  // lane coordinates again, from the value that crossed the block (same formulas as above)
  const int tid_e = (int)tid_after, wave_e = tid_e >> 6, lane_e = tid_e & 63;
  const int tx_e = ((wave_e & 1) << 3) + (lane_e & 7), ty_e = ((wave_e >> 1) << 3) + (lane_e >> 3);
#define tx tx_e
#define ty ty_e
  if (F64) {
    double *ratio = reinterpret_cast<double *>(lds_ab);
    __syncthreads();                                           // everyone has left the ring: the area becomes the table
    for (int c = tid_e; c <= n_hash; c += K2_THREADS) ratio[c] = (double)c / (double)n_hash;   // src/minHash.cpp:174
    __syncthreads();
    K2_STAMP(6);
    K2_STAMP(7);
    const char *tb = reinterpret_cast<const char *>(ratio);
    double *out = reinterpret_cast<double *>(out_v);
#pragma unroll
    for (int g = 0; g < 4; ++g) {                              // two of the lane's rows at a time keeps the epilogue in 128 VGPRs
      double v0[8], v1[8];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint32_t m0 = nn - mis[2 * g][c], m1 = nn - mis[2 * g + 1][c];
        v0[2 * c] = *reinterpret_cast<const double *>(tb + ((m0 << 3) & 0x7fff8u));
        v0[2 * c + 1] = *reinterpret_cast<const double *>(tb + ((m0 >> 13) & 0x7fff8u));
        v1[2 * c] = *reinterpret_cast<const double *>(tb + ((m1 << 3) & 0x7fff8u));
        v1[2 * c + 1] = *reinterpret_cast<const double *>(tb + ((m1 >> 13) & 0x7fff8u));
      }
      double *orow = out + (I0 + 32 * g + 2 * ty) * ld + (J0 + 2 * tx);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        nt_store2(orow + 32 * q, v0[2 * q], v0[2 * q + 1]);
        nt_store2(orow + ld + 32 * q, v1[2 * q], v1[2 * q + 1]);
      }
#pragma unroll
      for (int c = 0; c < 8; ++c)                              // mirrored store (src/minHash.cpp:176)
        nt_store2(out + (J0 + 32 * (c >> 1) + 2 * tx + (c & 1)) * ld + (I0 + 32 * g + 2 * ty), v0[c], v1[c]);
    }
  } else {
    uint16_t *out = reinterpret_cast<uint16_t *>(out_v);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      uint16_t *orow = out + (I0 + 32 * (r >> 1) + 2 * ty + (r & 1)) * ld + (J0 + 2 * tx);
#pragma unroll
      for (int g = 0; g < 4; ++g) *reinterpret_cast<uint32_t *>(orow + 32 * g) = nn - mis[r][g];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      uint16_t *orow = out + (J0 + 32 * (c >> 1) + 2 * tx + (c & 1)) * ld + (I0 + 2 * ty);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint32_t lo = mis[2 * g][c >> 1], hi = mis[2 * g + 1][c >> 1];
        const uint32_t pk = (c & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
        *reinterpret_cast<uint32_t *>(orow + 32 * g) = nn - pk;
      }
    }
  }
  K2_STAMP(3);
#undef tx
#undef ty
}

// ---- the same hand-scheduled 12-plane loop for the SHARD / row-block modes (uint16 output, no mirror): VERDICT r2 item 2's last part ----
// Tile geometry of k_mh_compare's non-symmetric mode (cyclic tile rows of a rank, folded shard rows, upper_only); takes the tiles that lie
// completely inside the matrix and the row range and whose 4-byte stores are aligned (s12_takes) -- the compiled kernel runs with only_edge = 1
// on the same grid and returns at once for those.
__device__ __forceinline__ bool s12_takes(int64_t I0, int64_t J0, int64_t n, int64_t row_end, int64_t ld, int64_t Jloc, const void *out) {
  return I0 != J0 &&                                             // a diagonal tile: the general kernel forces count(i, i) = n_hash (singleton codes never match)
         I0 + K2_TILE <= (row_end < n ? row_end : n) && J0 + K2_TILE <= n && (ld & 1) == 0 && (Jloc & 1) == 0 &&
         (reinterpret_cast<uintptr_t>(out) & 3) == 0;
}
struct RectTile { int ti, tj; bool valid; int64_t I0, J0, Iloc, Jloc; };
__device__ __forceinline__ RectTile decode_rect_tile(int64_t bid, int64_t n, int64_t row_begin, int64_t row_end, int tile_stride, int upper_only, int TR,
                                                     int fold_q, int64_t fold_w, int band) {
  RectTile t{0, 0, false, 0, 0, 0, 0};
  const int T = (int)((n + K2_TILE - 1) / K2_TILE);
  const uint32_t per_band = (uint32_t)band * (uint32_t)T;
  const uint32_t k = (uint32_t)(bid >> 3);
  const uint32_t kb = k / per_band;
  const int r0 = ((int)(bid & 7) + 8 * (int)kb) * band;
  const int l = (int)(k - kb * per_band);
  const int h = (TR - r0 < band) ? (TR - r0) : band;
  if (h <= 0) return t;
  t.tj = __builtin_amdgcn_readfirstlane(div_small(l, h));               // wave-uniform by construction: scalars, so that nothing of the tile's
  t.ti = __builtin_amdgcn_readfirstlane(r0 + (l - t.tj * h));           // geometry sits in a VGPR across the loop block (it clobbers the file)
  if (t.tj >= T) return t;
  t.I0 = row_begin + (int64_t)t.ti * tile_stride * K2_TILE;
  t.J0 = (int64_t)t.tj * K2_TILE;
  t.Iloc = (int64_t)t.ti * K2_TILE - t.I0;
  t.Jloc = 0;
  if (fold_q > 0) {
    const int q = t.ti;
    const bool front = q <= fold_q - 1 - q;
    t.Iloc = (int64_t)(front ? q : fold_q - 1 - q) * K2_TILE - t.I0;
    t.Jloc = front ? -t.I0 : shard_back(fold_w, n);
  }
  if (t.I0 >= row_end || t.I0 >= n) return t;
  if (upper_only && t.J0 + K2_TILE <= t.I0) return t;
  t.valid = true;
  return t;
}
__global__ __launch_bounds__(K2_THREADS, 4) void k_mh_compare_s12(const uint32_t *__restrict__ planes, int64_t n, int n_hash, int64_t row_begin,
                                                                  int64_t row_end, int tile_stride, int upper_only, int TR, uint16_t *__restrict__ out,
                                                                  int64_t ld, int fold_q, int64_t fold_w, int band) {
  constexpr int PL = 12, SEGS = 3, STAGE_UNITS = 2 * K2_TILE * SEGS;
  __shared__ __attribute__((aligned(16))) uint4 lds_ab[3 * STAGE_UNITS];   // 36 KiB ring
  const RectTile rt = decode_rect_tile(blockIdx.x, n, row_begin, row_end, tile_stride, upper_only, TR, fold_q, fold_w, band);
  if (!rt.valid || !s12_takes(rt.I0, rt.J0, n, row_end, ld, rt.Jloc, out)) return;
  const int64_t I0 = rt.I0, J0 = rt.J0;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tx = ((wave & 1) << 3) + (lane & 7), ty = ((wave >> 1) << 3) + (lane >> 3);
  const PlaneGeom pg = plane_geom(n, n_hash, PL);
  // DMA source of this lane in the wave's first instruction of stage 0; instructions q = 1, 2 read 1 KiB and 2 KiB
  // further (the operand is stored in staging order), stage s reads 128 * 12 words further
  const int u = wave * SEGS * 64 + lane, sl0 = u / SEGS, sl = sl0 & 127;
  const uint32_t *src = planes + (sl0 < 128 ? 0 : pg.copy_words) +
                        plane_unit_word(pg, (sl0 < 128 ? I0 : J0) + k2_row_of_slot(sl), 0, u - sl0 * SEGS);
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_t *)lds_ab);
  const uint32_t a_off = lds_base + (uint32_t)(ty * SEGS * 16);
  const uint32_t b_off = lds_base + (uint32_t)((K2_TILE * SEGS + tx * SEGS) * 16);
  const uint32_t src_lo = (uint32_t)reinterpret_cast<uintptr_t>(src), src_hi = (uint32_t)(reinterpret_cast<uintptr_t>(src) >> 32);
  const uint32_t wb = (uint32_t)tid * 4u;
  const uint32_t nstage = (uint32_t)((n_hash + K2_GROUP - 1) / K2_GROUP), stage_bytes = 128u * PL * 4u;
  const uint32_t wave_id = __builtin_amdgcn_readfirstlane((uint32_t)wave);
  const uint32_t src_lo_u = __builtin_amdgcn_readfirstlane(src_lo), src_hi_u = __builtin_amdgcn_readfirstlane(src_hi);   // lane 0's source
  // the lane's 32 packed mismatch counters (columns 2j, 2j+1 per register) leave the block in v64..v95
  uint32_t mis[8][4];
  uint32_t tid_after;
  {
    register uint32_t r120 asm("v120") = a_off;
    register uint32_t r121 asm("v121") = b_off;
    register uint32_t r122 asm("v122") = src_lo;
    register uint32_t r123 asm("v123") = src_hi;
    register uint32_t r124 asm("v124") = wb;
#define K2_CNT(i) register uint32_t c##i asm("v" #i);
    K2_CNT(64) K2_CNT(65) K2_CNT(66) K2_CNT(67) K2_CNT(68) K2_CNT(69) K2_CNT(70) K2_CNT(71) K2_CNT(72) K2_CNT(73) K2_CNT(74)
    K2_CNT(75) K2_CNT(76) K2_CNT(77) K2_CNT(78) K2_CNT(79) K2_CNT(80) K2_CNT(81) K2_CNT(82) K2_CNT(83) K2_CNT(84) K2_CNT(85)
    K2_CNT(86) K2_CNT(87) K2_CNT(88) K2_CNT(89) K2_CNT(90) K2_CNT(91) K2_CNT(92) K2_CNT(93) K2_CNT(94) K2_CNT(95)
#undef K2_CNT
    asm volatile(
#include K2_LOOP_INC
        : "+v"(r122), "+v"(r123),                                // the block reuses them as an operand buffer
          "=v"(c64), "=v"(c65), "=v"(c66), "=v"(c67), "=v"(c68), "=v"(c69), "=v"(c70), "=v"(c71), "=v"(c72), "=v"(c73), "=v"(c74),
          "=v"(c75), "=v"(c76), "=v"(c77), "=v"(c78), "=v"(c79), "=v"(c80), "=v"(c81), "=v"(c82), "=v"(c83), "=v"(c84), "=v"(c85),
          "=v"(c86), "=v"(c87), "=v"(c88), "=v"(c89), "=v"(c90), "=v"(c91), "=v"(c92), "=v"(c93), "=v"(c94), "=v"(c95)
        : [lb] "s"(lds_base), [ns] "s"(nstage), [st] "s"(stage_bytes), [wv] "s"(wave_id), [sl] "s"(src_lo_u), [sh] "s"(src_hi_u),
          "v"(r120), "v"(r121), "v"(r124)
        : "memory", "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "v125",   // m0 is saved in s47 and restored by the block
          "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",
          "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39",
          "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59",
          "v60", "v61", "v62", "v63", "v96", "v97", "v98", "v99",
          "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116",
          "v117", "v118", "v119");
    // everything lane-dependent the epilogue needs is re-derived from v124 (4 * thread id), which survives the block:
    // nothing per-lane has to live across it (the block clobbers all but four VGPRs; hipcc spilled to scratch otherwise)
    asm volatile("" : "+v"(r124));
    tid_after = r124 >> 2;
    const uint32_t cnt[32] = {c64, c65, c66, c67, c68, c69, c70, c71, c72, c73, c74, c75, c76, c77, c78, c79,
                              c80, c81, c82, c83, c84, c85, c86, c87, c88, c89, c90, c91, c92, c93, c94, c95};
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) mis[r][c] = cnt[4 * r + c];
  }
  const uint32_t nn = (uint32_t)n_hash * 0x10001u;             // two match counts per register (no borrow: each <= n_hash)
  const int tid_e = (int)tid_after, wave_e = tid_e >> 6, lane_e = tid_e & 63;
  const int tx_e = ((wave_e & 1) << 3) + (lane_e & 7), ty_e = ((wave_e >> 1) << 3) + (lane_e >> 3);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    uint16_t *orow = out + (I0 + 32 * (r >> 1) + 2 * ty_e + (r & 1) + rt.Iloc) * ld + (rt.Jloc + J0 + 2 * tx_e);
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<uint32_t *>(orow + 32 * g) = nn - mis[r][g];
  }
}

// ---- 16 code planes (uniform-like data) with the same hand-allocated loop ---------------------------------------
// k_mh_compare<.., 16> needs 168 VGPRs = 3 waves per SIMD (34.6 ms at N = 100k).  The generated block (K2ASM_PLANES=16,
// k2_loop_p16.inc: 8 two-plane steps per stage) reads the PADDED twin of the operand (da_common.hpp: 80-byte slots), stages
// it through a ring of TWO 20 KiB stages (40 KiB: four workgroups per CU) with five 1 KiB DMA pieces per wave and stage from a
// wave-uniform SGPR base, and hands the counters over in v64..v95 like k_mh_compare_a12.  Symmetric mode, interior
// off-diagonal tiles (a12_takes); diagonal / border tiles stay with k_mh_compare<.., 16> on the regular copies.
#ifndef K2_LOOP_INC_14
#define K2_LOOP_INC_14 "k2_loop_p14.inc"   // the same block with seven two-plane steps: 14-bit codes, planes 14 / 15 of the operand are zero
#endif
#ifndef K2_LOOP_INC_15
#define K2_LOOP_INC_15 "k2_loop_p15.inc"   // eight steps, the last one on plane 14 only: 15-bit codes
#endif
#ifndef K2_LOOP_INC_16
#define K2_LOOP_INC_16 "k2_loop_p16.inc"
#endif
constexpr int K2_A16_TABLE_MAX = 2 * 2 * K2_TILE * 80 / 8;   // doubles that fit the 40 KiB ring
template <bool F64, int CODE_BITS = 16>
__global__ __launch_bounds__(K2_THREADS, 4) void k_mh_compare_a16(const uint32_t *__restrict__ planes, int64_t n, int n_hash,
                                                                  void *__restrict__ out_v, int64_t ld, int64_t ntiles,
                                                                  int64_t per_xcd) {
  constexpr int SLOT = 80, STAGE_BYTES = 2 * K2_TILE * SLOT;
  __shared__ __attribute__((aligned(16))) uint4 lds_ab[2 * STAGE_BYTES / 16];   // 40 KiB ring; afterwards the ratio table
  static_assert(sizeof(lds_ab) / (sizeof(double)) == K2_A16_TABLE_MAX, "launch_mh_compare's table guard must match the ring size");
  const int64_t bid = blockIdx.x;
  const int T = (int)((n + K2_TILE - 1) / K2_TILE);
  const int64_t L = (bid & 7) * per_xcd + (bid >> 3);
  if (L >= ntiles) return;
  const TileId tl = decode_tile(L, T, T, true);
  if (!tl.valid || !a12_takes(tl.ti, tl.tj, n, ld, out_v, F64)) return;
  const int64_t I0 = (int64_t)tl.ti * K2_TILE, J0 = (int64_t)tl.tj * K2_TILE;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tx0 = ((wave & 1) << 3) + (lane & 7), ty0 = ((wave >> 1) << 3) + (lane >> 3);
  // wave-uniform DMA source: wave w stages LDS slots [64w, 64w + 64) = rows (w < 2, row copy) or columns (w >= 2, column copy)
  const uint32_t wave_u = __builtin_amdgcn_readfirstlane((uint32_t)wave);
  const uint32_t nstage = (uint32_t)((n_hash + K2_GROUP - 1) / K2_GROUP), stage_bytes = (uint32_t)(K2_TILE * SLOT);
  const uint64_t src = reinterpret_cast<uint64_t>(planes) +
                       4u * (uint64_t)(pad16_base_words(n, n_hash) + (wave_u >= 2 ? pad16_copy_words(n, n_hash) : 0)) +
                       (uint64_t)(wave_u >= 2 ? tl.tj : tl.ti) * nstage * stage_bytes + (wave_u & 1u) * (64u * SLOT);
  const uint32_t sl = __builtin_amdgcn_readfirstlane((uint32_t)src), sh = __builtin_amdgcn_readfirstlane((uint32_t)(src >> 32));
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_t *)lds_ab);
  uint32_t mis[8][4];
  uint32_t tid_after;
  {
    register uint32_t r120 asm("v120") = lds_base + (uint32_t)(ty0 * SLOT);
    register uint32_t r121 asm("v121") = lds_base + (uint32_t)((K2_TILE + tx0) * SLOT);
    register uint32_t r124 asm("v124") = (uint32_t)tid * 4u;
#define K2_CNT(i) register uint32_t c##i asm("v" #i);
    K2_CNT(64) K2_CNT(65) K2_CNT(66) K2_CNT(67) K2_CNT(68) K2_CNT(69) K2_CNT(70) K2_CNT(71) K2_CNT(72) K2_CNT(73) K2_CNT(74)
    K2_CNT(75) K2_CNT(76) K2_CNT(77) K2_CNT(78) K2_CNT(79) K2_CNT(80) K2_CNT(81) K2_CNT(82) K2_CNT(83) K2_CNT(84) K2_CNT(85)
    K2_CNT(86) K2_CNT(87) K2_CNT(88) K2_CNT(89) K2_CNT(90) K2_CNT(91) K2_CNT(92) K2_CNT(93) K2_CNT(94) K2_CNT(95)
#undef K2_CNT
#define K2_A16_OPERANDS                                                                                                                      \
        : "=v"(c64), "=v"(c65), "=v"(c66), "=v"(c67), "=v"(c68), "=v"(c69), "=v"(c70), "=v"(c71), "=v"(c72), "=v"(c73), "=v"(c74),  \
          "=v"(c75), "=v"(c76), "=v"(c77), "=v"(c78), "=v"(c79), "=v"(c80), "=v"(c81), "=v"(c82), "=v"(c83), "=v"(c84), "=v"(c85),  \
          "=v"(c86), "=v"(c87), "=v"(c88), "=v"(c89), "=v"(c90), "=v"(c91), "=v"(c92), "=v"(c93), "=v"(c94), "=v"(c95)  \
        : [lb] "s"(lds_base), [ns] "s"(nstage), [st] "s"(stage_bytes), [wv] "s"(wave_u), [sl] "s"(sl), [sh] "s"(sh),  \
          "v"(r120), "v"(r121), "v"(r124)  \
        : "memory", "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "v125",  \
          "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",  \
          "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39",  \
          "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59",  \
          "v60", "v61", "v62", "v63", "v96", "v97", "v98", "v99",  \
          "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116",  \
          "v117", "v118", "v119", "v122", "v123"
    if (CODE_BITS == 14) {
      asm volatile(
#include K2_LOOP_INC_14
          K2_A16_OPERANDS);
    } else if (CODE_BITS == 15) {
      asm volatile(
#include K2_LOOP_INC_15
          K2_A16_OPERANDS);
    } else {
      asm volatile(
#include K2_LOOP_INC_16
          K2_A16_OPERANDS);
    }
#undef K2_A16_OPERANDS
    asm volatile("" : "+v"(r124));                               // lane ids are re-derived from the value that crossed the block
    tid_after = r124 >> 2;
    const uint32_t cnt[32] = {c64, c65, c66, c67, c68, c69, c70, c71, c72, c73, c74, c75, c76, c77, c78, c79,
                              c80, c81, c82, c83, c84, c85, c86, c87, c88, c89, c90, c91, c92, c93, c94, c95};
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) mis[r][c] = cnt[4 * r + c];
  }
  const uint32_t nn = (uint32_t)n_hash * 0x10001u;             // two match counts per register (no borrow: each <= n_hash)
  const int tid_e = (int)tid_after, wave_e = tid_e >> 6, lane_e = tid_e & 63;
  const int tx = ((wave_e & 1) << 3) + (lane_e & 7), ty = ((wave_e >> 1) << 3) + (lane_e >> 3);
  if (F64) {
    double *ratio = reinterpret_cast<double *>(lds_ab);
    __syncthreads();                                           // everyone has left the ring: the area becomes the table
    for (int c = tid_e; c <= n_hash; c += K2_THREADS) ratio[c] = (double)c / (double)n_hash;   // src/minHash.cpp:174
    __syncthreads();
    const char *tb = reinterpret_cast<const char *>(ratio);
    double *out = reinterpret_cast<double *>(out_v);
#pragma unroll
    for (int g = 0; g < 4; ++g) {                              // two of the lane's rows at a time keeps the epilogue in 128 VGPRs
      double v0[8], v1[8];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint32_t m0 = nn - mis[2 * g][c], m1 = nn - mis[2 * g + 1][c];
        v0[2 * c] = *reinterpret_cast<const double *>(tb + ((m0 << 3) & 0x7fff8u));
        v0[2 * c + 1] = *reinterpret_cast<const double *>(tb + ((m0 >> 13) & 0x7fff8u));
        v1[2 * c] = *reinterpret_cast<const double *>(tb + ((m1 << 3) & 0x7fff8u));
        v1[2 * c + 1] = *reinterpret_cast<const double *>(tb + ((m1 >> 13) & 0x7fff8u));
      }
      double *orow = out + (I0 + 32 * g + 2 * ty) * ld + (J0 + 2 * tx);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        nt_store2(orow + 32 * q, v0[2 * q], v0[2 * q + 1]);
        nt_store2(orow + ld + 32 * q, v1[2 * q], v1[2 * q + 1]);
      }
#pragma unroll
      for (int c = 0; c < 8; ++c)                              // mirrored store (src/minHash.cpp:176)
        nt_store2(out + (J0 + 32 * (c >> 1) + 2 * tx + (c & 1)) * ld + (I0 + 32 * g + 2 * ty), v0[c], v1[c]);
    }
  } else {
    uint16_t *out = reinterpret_cast<uint16_t *>(out_v);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      uint16_t *orow = out + (I0 + 32 * (r >> 1) + 2 * ty + (r & 1)) * ld + (J0 + 2 * tx);
#pragma unroll
      for (int g = 0; g < 4; ++g) *reinterpret_cast<uint32_t *>(orow + 32 * g) = nn - mis[r][g];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      uint16_t *orow = out + (J0 + 32 * (c >> 1) + 2 * tx + (c & 1)) * ld + (I0 + 2 * ty);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint32_t lo = mis[2 * g][c >> 1], hi = mis[2 * g + 1][c >> 1];
        const uint32_t pk = (c & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
        *reinterpret_cast<uint32_t *>(orow + 32 * g) = nn - pk;
      }
    }
  }
}

// ---- the 16-plane loop for the SHARD / row-block modes (uint16 output, no mirror): k_mh_compare_s12's tile geometry in front of
// k_mh_compare_a16's block (round 4: the per-rank compare of inputs whose column dictionaries need 14 - 16 code bits -- uniform
// peptides -- ran the compiled kernel).  Takes what s12_takes takes; the compiled kernel keeps diagonal / border tiles (only_edge).
template <int CODE_BITS>
__global__ __launch_bounds__(K2_THREADS, 4) void k_mh_compare_s16(const uint32_t *__restrict__ planes, int64_t n, int n_hash, int64_t row_begin,
                                                                  int64_t row_end, int tile_stride, int upper_only, int TR, uint16_t *__restrict__ out,
                                                                  int64_t ld, int fold_q, int64_t fold_w, int band) {
  constexpr int SLOT = 80, STAGE_BYTES = 2 * K2_TILE * SLOT;
  __shared__ __attribute__((aligned(16))) uint4 lds_ab[2 * STAGE_BYTES / 16];   // 40 KiB ring
  const RectTile rt = decode_rect_tile(blockIdx.x, n, row_begin, row_end, tile_stride, upper_only, TR, fold_q, fold_w, band);
  if (!rt.valid || !s12_takes(rt.I0, rt.J0, n, row_end, ld, rt.Jloc, out)) return;
  const int64_t I0 = rt.I0, J0 = rt.J0;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tx0 = ((wave & 1) << 3) + (lane & 7), ty0 = ((wave >> 1) << 3) + (lane >> 3);
  // wave-uniform DMA source: wave w stages LDS slots [64w, 64w + 64) = rows (w < 2, row copy) or columns (w >= 2, column copy) of the padded twin
  const uint32_t wave_u = __builtin_amdgcn_readfirstlane((uint32_t)wave);
  const uint32_t nstage = (uint32_t)((n_hash + K2_GROUP - 1) / K2_GROUP), stage_bytes = (uint32_t)(K2_TILE * SLOT);
  const uint64_t src = reinterpret_cast<uint64_t>(planes) +
                       4u * (uint64_t)(pad16_base_words(n, n_hash) + (wave_u >= 2 ? pad16_copy_words(n, n_hash) : 0)) +
                       (uint64_t)((wave_u >= 2 ? J0 : I0) / K2_TILE) * nstage * stage_bytes + (wave_u & 1u) * (64u * SLOT);
  const uint32_t sl = __builtin_amdgcn_readfirstlane((uint32_t)src), sh = __builtin_amdgcn_readfirstlane((uint32_t)(src >> 32));
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_t *)lds_ab);
  uint32_t mis[8][4];
  uint32_t tid_after;
  {
    register uint32_t r120 asm("v120") = lds_base + (uint32_t)(ty0 * SLOT);
    register uint32_t r121 asm("v121") = lds_base + (uint32_t)((K2_TILE + tx0) * SLOT);
    register uint32_t r124 asm("v124") = (uint32_t)tid * 4u;
#define K2_CNT(i) register uint32_t c##i asm("v" #i);
    K2_CNT(64) K2_CNT(65) K2_CNT(66) K2_CNT(67) K2_CNT(68) K2_CNT(69) K2_CNT(70) K2_CNT(71) K2_CNT(72) K2_CNT(73) K2_CNT(74)
    K2_CNT(75) K2_CNT(76) K2_CNT(77) K2_CNT(78) K2_CNT(79) K2_CNT(80) K2_CNT(81) K2_CNT(82) K2_CNT(83) K2_CNT(84) K2_CNT(85)
    K2_CNT(86) K2_CNT(87) K2_CNT(88) K2_CNT(89) K2_CNT(90) K2_CNT(91) K2_CNT(92) K2_CNT(93) K2_CNT(94) K2_CNT(95)
#undef K2_CNT
#define K2_S16_OPERANDS                                                                                                                      \
        : "=v"(c64), "=v"(c65), "=v"(c66), "=v"(c67), "=v"(c68), "=v"(c69), "=v"(c70), "=v"(c71), "=v"(c72), "=v"(c73), "=v"(c74),  \
          "=v"(c75), "=v"(c76), "=v"(c77), "=v"(c78), "=v"(c79), "=v"(c80), "=v"(c81), "=v"(c82), "=v"(c83), "=v"(c84), "=v"(c85),  \
          "=v"(c86), "=v"(c87), "=v"(c88), "=v"(c89), "=v"(c90), "=v"(c91), "=v"(c92), "=v"(c93), "=v"(c94), "=v"(c95)  \
        : [lb] "s"(lds_base), [ns] "s"(nstage), [st] "s"(stage_bytes), [wv] "s"(wave_u), [sl] "s"(sl), [sh] "s"(sh),  \
          "v"(r120), "v"(r121), "v"(r124)  \
        : "memory", "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "v125",  \
          "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",  \
          "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39",  \
          "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59",  \
          "v60", "v61", "v62", "v63", "v96", "v97", "v98", "v99",  \
          "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116",  \
          "v117", "v118", "v119", "v122", "v123"
    if (CODE_BITS == 14) {
      asm volatile(
#include K2_LOOP_INC_14
          K2_S16_OPERANDS);
    } else if (CODE_BITS == 15) {
      asm volatile(
#include K2_LOOP_INC_15
          K2_S16_OPERANDS);
    } else {
      asm volatile(
#include K2_LOOP_INC_16
          K2_S16_OPERANDS);
    }
#undef K2_S16_OPERANDS
    asm volatile("" : "+v"(r124));                               // lane ids are re-derived from the value that crossed the block
    tid_after = r124 >> 2;
    const uint32_t cnt[32] = {c64, c65, c66, c67, c68, c69, c70, c71, c72, c73, c74, c75, c76, c77, c78, c79,
                              c80, c81, c82, c83, c84, c85, c86, c87, c88, c89, c90, c91, c92, c93, c94, c95};
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) mis[r][c] = cnt[4 * r + c];
  }
  const uint32_t nn = (uint32_t)n_hash * 0x10001u;             // two match counts per register (no borrow: each <= n_hash)
  const int tid_e = (int)tid_after, wave_e = tid_e >> 6, lane_e = tid_e & 63;
  const int tx_e = ((wave_e & 1) << 3) + (lane_e & 7), ty_e = ((wave_e >> 1) << 3) + (lane_e >> 3);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    uint16_t *orow = out + (I0 + 32 * (r >> 1) + 2 * ty_e + (r & 1) + rt.Iloc) * ld + (rt.Jloc + J0 + 2 * tx_e);
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<uint32_t *>(orow + 32 * g) = nn - mis[r][g];
  }
}

// ---- the same 12-plane loop, PERSISTENT: a workgroup walks a sequence of tiles and its DMA ring never drains ----
// (Opt-in, see launch_mh_compare.)  k_mh_compare_a12 spends 29 % of a workgroup's life outside the plane loop (profiles/r02_b_k2_timeline_*.json: 8.6 us from
// launch to the loop, 14.5 us of stores, 1.4 us until the slot's next workgroup starts, against 59 us in the loop), so on
// average only 2.8 of the CU's 4 workgroups feed the VALU.  Here the grid is one workgroup per resident slot (4 per CU);
// workgroup (XCD x, slot s) takes the tile ids x*per_xcd + s + k*wg_per_xcd, k = 0, 1, ... -- the XCD's workgroups sweep a
// window of consecutive ids together, so the L2 reuse of the banded order is kept.  The generated block (K2ASM_PERSIST=1,
// k2_loop_p12p.inc) issues the NEXT tile's first two stages during the last two stages of the current one and waits for them
// before it ends; the C++ epilogue then issues the tile's stores and the next block starts computing at once -- the stores
// drain under its first two stages (vmcnt counts loads and stores together and in order: the block's first counted wait is at
// stage 2).  The count -> double table lives in its own 4 KiB of LDS (built once per workgroup; 36 + 4 KiB x 4 workgroups =
// the CU's 160 KiB), so the epilogue needs no barrier.  float64 output therefore needs n_hash <= 511; larger n_hash and
// single-stage inputs (n_hash <= 32) stay with k_mh_compare_a12.
#ifndef K2_LOOP_INC_P
#define K2_LOOP_INC_P "k2_loop_p12p.inc"
#endif
#ifdef K2_DYNAMIC   // experiment: tile ids handed out by a per-XCD atomic counter (a sliding window like the hardware dispatcher's)
__device__ unsigned int g_k2_next[8];
#endif
constexpr int K2_P12_TABLE = 512, K2_P8_TABLE = 2048;
// PL = 12, or 8: the dense half of the heavy / rare split (round 4; dict_kernels.hip k_hy_split): 32-byte slots, a 24 KiB ring, four two-plane
// steps per stage -- k2_loop_p8p.inc (band kernel: no wave priority) / k2_loop_p8.inc (ONE = one tile per workgroup, the grid of k_mh_compare_a12:
// the launcher passes wg_per_xcd = per_xcd; priority 2 inside the stage loop like k_mh_compare_a12's block).
template <bool F64, int PL = 12, bool ONE = false>
__global__ __launch_bounds__(K2_THREADS, 4) void k_mh_compare_p12(const uint32_t *__restrict__ planes, int64_t n, int n_hash,
                                                                  void *__restrict__ out_v, int64_t ld, int64_t ntiles,
                                                                  int64_t per_xcd, int wg_per_xcd, int64_t tile_begin) {
  // the tile ids [tile_begin, ntiles) are dealt in 8 runs of per_xcd (the pipelined duplicate route launches one band range at a time)
  static_assert(PL == 12 || PL == 8, "generated blocks exist for 12 and 8 planes");
  constexpr int SEGS = PL / 4, STAGE_UNITS = 2 * K2_TILE * SEGS;
  __shared__ __attribute__((aligned(16))) uint4 lds_ab[3 * STAGE_UNITS];   // 36 KiB ring (24 KiB at PL = 8)
  __shared__ double ratio_tab[F64 ? (PL == 8 ? K2_P8_TABLE : K2_P12_TABLE) : 1];   // (PL = 8: the ring is 12 KiB smaller, the table may be 12 KiB larger)
  const int T = (int)((n + K2_TILE - 1) / K2_TILE);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  // tile ids fit 31 bits (the launcher checks): 32-bit scalars, so that loop control stays on the scalar unit -- nothing
  // wave-uniform may end up in a VGPR that has to live across the block (it clobbers the VGPR file; hipcc would spill to
  // scratch and reload behind a vmcnt wait in the middle of the tile's stores)
  const int lim = (int)((tile_begin + (int64_t)(xcd + 1) * per_xcd < ntiles) ? tile_begin + (int64_t)(xcd + 1) * per_xcd : ntiles);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (F64) {
    for (int c = tid; c <= n_hash; c += K2_THREADS) ratio_tab[c] = (double)c / (double)n_hash;   // src/minHash.cpp:174
    __syncthreads();
  }
  const PlaneGeom pg = plane_geom(n, n_hash, PL);
  // wave-uniform DMA source: wave w stages LDS slots [64w, 64w + 64) of a stage = rows (w < 2) or columns (w >= 2) of the
  // tile, 3 KiB per stage; the operand is stored in staging order, so block b of an operand starts b * nst * 6144 bytes in
  const uint32_t wave_u = __builtin_amdgcn_readfirstlane((uint32_t)wave);
  const uint64_t block_bytes = (uint64_t)pg.nst * (128u * PL * 4u);
  const uint64_t wave_base = reinterpret_cast<uint64_t>(planes) + (wave_u >= 2 ? (uint64_t)pg.copy_words * 4u : 0u) + (wave_u & 1u) * (uint32_t)(SEGS * 1024);
  auto source_of = [&](const TileId &t) -> uint64_t { return wave_base + (uint64_t)(wave_u >= 2 ? t.tj : t.ti) * block_bytes; };
  auto next_taken = [&](int from, TileId &t) -> int {               // first tile id >= from (stride wg_per_xcd) the asm kernels take
    for (int L = from; L < lim; L += wg_per_xcd) {
      int To = T;
      asm volatile("" : "+s"(To));                              // opaque per call: no float invariants of the decode hoisted into VGPRs
      t = decode_tile(L, To, To, true);
      t.ti = __builtin_amdgcn_readfirstlane(t.ti);             // wave-uniform by construction: keep them in SGPRs, nothing
      t.tj = __builtin_amdgcn_readfirstlane(t.tj);             // per-lane may live across the block (it clobbers the VGPR file)
      if (t.valid && a12_takes(t.ti, t.tj, n, ld, out_v, F64)) return L;
    }
    return lim;
  };
  const int tx0 = ((wave & 1) << 3) + (lane & 7), ty0 = ((wave >> 1) << 3) + (lane >> 3);
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_t *)lds_ab);
  const uint32_t nstage = (uint32_t)((n_hash + K2_GROUP - 1) / K2_GROUP), stage_bytes = 128u * PL * 4u;
  const uint32_t nn = (uint32_t)n_hash * 0x10001u;             // two match counts per register (no borrow: each <= n_hash)
  register uint32_t r120 asm("v120") = lds_base + (uint32_t)(ty0 * SEGS * 16);                       // the lane's first row operand inside a stage
  register uint32_t r121 asm("v121") = lds_base + (uint32_t)((K2_TILE * SEGS + tx0 * SEGS) * 16);   // ... first column operand
  register uint32_t r124 asm("v124") = (uint32_t)tid * 4u;

  TileId cur, nxt;
#ifdef K2_DYNAMIC
  __shared__ unsigned int s_next[2];
  int fetch_no = 0;
  auto next_dyn = [&](TileId &t) -> int {
    for (;;) {
      if (tid == 0) s_next[fetch_no & 1] = atomicAdd(&g_k2_next[xcd], 1u);
      __syncthreads();
      const int Lq = (int)(tile_begin + (int64_t)xcd * per_xcd) + (int)__builtin_amdgcn_readfirstlane(s_next[fetch_no & 1]);
      ++fetch_no;
      if (Lq >= lim) return lim;
      int To = T;
      asm volatile("" : "+s"(To));
      t = decode_tile(Lq, To, To, true);
      t.ti = __builtin_amdgcn_readfirstlane(t.ti);
      t.tj = __builtin_amdgcn_readfirstlane(t.tj);
      if (t.valid && a12_takes(t.ti, t.tj, n, ld, out_v, F64)) return Lq;
    }
  };
  int L = next_dyn(cur);
#else
  int L = next_taken((int)(tile_begin + (int64_t)xcd * per_xcd) + slot, cur);
#endif
  uint32_t flags = 1u, phase = 0u;                             // bit 0: first tile of this workgroup; ring slot of the tile's stage 0
#ifdef DA_K2_TIMING
  int it_stamp = 0;
#define K2P_STAMP(j) do { if (g_k2_timing && threadIdx.x == 0 && it_stamp < 64) \
    g_k2_timing[((size_t)blockIdx.x * 64 + it_stamp) * 4 + (j)] = wall_clock64(); } while (0)
#else
#define K2P_STAMP(j) do { } while (0)
#endif
  while (L < lim) {
    K2P_STAMP(0);
#ifdef K2_DYNAMIC
    const int Ln = next_dyn(nxt);
#else
    const int Ln = next_taken(L + wg_per_xcd, nxt);
#endif
    if (Ln < lim) flags |= 2u;
    const uint64_t src = source_of(cur), src_n = source_of(nxt);
    const uint32_t sl = __builtin_amdgcn_readfirstlane((uint32_t)src), sh = __builtin_amdgcn_readfirstlane((uint32_t)(src >> 32));
    const uint32_t nl = __builtin_amdgcn_readfirstlane((uint32_t)src_n), nh = __builtin_amdgcn_readfirstlane((uint32_t)(src_n >> 32));
    const uint32_t fl = __builtin_amdgcn_readfirstlane(flags), sp = __builtin_amdgcn_readfirstlane(phase * (uint32_t)(STAGE_UNITS * 16));
    uint32_t mis[8][4];
    uint32_t tid_after;
    K2P_STAMP(1);
    {
#define K2_CNT(i) register uint32_t c##i asm("v" #i);
      K2_CNT(64) K2_CNT(65) K2_CNT(66) K2_CNT(67) K2_CNT(68) K2_CNT(69) K2_CNT(70) K2_CNT(71) K2_CNT(72) K2_CNT(73) K2_CNT(74)
      K2_CNT(75) K2_CNT(76) K2_CNT(77) K2_CNT(78) K2_CNT(79) K2_CNT(80) K2_CNT(81) K2_CNT(82) K2_CNT(83) K2_CNT(84) K2_CNT(85)
      K2_CNT(86) K2_CNT(87) K2_CNT(88) K2_CNT(89) K2_CNT(90) K2_CNT(91) K2_CNT(92) K2_CNT(93) K2_CNT(94) K2_CNT(95)
#undef K2_CNT
#define K2P_OPERANDS \
          : "=v"(c64), "=v"(c65), "=v"(c66), "=v"(c67), "=v"(c68), "=v"(c69), "=v"(c70), "=v"(c71), "=v"(c72), "=v"(c73), "=v"(c74), \
            "=v"(c75), "=v"(c76), "=v"(c77), "=v"(c78), "=v"(c79), "=v"(c80), "=v"(c81), "=v"(c82), "=v"(c83), "=v"(c84), "=v"(c85), \
            "=v"(c86), "=v"(c87), "=v"(c88), "=v"(c89), "=v"(c90), "=v"(c91), "=v"(c92), "=v"(c93), "=v"(c94), "=v"(c95) \
          : [lb] "s"(lds_base), [ns] "s"(nstage), [st] "s"(stage_bytes), [wv] "s"(wave_u), [sl] "s"(sl), [sh] "s"(sh), [nl] "s"(nl), \
            [nh] "s"(nh), [fl] "s"(fl), [sp] "s"(sp), "v"(r120), "v"(r121), "v"(r124) \
          : "memory", "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "v125",   /* m0 is saved in s47 and restored by the block */ \
            "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", \
            "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", \
            "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", \
            "v60", "v61", "v62", "v63", "v96", "v97", "v98", "v99", \
            "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", \
            "v117", "v118", "v119", "v122", "v123"
      if constexpr (PL == 12) {
        asm volatile(
#include K2_LOOP_INC_P
            K2P_OPERANDS);
      } else if constexpr (ONE) {
        asm volatile(
#include "k2_loop_p8.inc"
            K2P_OPERANDS);
      } else {
        asm volatile(
#include "k2_loop_p8p.inc"
            K2P_OPERANDS);
      }
#undef K2P_OPERANDS
      asm volatile("" : "+v"(r124));                             // lane ids are re-derived from the value that crossed the block
      tid_after = r124 >> 2;
      const uint32_t cnt[32] = {c64, c65, c66, c67, c68, c69, c70, c71, c72, c73, c74, c75, c76, c77, c78, c79,
                                c80, c81, c82, c83, c84, c85, c86, c87, c88, c89, c90, c91, c92, c93, c94, c95};
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) mis[r][c] = cnt[4 * r + c];
    }
    K2P_STAMP(2);
    const int wave_e = (int)tid_after >> 6, lane_e = (int)tid_after & 63;
    const int tx = ((wave_e & 1) << 3) + (lane_e & 7), ty = ((wave_e >> 1) << 3) + (lane_e >> 3);
    const int64_t I0 = (int64_t)cur.ti * K2_TILE, J0 = (int64_t)cur.tj * K2_TILE;
#ifdef K2_NO_STORES
    if (mis[0][0] == 0xdeadbeefu) reinterpret_cast<uint32_t *>(out_v)[0] = tid_after;
    if (n_hash > 0) { cur = nxt; L = Ln; flags = 0u; phase = (phase + nstage) % 3u; continue; }
#endif
    if (F64) {
      const char *tb = reinterpret_cast<const char *>(ratio_tab);
      double *out = reinterpret_cast<double *>(out_v);
#pragma unroll
      for (int g = 0; g < 4; ++g) {                              // two of the lane's rows at a time keeps the epilogue in 128 VGPRs
        double v0[8], v1[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint32_t m0 = nn - mis[2 * g][c], m1 = nn - mis[2 * g + 1][c];
          v0[2 * c] = *reinterpret_cast<const double *>(tb + ((m0 << 3) & 0x7fff8u));
          v0[2 * c + 1] = *reinterpret_cast<const double *>(tb + ((m0 >> 13) & 0x7fff8u));
          v1[2 * c] = *reinterpret_cast<const double *>(tb + ((m1 << 3) & 0x7fff8u));
          v1[2 * c + 1] = *reinterpret_cast<const double *>(tb + ((m1 >> 13) & 0x7fff8u));
        }
        double *orow = out + (I0 + 32 * g + 2 * ty) * ld + (J0 + 2 * tx);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          nt_store2(orow + 32 * q, v0[2 * q], v0[2 * q + 1]);
          nt_store2(orow + ld + 32 * q, v1[2 * q], v1[2 * q + 1]);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c)                              // mirrored store (src/minHash.cpp:176)
          nt_store2(out + (J0 + 32 * (c >> 1) + 2 * tx + (c & 1)) * ld + (I0 + 32 * g + 2 * ty), v0[c], v1[c]);
      }
    } else {
      uint16_t *out = reinterpret_cast<uint16_t *>(out_v);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        uint16_t *orow = out + (I0 + 32 * (r >> 1) + 2 * ty + (r & 1)) * ld + (J0 + 2 * tx);
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<uint32_t *>(orow + 32 * g) = nn - mis[r][g];
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        uint16_t *orow = out + (J0 + 32 * (c >> 1) + 2 * tx + (c & 1)) * ld + (I0 + 2 * ty);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint32_t lo = mis[2 * g][c >> 1], hi = mis[2 * g + 1][c >> 1];
          const uint32_t pk = (c & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
          *reinterpret_cast<uint32_t *>(orow + 32 * g) = nn - pk;
        }
      }
    }
    K2P_STAMP(3);
#ifdef DA_K2_TIMING
    ++it_stamp;
#endif
    cur = nxt;
    L = Ln;
    flags = 0u;
    phase = (phase + nstage) % 3u;
  }
#undef K2P_STAMP
}

#ifdef DA_K2_EXPERIMENTS   // round 3's two float64-store experiments (k_mh_compare_r12 / _q12): built only by tools/experiments/build.sh, never into the product library
#include "../../tools/experiments/k2_store_kernels.inc"
#endif

// PL = bit planes per group of 32 hash functions: 32 (raw uint32 values) or 16 / 12 / 8 (dictionary
// codes of dict_kernels.hip, as many planes as the largest column dictionary needs: same equalities
// off the diagonal, a fraction of the planes; the diagonal is forced).
template <bool SYM, bool F64, int PL>
__global__ __launch_bounds__(K2_THREADS, 3) void k_mh_compare(
    const uint32_t *__restrict__ planes, int64_t n, int n_hash, int64_t row_begin,
    int64_t row_end, int tile_stride, int upper_only, int TR, void *__restrict__ out_v, int64_t ld,
    int64_t ntiles, int64_t per_xcd, int fold_q, int64_t fold_w, int band, int only_edge) {
  // Row-block geometry: local tile row q covers global rows row_begin + q*tile_stride*128 + [0,128)
  // (tile_stride = 1: a contiguous block; = world: the cyclic shard of one rank) and is stored at
  // local rows q*128 + [0,128) of `out`.  upper_only skips tiles left of the diagonal.
  // ring of 3 stages; a stage = SP planes of {128 a-rows, 128 b-rows} = 256 x 4*SP B (16 KiB at SP = 16)
  constexpr int SP = (PL == 32) ? 16 : PL;       // planes per stage
  constexpr int SEGS = SP / 4;                   // 16-byte segments per row per stage
  constexpr int SPG = PL / SP;                   // stages per group: 2 (PL = 32) or 1
  constexpr int STAGE_UNITS = 2 * K2_TILE * SEGS;
  constexpr int K2_NSTAGE = K2_RING_DEPTH(PL);   // LDS ring depth
  __shared__ __attribute__((aligned(16))) uint4 lds_ab[K2_NSTAGE * STAGE_UNITS];

  // ---- which tile.  Blocks b and b+8 share an XCD (speed only).
  //   symmetric: XCD x walks the contiguous id range [x*per_xcd, (x+1)*per_xcd) of the banded
  //              triangle enumeration (equal work per XCD by construction);
  //   row block: bands of 8 tile rows are dealt round-robin to the XCDs -- with upper_only the
  //              valid part of a band shrinks with its row index, and contiguous ranges would
  //              leave the last XCDs idle (measured: 1.7x slower).
  K2_STAMP(0);
  K2_STAMP_HW();
  const int64_t bid = blockIdx.x;
  const int T = (int)((n + K2_TILE - 1) / K2_TILE);
  TileId tid2;
  if (SYM && only_edge) {
    // the tiles k_mh_compare_a12 leaves: the T diagonal tiles, then the last tile column (when n is not a multiple of 128)
    tid2.valid = bid < 2 * (int64_t)T - 1;
    tid2.ti = bid < T ? (int)bid : (int)(bid - T);
    tid2.tj = bid < T ? (int)bid : T - 1;
  } else if (SYM) {
    const int64_t L = (bid & 7) * per_xcd + (bid >> 3);
    if (L >= ntiles) return;
    tid2 = decode_tile(L, TR, T, true);
  } else {
    const uint32_t per_band = (uint32_t)band * (uint32_t)T;               // band = tile rows per band (launcher's choice)
    const uint32_t k = (uint32_t)(bid >> 3);
    const uint32_t kb = k / per_band;
    const int r0 = ((int)(bid & 7) + 8 * (int)kb) * band;                 // first local tile row of the band
    const int l = (int)(k - kb * per_band);
    const int h = (TR - r0 < band) ? (TR - r0) : band;
    if (h <= 0) return;
    tid2.tj = div_small(l, h);
    tid2.ti = r0 + (l - tid2.tj * h);
    tid2.valid = tid2.tj < T;
  }
  if (!tid2.valid) return;
  const int64_t I0 = row_begin + (int64_t)tid2.ti * tile_stride * K2_TILE;  // global row of tile row 0
  const int64_t J0 = (int64_t)tid2.tj * K2_TILE;
  int64_t Iloc = (int64_t)tid2.ti * K2_TILE - I0;                           // local row = global row + Iloc
  int64_t Jloc = 0;                                                         // local col = global col + Jloc
  if (!SYM && fold_q > 0) {  // folded shard layout (ShardGeom): tile rows q and Q-1-q share a stored row
    const int q = tid2.ti;
    const bool front = q <= fold_q - 1 - q;
    Iloc = (int64_t)(front ? q : fold_q - 1 - q) * K2_TILE - I0;
    Jloc = front ? -I0 : shard_back(fold_w, n);
  }
  if (I0 >= row_end || I0 >= n) return;
  if (!SYM && upper_only && J0 + K2_TILE <= I0) return;                     // tile entirely left of the diagonal
  // the interior tiles were taken by k_mh_compare_a12 (hand-scheduled 12-plane loop): only border / diagonal tiles here
  if (SYM && only_edge && a12_takes(tid2.ti, tid2.tj, n, ld, out_v, F64)) return;
  if (!SYM && !F64 && only_edge && s12_takes(I0, J0, n, row_end, ld, Jloc, out_v)) return;   // ... or by k_mh_compare_s12 (shard / row-block modes)

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int tx = ((wave & 1) << 3) + (lane & 7);   // column coordinate 0..15
  const int ty = ((wave >> 1) << 3) + (lane >> 3); // row coordinate 0..15

  // ---- staging by LDS-DMA (global_load_lds_dwordx4): one wave instruction
  // lands 64 x 16 B contiguously in LDS -- no VGPRs, no ds_write.  LDS is
  // written linearly (base + lane*16): 16-byte unit u of a stage is physical
  // position u % SEGS of slot u / SEGS.  The operand is stored in that very
  // order per 128-row block (plane_unit_word), so for a tile aligned to 128 rows
  // the 64 lanes of an instruction read 1 KiB of consecutive memory; an unaligned
  // row block (arbitrary row_begin) just gathers.  A wave issues SEGS instructions per stage.
  const PlaneGeom pg = plane_geom(n, n_hash, PL);
  const uint32_t *src[SEGS];
#pragma unroll
  for (int q = 0; q < SEGS; ++q) {
    const int u = (wave * SEGS + q) * 64 + lane;
    const int s = u / SEGS;                           // slot 0..255 (a: 0..127, b: 128..255)
    const int sl = s & 127;
    const int seg = (u - s * SEGS) ^ (SEGS == 4 ? ((sl >> 2) & 3) : 0);   // logical segment wanted at this position
    int64_t g = ((s < 128) ? I0 : J0) + k2_row_of_slot(sl);
    if (g > n - 1) g = n - 1;                          // rows past the end: any valid row, never stored
    // row operand (a) from the plain copy, column operand (b) from the pair-swapped copy
    src[q] = planes + ((s < 128) ? 0 : pg.copy_words) + plane_unit_word(pg, g, 0, seg);
  }
  // The DMA is issued from inline asm on purpose: with the builtin, hipcc must assume the LDS write may
  // alias the ring slots being read and puts s_waitcnt vmcnt(0) in front of the first ds_read of every
  // segment, which drains the "asynchronous" copy at every stage (seen in the ISA).  Hidden in asm the
  // copies stay in flight; the counted waits below (wait_stage) are then OUR responsibility:
  // each wave issues exactly SEGS VMEM operations per stage and nothing else touches vmcnt in the loop.
  const uint32_t lds_wave_base =
      __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_t *)(lds_ab + wave * SEGS * 64));
  auto issue = [&](int stage) {  // stage s of a block: 128 * SP words further
    const uint32_t base = lds_wave_base + (uint32_t)(stage % K2_NSTAGE) * (uint32_t)(STAGE_UNITS * sizeof(uint4));
#pragma unroll
    for (int q = 0; q < SEGS; ++q) {
      const uint32_t *g = src[q] + stage * (128 * SP);
      // m0 (the LDS-DMA destination) is compiler-reserved: saved and restored inside the statement instead of clobbered
      uint32_t keep_m0;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep_m0) : "v"(g), "s"(base + (uint32_t)(q * 64 * sizeof(uint4))) : "memory");
    }
  };
  // before the barrier that opens stage s: its SEGS copies must have landed; those of the `younger`
  // stages issued after it may still fly
  auto wait_stage = [&](int younger) {
    const int outstanding = younger * SEGS;
    if (outstanding >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (outstanding == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (outstanding == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (outstanding == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (outstanding == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  uint32_t mis[8][4];  // mismatch counters, two 16-bit counters per register (columns 2j, 2j+1)
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) mis[r][c] = 0;

  // lane-constant parts of the LDS read addresses (16-byte units); the r / c
  // strides (16 slots = 64 units) fold into the ds_read immediate offset.
  const int base_a = ty * SEGS, base_b = K2_TILE * SEGS + tx * SEGS;
  const int xa = SEGS == 4 ? (ty >> 2) & 3 : 0, xb = SEGS == 4 ? (tx >> 2) & 3 : 0;  // (slot >> 2) & 3 of the lane's slots

  uint32_t d[8][8];  // d |= a_p ^ b_p over the planes of a group
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) d[r][c] = 0;
  // Operands of the 4-plane segment about to be computed.  A stage is walked segment by segment;
  // the row operands of segment s+1 are fetched into a[r] right after a[r]'s last use in segment s
  // (column 7), and the first operands of a stage right after its barrier, BEFORE the popcounts of
  // the previous group -- so a wave meets an LDS round trip with VALU work in hand instead of idle.
  uint4 a[8], b;
  auto preload = [&](int stage) __attribute__((always_inline)) {
    const uint4 *S = lds_ab + (stage % K2_NSTAGE) * STAGE_UNITS;
    b = S[base_b + xb];
#pragma unroll
    for (int r = 0; r < 8; ++r) a[r] = S[base_a + xa + r * 16 * SEGS];
  };
  // init: first segment of a group -- d = a ^ b (v_xor, full rate) instead of d |= a ^ b, which saves
  // clearing the 64 accumulators after every popcount
  auto segment = [&](const uint4 *S, int seg, auto more_tag, auto init_tag) __attribute__((always_inline)) {
    constexpr bool more = decltype(more_tag)::value;               // another segment of this stage follows
    constexpr bool init = decltype(init_tag)::value;
    const uint4 *Sb = S + base_b + (seg ^ xb);
    const uint4 *nSa = S + base_a + ((seg + 1) ^ xa);
    const uint4 *nSb = S + base_b + ((seg + 1) ^ xb);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      uint4 bn = b;                                                // next column's operand in flight
      if (c + 1 < 8) bn = Sb[(c + 1) * 16 * SEGS];
      else if (more) { bn = nSb[0]; __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        // b holds planes (1,0,3,2): plane p sits in registers of opposite index parity in
        // a and b, so no v_bitop3 has all three sources in one VGPR bank (half rate otherwise)
        uint32_t v = init ? (a[r].x ^ b.y) : or_xor(d[r][c], a[r].x, b.y);
        v = or_xor(v, a[r].y, b.x);
        v = or_xor(v, a[r].z, b.w);
        d[r][c] = or_xor(v, a[r].w, b.z);
        if (c == 7 && more) {
          // last column: rows strictly one after the other (a wave issues one VALU op per 4 cycles, so the
          // dependent chain costs nothing) and each row operand is re-fetched the moment it is dead --
          // left alone the scheduler interleaves the 8 chains and all 8 reads land in the last 8 ops
          a[r] = nSa[r * 16 * SEGS];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      b = bn;
    }
  };
  auto compute = [&](int stage) __attribute__((always_inline)) {
    const uint4 *S = lds_ab + (stage % K2_NSTAGE) * STAGE_UNITS;
    if (SPG == 1) {   // the stage is a whole group: first segment initialises d
      if (SEGS > 1) segment(S, 0, std::true_type{}, std::true_type{});
      else segment(S, 0, std::false_type{}, std::true_type{});
#pragma unroll 1
      for (int seg = 1; seg < SEGS - 1; ++seg) segment(S, seg, std::true_type{}, std::false_type{});
      if (SEGS > 1) segment(S, SEGS - 1, std::false_type{}, std::false_type{});
    } else {
#pragma unroll 1
      for (int seg = 0; seg < SEGS - 1; ++seg) segment(S, seg, std::true_type{}, std::false_type{});
      segment(S, SEGS - 1, std::false_type{}, std::false_type{});
    }
  };

  const int ngroup = (n_hash + K2_GROUP - 1) / K2_GROUP;
  const int nstage = SPG * ngroup;
  auto count_group = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        mis[r][c] += (uint32_t)__builtin_popcount(d[r][2 * c]) + ((uint32_t)__builtin_popcount(d[r][2 * c + 1]) << 16);
    if (SPG != 1) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c) d[r][c] = 0;
    }
  };
  // wait_stage: this wave's copies for the stage have landed; the barrier: so have everyone's, and
  // nobody still reads the ring slot the next issue() overwrites (it was consumed two stages ago).
  K2_STAMP(1);
  // waves inside their plane loop issue ahead of waves that are decoding or storing a tile (measured on the
  // hand-scheduled kernel: -4 %)
  __builtin_amdgcn_s_setprio(2);
  constexpr int AHEAD = K2_NSTAGE - 1;           // stages in flight ahead of the one being computed
#pragma unroll
  for (int st = 0; st < AHEAD; ++st)
    if (st < nstage) issue(st);
  for (int st = 0; st < nstage; ++st) {
    const int left = nstage - 1 - st;            // stages after this one
    wait_stage(left < AHEAD - 1 ? left : AHEAD - 1);
    __syncthreads();
    if (st + AHEAD < nstage) issue(st + AHEAD);
    preload(st);
    if (st > 0 && (SPG == 1 || (st & 1) == 0)) count_group();   // the group that ended with stage st-1
    compute(st);
  }
  count_group();
  __builtin_amdgcn_s_setprio(0);
  if (PL != 32) {
    // dictionary codes make a sequence differ from itself wherever it holds a singleton value:
    // the diagonal is n_hash matches by definition (src/minHash.cpp:161)
    if (I0 - J0 < K2_TILE && J0 - I0 < K2_TILE) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int64_t gi = I0 + 32 * (r >> 1) + 2 * ty + (r & 1);
          const int64_t gj = J0 + 32 * (c >> 1) + 2 * tx + (c & 1);
          if (gi == gj) mis[r][c >> 1] &= (c & 1) ? 0x0000ffffu : 0xffff0000u;
        }
    }
  }
  K2_STAMP(2);
  auto matches = [&](int r, int c) -> uint32_t {  // reference src/minHash.cpp:168-173
    return (uint32_t)n_hash - ((mis[r][c >> 1] >> ((c & 1) * 16)) & 0xffffu);
  };

  // ---- epilogue.  count -> double through a table built with the same
  // IEEE divide the reference does ((double)matches / n_hash,
  // src/minHash.cpp:174); the staging LDS is free now and holds the table.
  double *ratio = reinterpret_cast<double *>(lds_ab);
  const bool use_table = F64 && (n_hash + 1) <= (int)(sizeof(lds_ab) / (sizeof(double)));
  if (F64) {
    __syncthreads();
    if (use_table)
      for (int c = tid; c <= n_hash; c += K2_THREADS) ratio[c] = (double)c / (double)n_hash;
    __syncthreads();
  }
  K2_STAMP(6);
  auto widen = [&](uint32_t c) -> double {
    return use_table ? ratio[c] : (double)c / (double)n_hash;
  };

  // lane's rows: 32*g + 2*ty + e  <-> index r = 2*g + e ; cols likewise.
  // Tiles that lie wholly inside the matrix (all but the last tile row / column) take a straight-line
  // epilogue: no per-element bounds tests, one table read per element, 16-byte stores at immediate
  // offsets.  The workgroup shares its SIMDs with two others that are in their plane loops, so every
  // instruction here costs ~3 issue slots: the general path below ran 10 us per tile, this one ~3.
  const bool interior = I0 + K2_TILE <= (row_end < n ? row_end : n) && J0 + K2_TILE <= n;
  if (F64) {
    double *out = reinterpret_cast<double *>(out_v);
    const bool vec_ok = ((ld & 1) == 0) && ((Jloc & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    if (use_table && interior && vec_ok) {
      const char *tb = reinterpret_cast<const char *>(ratio);
      double v[8][8];
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint32_t mm = (uint32_t)n_hash * 0x10001u - mis[r][c];          // two match counts (no borrow: each <= n_hash)
          v[r][2 * c] = *reinterpret_cast<const double *>(tb + ((mm << 3) & 0x7fff8u));
          v[r][2 * c + 1] = *reinterpret_cast<const double *>(tb + ((mm >> 13) & 0x7fff8u));
        }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        double *orow = out + (I0 + 32 * (r >> 1) + 2 * ty + (r & 1) + Iloc) * ld + (Jloc + J0 + 2 * tx);
#pragma unroll
        for (int g = 0; g < 4; ++g) nt_store2(orow + 32 * g, v[r][2 * g], v[r][2 * g + 1]);
      }
      K2_STAMP(7);
      if (SYM && tid2.ti != tid2.tj) {  // mirrored store (src/minHash.cpp:176)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          double *orow = out + (J0 + 32 * (c >> 1) + 2 * tx + (c & 1)) * ld + (I0 + 2 * ty);
#pragma unroll
          for (int g = 0; g < 4; ++g) nt_store2(orow + 32 * g, v[2 * g][c], v[2 * g + 1][c]);
        }
      }
      K2_STAMP(3);
      return;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int64_t gi = I0 + 32 * (r >> 1) + 2 * ty + (r & 1);
      if (gi >= row_end || gi >= n) continue;
      double *orow = out + (gi + Iloc) * ld + Jloc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t gj = J0 + 32 * g + 2 * tx;
        const double v0 = widen(matches(r, 2 * g)), v1 = widen(matches(r, 2 * g + 1));
        if (vec_ok && gj + 1 < n) {
          *reinterpret_cast<double2 *>(orow + gj) = make_double2(v0, v1);
        } else {
          if (gj < n) orow[gj] = v0;
          if (gj + 1 < n) orow[gj + 1] = v1;
        }
      }
    }
    K2_STAMP(7);
    if (SYM && tid2.ti != tid2.tj) {  // mirrored store (src/minHash.cpp:176)
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int64_t gj = J0 + 32 * (c >> 1) + 2 * tx + (c & 1);
        if (gj >= n) continue;
        double *orow = out + gj * ld;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int64_t gi = I0 + 32 * g + 2 * ty;
          const double v0 = widen(matches(2 * g, c)), v1 = widen(matches(2 * g + 1, c));
          if (vec_ok && gi + 1 < n) {
            *reinterpret_cast<double2 *>(orow + gi) = make_double2(v0, v1);
          } else {
            if (gi < n) orow[gi] = v0;
            if (gi + 1 < n) orow[gi + 1] = v1;
          }
        }
      }
    }
  } else {
    uint16_t *out = reinterpret_cast<uint16_t *>(out_v);
    const bool vec_ok = ((ld & 1) == 0) && ((Jloc & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & 3) == 0);
    if (interior && vec_ok) {   // two adjacent counts per 4-byte store
      const uint32_t nn = (uint32_t)n_hash * 0x10001u;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        uint16_t *orow = out + (I0 + 32 * (r >> 1) + 2 * ty + (r & 1) + Iloc) * ld + (Jloc + J0 + 2 * tx);
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<uint32_t *>(orow + 32 * g) = nn - mis[r][g];
      }
      if (SYM && tid2.ti != tid2.tj) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          uint16_t *orow = out + (J0 + 32 * (c >> 1) + 2 * tx + (c & 1)) * ld + (I0 + 2 * ty);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const uint32_t lo = mis[2 * g][c >> 1], hi = mis[2 * g + 1][c >> 1];
            const uint32_t pk = (c & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
            *reinterpret_cast<uint32_t *>(orow + 32 * g) = nn - pk;
          }
        }
      }
      K2_STAMP(3);
      return;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int64_t gi = I0 + 32 * (r >> 1) + 2 * ty + (r & 1);
      if (gi >= row_end || gi >= n) continue;
      uint16_t *orow = out + (gi + Iloc) * ld + Jloc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t gj = J0 + 32 * g + 2 * tx;
        if (gj < n) orow[gj] = (uint16_t)matches(r, 2 * g);
        if (gj + 1 < n) orow[gj + 1] = (uint16_t)matches(r, 2 * g + 1);
      }
    }
    if (SYM && tid2.ti != tid2.tj) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int64_t gj = J0 + 32 * (c >> 1) + 2 * tx + (c & 1);
        if (gj >= n) continue;
        uint16_t *orow = out + gj * ld;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int64_t gi = I0 + 32 * g + 2 * ty;
          if (gi < n) orow[gi] = (uint16_t)matches(2 * g, c);
          if (gi + 1 < n) orow[gi + 1] = (uint16_t)matches(2 * g + 1, c);
        }
      }
    }
  }
  K2_STAMP(3);
}

// Lower triangle <- upper triangle (after a gather of upper-triangular rows).
template <typename T>
__global__ __launch_bounds__(256) void k_symmetrize(T *__restrict__ m, int64_t n, int64_t ld) {
  __shared__ T tile[32][33];
  const int64_t bi = blockIdx.y, bj = blockIdx.x;  // tile (bi,bj) of the UPPER part is read
  if (bj < bi) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int64_t i = bi * 32 + r, j = bj * 32 + tx;
    if (i < n && j < n) tile[r][tx] = m[i * ld + j];
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t j = bj * 32 + r, i = bi * 32 + tx;  // write m[j][i] = upper(i,j)
    if (i < n && j < n && j > i) m[j * ld + i] = tile[tx][r];
  }
}

__global__ __launch_bounds__(256) void k_widen(const uint16_t *__restrict__ in, double *__restrict__ out,
                                               int64_t count, int is_nw, int n_hash) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    const uint32_t v = in[i];
    out[i] = is_nw ? (double)(v >> 8) / (double)(v & 255u) : (double)v / (double)n_hash;
  }
}

// n_hash > 65535 (host-pointer path only): the compare kernels count in 16 bits, so the hash functions are processed in
// chunks and the chunk counts summed in 32 bits; the divide happens once, on the total (src/minHash.cpp:174)
__global__ __launch_bounds__(256) void k_acc_counts(uint32_t *__restrict__ acc, const uint16_t *__restrict__ cnt, int64_t count, int first) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride)
    acc[i] = (first ? 0u : acc[i]) + cnt[i];
}
__global__ __launch_bounds__(256) void k_counts32_to_f64(const uint32_t *__restrict__ acc, double *__restrict__ out, int64_t count, int n_hash) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride)
    out[i] = (double)acc[i] / (double)n_hash;
}

}  // namespace

int launch_acc_counts(uint32_t *d_acc, const uint16_t *d_cnt, int64_t count, bool first, hipStream_t stream) {
  if (count <= 0) return DA_OK;
  hipLaunchKernelGGL(k_acc_counts, dim3(256 * 16), dim3(256), 0, stream, d_acc, d_cnt, count, first ? 1 : 0);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}
int launch_counts32_to_f64(const uint32_t *d_acc, double *d_out, int64_t count, int n_hash, hipStream_t stream) {
  if (count <= 0) return DA_OK;
  hipLaunchKernelGGL(k_counts32_to_f64, dim3(256 * 16), dim3(256), 0, stream, d_acc, d_out, count, n_hash);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

#ifdef DA_K2_TIMING
extern "C" int da_debug_set_k2_timing(unsigned long long *d_buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_k2_timing), &d_buf, sizeof(d_buf));
}
#endif

#ifdef DA_K2_DEBUG
extern "C" int da_debug_set_k2_dump(uint32_t *d_buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_k2_debug), &d_buf, sizeof(d_buf));
}
#endif

int launch_minhash_signatures(const uint8_t *d_res, const int64_t *d_off, int64_t n, int k,
                              int n_hash, const uint32_t *d_seeds, uint32_t *d_sig,
                              int64_t ld_sig, hipStream_t stream) {
  if (n <= 0) return DA_OK;
  if (n > 0x7fffffffLL) return fail(DA_ERR_UNSUPPORTED, "more than 2^31-1 sequences");
  const size_t lds = 2 * (size_t)(K1_CHUNK + k) * sizeof(uint32_t);
  if (lds > 64 * 1024) return fail(DA_ERR_UNSUPPORTED, "k = %d is larger than the signature kernel supports (k <= 7168)", k);
  dim3 grid((unsigned)n), block(K1_THREADS);
  if (k == 4)
    hipLaunchKernelGGL(k_minhash_signatures<true>, grid, block, lds, stream, d_res, d_off, k, n_hash, d_seeds, d_sig, ld_sig);
  else
    hipLaunchKernelGGL(k_minhash_signatures<false>, grid, block, lds, stream, d_res, d_off, k, n_hash, d_seeds, d_sig, ld_sig);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

#ifdef DA_K2_EXPERIMENTS
#include "../../tools/experiments/k2_store_host.inc"
#endif

int launch_mh_compare(const uint32_t *d_planes, int64_t n, int n_hash,
                      int64_t row_begin, int64_t row_end, bool symmetric, int kind,
                      void *d_out, int64_t ld, hipStream_t stream, int plane_bits, int tile_stride, bool upper_only,
                      int fold_q, int64_t fold_w) {
  if (row_end <= row_begin) return DA_OK;
  if (plane_bits != 8 && plane_bits != 12 && plane_bits != 14 && plane_bits != 15 && plane_bits != 16 && plane_bits != 32)
    return fail(DA_ERR_BAD_ARG, "plane_bits must be 8, 12, 14, 15, 16 or 32 (got %d)", plane_bits);
  const int code_bits = plane_bits;           // 14 / 15: the 16-plane operand with its top planes zero -- everything but k_mh_compare_a16
  if (plane_bits == 14 || plane_bits == 15) plane_bits = 16;   // treats it as 16 planes
  const int T = (int)ceil_div(n, K2_TILE);
  const int TR = (int)ceil_div(ceil_div(row_end - row_begin, K2_TILE), tile_stride);
  const int64_t ntiles = count_tiles(TR, T, symmetric);
  const int64_t per_xcd = ceil_div(ntiles, 8);
  // row-block modes: bands of `band` tile rows dealt round-robin to the 8 XCDs; keep >= ~6 bands per
  // XCD so the deal is even (a band's column operand is re-read from L2 `band` times, so not smaller than needed)
  int band = K2_BAND;
  while (band > 1 && ceil_div(TR, band) < 48) band >>= 1;
  const int64_t nblocks = symmetric ? per_xcd * 8 : 8 * ceil_div(ceil_div(TR, band), 8) * (int64_t)band * T;
  if (nblocks > 0x7fffffffLL) return fail(DA_ERR_UNSUPPORTED, "pair space too large for one launch");
  dim3 grid((unsigned)nblocks), block(K2_THREADS);
  // symmetric 12-plane compares: interior tiles by the hand-scheduled kernel, the rest by the general one
  // (float64 output: its count -> double table of n_hash + 1 entries lives in the kernel's 36 KiB ring, K2_A12_TABLE_MAX
  // doubles; a larger n_hash stays with k_mh_compare, which divides directly when its table does not fit)
  const bool a12 = symmetric && plane_bits == 12 && !config().k2_no_asm && (ld & 1) == 0 &&
                   (kind != DA_OUT_F64 || (int64_t)n_hash + 1 <= K2_A12_TABLE_MAX) &&
                   (reinterpret_cast<uintptr_t>(d_out) & (kind == DA_OUT_F64 ? 15 : 3)) == 0;   // = a12_takes' alignment test
  // symmetric 16-plane compares likewise (k_mh_compare_a16 on the padded twin of the operand)
  const bool a16 = symmetric && plane_bits == 16 && !config().k2_no_asm && (ld & 1) == 0 &&
                   (kind != DA_OUT_F64 || (int64_t)n_hash + 1 <= K2_A16_TABLE_MAX) &&
                   (reinterpret_cast<uintptr_t>(d_out) & (kind == DA_OUT_F64 ? 15 : 3)) == 0;
  // symmetric 8-plane compares (the dense half of the heavy / rare split): the persistent kernel's 8-plane block, one tile per workgroup
  // (wg_per_xcd = per_xcd: the grid and tile order of k_mh_compare_a12); float64 needs its table (16 KiB beside the 24 KiB ring: n_hash <= 2047), two stages at least
  const bool a8 = symmetric && plane_bits == 8 && !config().k2_no_asm && (ld & 1) == 0 && n_hash > K2_GROUP &&
                  (kind != DA_OUT_F64 || n_hash < K2_P8_TABLE) && per_xcd < 0x7fffffffLL &&
                  (reinterpret_cast<uintptr_t>(d_out) & (kind == DA_OUT_F64 ? 15 : 3)) == 0;
  if (a8 && config().k2_persist) {            // DYNAALIGN_K2_PERSIST=1 (experiment): four resident workgroups per CU walk the tiles
    int wg_per_xcd = 4 * 32;
    if (config().k2_wg_per_cu > 0) wg_per_xcd = config().k2_wg_per_cu * 32;
    if ((int64_t)wg_per_xcd > per_xcd) wg_per_xcd = (int)(per_xcd > 0 ? per_xcd : 1);
    const dim3 pgrid((unsigned)(8 * wg_per_xcd));
    if (kind == DA_OUT_F64)
      hipLaunchKernelGGL((k_mh_compare_p12<true, 8, false>), pgrid, block, 0, stream, d_planes, n, n_hash, d_out, ld, ntiles, per_xcd, wg_per_xcd, (int64_t)0);
    else
      hipLaunchKernelGGL((k_mh_compare_p12<false, 8, false>), pgrid, block, 0, stream, d_planes, n, n_hash, d_out, ld, ntiles, per_xcd, wg_per_xcd, (int64_t)0);
  } else if (a8) {
    if (kind == DA_OUT_F64)
      hipLaunchKernelGGL((k_mh_compare_p12<true, 8, true>), grid, block, 0, stream, d_planes, n, n_hash, d_out, ld, ntiles, per_xcd, (int)per_xcd, (int64_t)0);
    else
      hipLaunchKernelGGL((k_mh_compare_p12<false, 8, true>), grid, block, 0, stream, d_planes, n, n_hash, d_out, ld, ntiles, per_xcd, (int)per_xcd, (int64_t)0);
  }
  if (a16) {
#define DA_A16(F, B) hipLaunchKernelGGL((k_mh_compare_a16<F, B>), grid, block, 0, stream, d_planes, n, n_hash, d_out, ld, ntiles, per_xcd)
    if (kind == DA_OUT_F64) { if (code_bits == 14) DA_A16(true, 14); else if (code_bits == 15) DA_A16(true, 15); else DA_A16(true, 16); }
    else { if (code_bits == 14) DA_A16(false, 14); else if (code_bits == 15) DA_A16(false, 15); else DA_A16(false, 16); }
#undef DA_A16
  }
  // the persistent form of that kernel: >= 2 stages per tile, float64 needs its 4 KiB table (n_hash <= 511)
  // OPT-IN (DYNAALIGN_K2_PERSIST=1): measured slower than one tile per workgroup on MI355X -- 28.8 vs 24.9 ms (float64),
  // 26.3 vs 22.2 ms (uint16) at N = 100k (profiles/r02_c_k2_persistent_*.json): a wave that does not exit has to
  // wait for its own stores (vmcnt is in order), and a CU drains only ~20 GB/s of stores, so the 256 KiB of a tile take
  // ~13 us either way; with one tile per workgroup that time is spent by an exiting workgroup while the slot's successor
  // already loads.  Kept because it is bit-exact, tested, and the structure the next step needs (stores interleaved into
  // the following tile's stage loop).
  const bool p12 = a12 && n_hash > K2_GROUP && (kind != DA_OUT_F64 || n_hash < K2_P12_TABLE) && config().k2_persist;
  if (p12) {
    static std::atomic<int> occ_cache[2], cus_cache;              // resident workgroups per CU (4 expected), CUs of the device;
    const int ki = kind == DA_OUT_F64 ? 0 : 1;                    // (atomics: the multi-device entry points launch from several host threads)
    if (!occ_cache[ki].load()) {
      int occ = 0;
      if (ki == 0) DA_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_mh_compare_p12<true>, K2_THREADS, 0));
      else DA_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_mh_compare_p12<false>, K2_THREADS, 0));
      int dev = 0;
      hipDeviceProp_t prop;
      DA_HIP_TRY(hipGetDevice(&dev));
      DA_HIP_TRY(hipGetDeviceProperties(&prop, dev));
      if (config().k2_wg_per_cu > 0) occ = config().k2_wg_per_cu;   // DYNAALIGN_K2_WG_PER_CU
      cus_cache.store(prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
      occ_cache[ki].store(occ > 0 ? occ : 1);
      if (config().trace)
        fprintf(stderr, "[dynaalign] k_mh_compare_p12<%s>: %d resident workgroups per CU, %d CUs\n", ki == 0 ? "f64" : "u16", occ_cache[ki].load(), cus_cache.load());
    }
    int wg_per_xcd = occ_cache[ki].load() * ((cus_cache.load() + 7) / 8);
    if ((int64_t)wg_per_xcd > per_xcd) wg_per_xcd = (int)(per_xcd > 0 ? per_xcd : 1);
    const dim3 pgrid((unsigned)(8 * wg_per_xcd));
#ifdef K2_DYNAMIC
    {
      static const unsigned int zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      DA_HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_k2_next), zeros, sizeof(zeros), 0, hipMemcpyHostToDevice, stream));
    }
#endif
    if (kind == DA_OUT_F64)
      hipLaunchKernelGGL(k_mh_compare_p12<true>, pgrid, block, 0, stream, d_planes, n, n_hash, d_out, ld, ntiles, per_xcd, wg_per_xcd, (int64_t)0);
    else
      hipLaunchKernelGGL(k_mh_compare_p12<false>, pgrid, block, 0, stream, d_planes, n, n_hash, d_out, ld, ntiles, per_xcd, wg_per_xcd, (int64_t)0);
  } else if (a12) {
    bool roles = false;
#ifdef DA_K2_EXPERIMENTS
    // (experiment library only, tools/experiments/: DYNAALIGN_K2_INLOOP=1 -> k_mh_compare_q12, DYNAALIGN_K2_ROLES=1 -> k_mh_compare_r12)
    const int nst12 = (n_hash + K2_GROUP - 1) / K2_GROUP;
    if (kind == DA_OUT_F64 && ntiles < 0x7fffffffLL && T < 65536) {
      int rc_r = DA_OK;
      if (getenv("DYNAALIGN_K2_ROLES") && n_hash > 2 * K2_GROUP && n_hash < R12_TABLE)
        rc_r = launch_persistent_f64(true, d_planes, n, n_hash, static_cast<double *>(d_out), ld, ntiles, per_xcd, T, stream, &roles);
      else if (getenv("DYNAALIGN_K2_INLOOP") && (nst12 == 16 || nst12 == 8 || nst12 == 4) && ld < ((int64_t)1 << 24))
        rc_r = launch_persistent_f64(false, d_planes, n, n_hash, static_cast<double *>(d_out), ld, ntiles, per_xcd, T, stream, &roles);
      if (rc_r != DA_OK) return rc_r;
    }
#endif
    if (roles) {
    } else if (kind == DA_OUT_F64)
      hipLaunchKernelGGL(k_mh_compare_a12<true>, grid, block, 0, stream, d_planes, n, n_hash, d_out, ld, ntiles, per_xcd);
    else
      hipLaunchKernelGGL(k_mh_compare_a12<false>, grid, block, 0, stream, d_planes, n, n_hash, d_out, ld, ntiles, per_xcd);
  }
  // shard / row-block modes with uint16 output (what the sharded routes compute per rank): the same loop behind the rectangular tile geometry
  const bool s12 = !symmetric && plane_bits == 12 && kind == DA_OUT_COMPACT && !config().k2_no_asm && (ld & 1) == 0 &&
                   (row_begin % K2_TILE) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 3) == 0;
  if (s12)
    hipLaunchKernelGGL(k_mh_compare_s12, grid, block, 0, stream, d_planes, n, n_hash, row_begin, row_end, tile_stride, upper_only ? 1 : 0, TR,
                       static_cast<uint16_t *>(d_out), ld, fold_q, fold_w, band);
  // ... and with 14 - 16 code bits (k_mh_compare_a16's block behind the same geometry)
  const bool s16 = !symmetric && plane_bits == 16 && kind == DA_OUT_COMPACT && !config().k2_no_asm && (ld & 1) == 0 &&
                   (row_begin % K2_TILE) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 3) == 0;
  if (s16) {
#define DA_S16(B) hipLaunchKernelGGL((k_mh_compare_s16<B>), grid, block, 0, stream, d_planes, n, n_hash, row_begin, row_end, tile_stride, upper_only ? 1 : 0, TR, \
                                     static_cast<uint16_t *>(d_out), ld, fold_q, fold_w, band)
    if (code_bits == 14) DA_S16(14); else if (code_bits == 15) DA_S16(15); else DA_S16(16);
#undef DA_S16
  }
  const int only_edge = (a12 || a16 || a8 || s12 || s16) ? 1 : 0;
  // what is left for the general kernel then: the diagonal tiles + the last tile column, enumerated directly
  if (a12 || a16 || a8) grid = dim3((unsigned)(2 * (int64_t)T - 1));
#define DA_K2(SYM, F64, PL)                                                                              \
  hipLaunchKernelGGL((k_mh_compare<SYM, F64, PL>), grid, block, 0, stream, d_planes, n, n_hash, \
                     row_begin, row_end, tile_stride, upper_only ? 1 : 0, TR, d_out, ld, ntiles, per_xcd, fold_q, fold_w, band, only_edge)
#define DA_K2_PL(PL)                                                                     \
  do {                                                                                   \
    if (symmetric) { if (kind == DA_OUT_F64) DA_K2(true, true, PL); else DA_K2(true, false, PL); } \
    else           { if (kind == DA_OUT_F64) DA_K2(false, true, PL); else DA_K2(false, false, PL); } \
  } while (0)
  switch (plane_bits) {
    case 8: DA_K2_PL(8); break;
    case 12: DA_K2_PL(12); break;
    case 16: DA_K2_PL(16); break;
    default: DA_K2_PL(32); break;
  }
#undef DA_K2_PL
#undef DA_K2
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

// ---- the symmetric uint16 12-plane compare in pieces (pipelined duplicate route, api.cpp) --------------------------------
// decode_tile numbers the symmetric tiles band by band (K2_BAND tile rows, every tile right of the diagonal), so the tile rows
// [8 b0, 8 b1) are the id range [prefix(b0), prefix(b1)) and, bands taken in order, the table rows of band b are complete (direct
// and mirrored stores) once the bands <= b have run.
int64_t mh_sym_band_prefix(int64_t n, int64_t band) {
  const int64_t T = ceil_div(n, K2_TILE), S = K2_BAND, nfull = T / S;
  if (band <= 0) return 0;
  if (band > nfull) return T * (T + 1) / 2;
  const int64_t c0 = S * (S + 1) / 2 + (T - S) * S;
  return band * c0 - (S * S / 2) * band * (band - 1);
}
int64_t mh_sym_bands(int64_t n) { return ceil_div(ceil_div(n, K2_TILE), K2_BAND); }
// tests (host only, no device): the id range of a band, and decode_tile's host twin for one id
extern "C" int64_t da_debug_sym_band_prefix(int64_t n, int64_t band) { return mh_sym_band_prefix(n, band); }
extern "C" int da_debug_decode_sym_tile(int64_t L, int T, int *ti, int *tj) {
  const TileId t = decode_tile(L, T, T, true);
  if (ti) *ti = t.ti;
  if (tj) *tj = t.tj;
  return t.valid ? 1 : 0;
}
bool mh_compare_bands_ok(int64_t n, int n_hash, int plane_bits, const void *d_out, int64_t ld) {
  const int64_t T = ceil_div(n, K2_TILE);
  return (plane_bits == 12 || plane_bits == 8) && n_hash > K2_GROUP && n_hash <= 65535 && !config().k2_no_asm && (ld & 1) == 0 &&
         (reinterpret_cast<uintptr_t>(d_out) & 3) == 0 && T * (T + 1) / 2 < 0x7fffffffLL;
}
// interior tiles of the bands [band_begin, band_end) by the persistent kernel with at most wg_per_cu resident workgroups per CU:
// a grid that small leaves the rest of every CU to kernels of other streams (the expansion's stores)
int launch_mh_compare_bands_u16(const uint32_t *d_planes, int64_t n, int n_hash, uint16_t *d_out, int64_t ld, int64_t band_begin,
                                int64_t band_end, int wg_per_cu, hipStream_t stream, int plane_bits) {
  if (!mh_compare_bands_ok(n, n_hash, plane_bits, d_out, ld)) return fail(DA_ERR_UNSUPPORTED, "banded compare: shape not covered");
  const int64_t t0 = mh_sym_band_prefix(n, band_begin), t1 = mh_sym_band_prefix(n, band_end);
  if (t1 <= t0) return DA_OK;
  static std::atomic<int> cus_cache;
  if (!cus_cache.load()) {
    int dev = 0;
    hipDeviceProp_t prop;
    DA_HIP_TRY(hipGetDevice(&dev));
    DA_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    cus_cache.store(prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
  }
  const int64_t per_xcd = ceil_div(t1 - t0, 8);
  int wg_per_xcd = std::max(1, std::min(wg_per_cu, 4)) * ((cus_cache.load() + 7) / 8);
  if ((int64_t)wg_per_xcd > per_xcd) wg_per_xcd = (int)per_xcd;
  if (plane_bits == 8)
    hipLaunchKernelGGL((k_mh_compare_p12<false, 8, false>), dim3((unsigned)(8 * wg_per_xcd)), dim3(K2_THREADS), 0, stream, d_planes, n, n_hash,
                       static_cast<void *>(d_out), ld, t1, per_xcd, wg_per_xcd, t0);
  else
    hipLaunchKernelGGL(k_mh_compare_p12<false>, dim3((unsigned)(8 * wg_per_xcd)), dim3(K2_THREADS), 0, stream, d_planes, n, n_hash,
                       static_cast<void *>(d_out), ld, t1, per_xcd, wg_per_xcd, t0);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}
// the tiles the kernel above leaves everywhere: diagonal tiles and the last tile column (with their mirrors)
int launch_mh_compare_edges_u16(const uint32_t *d_planes, int64_t n, int n_hash, uint16_t *d_out, int64_t ld, hipStream_t stream, int plane_bits) {
  const int T = (int)ceil_div(n, K2_TILE);
  const int64_t ntiles = count_tiles(T, T, true);
  if (plane_bits == 8)
    hipLaunchKernelGGL((k_mh_compare<true, false, 8>), dim3((unsigned)(2 * (int64_t)T - 1)), dim3(K2_THREADS), 0, stream, d_planes, n, n_hash,
                       (int64_t)0, n, 1, 0, T, static_cast<void *>(d_out), ld, ntiles, ceil_div(ntiles, 8), 0, (int64_t)0, K2_BAND, 1);
  else
  hipLaunchKernelGGL((k_mh_compare<true, false, 12>), dim3((unsigned)(2 * (int64_t)T - 1)), dim3(K2_THREADS), 0, stream, d_planes, n, n_hash,
                     (int64_t)0, n, 1, 0, T, static_cast<void *>(d_out), ld, ntiles, ceil_div(ntiles, 8), 0, (int64_t)0, K2_BAND, 1);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

// Store phase shared by the three "64 x 64 uint16 tile in LDS -> float64 output, direct + mirrored" kernels below.  Interior
// off-diagonal tiles of an even-ld, 16-byte aligned matrix take 16-byte streaming stores (two adjacent doubles per lane: a
// wave-instruction covers two 512-byte row pieces; 8-byte stores run at 0.5-0.7x the rate, MI355X_MICROARCH.md); everything
// else (diagonal tiles, borders, odd ld) the bounds-checked scalar path.
template <int FT, typename WIDEN>
__device__ __forceinline__ void store_tile_f64(const uint16_t (&t)[FT][FT + 2], WIDEN widen, double *__restrict__ out, int64_t ld, int n,
                                                int i0, int j0) {
  const int tx = threadIdx.x & (FT - 1), ty = threadIdx.x / FT;  // FT x (256 / FT) (scalar path)
  const bool fast = i0 != j0 && i0 + FT <= n && j0 + FT <= n && (ld & 1) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                    ((i0 | j0) & 1) == 0;
  if (fast) {
    constexpr int PAIRS = FT / 2, RG = 256 / PAIRS;                     // column pairs per row; row groups
    const int k2 = 2 * (threadIdx.x % PAIRS), rg = threadIdx.x / PAIRS;
#pragma unroll 8
    for (int r = rg; r < FT; r += RG)
      nt_store2(out + (int64_t)(i0 + r) * ld + j0 + k2, widen(t[r][k2]), widen(t[r][k2 + 1]));          // upper part as computed
#pragma unroll 8
    for (int r = rg; r < FT; r += RG)
      nt_store2(out + (int64_t)(j0 + r) * ld + i0 + k2, widen(t[k2][r]), widen(t[k2 + 1][r]));          // out[j][i] = upper(i, j)
    return;
  }
  constexpr int RS = 256 / FT;
  const int j = j0 + tx;
  double *o = out + (int64_t)i0 * ld + j;
  for (int r = ty; r < FT; r += RS) {
    const int i = i0 + r;
    if (i < n && j < n && j >= i) o[(int64_t)r * ld] = widen(t[r][tx]);  // upper part as computed
  }
  const int im = i0 + tx;                                                // out[j][i] = upper(i, j)
  double *om = out + (int64_t)j0 * ld + im;
  for (int r = ty; r < FT; r += RS) {
    const int jm = j0 + r;
    if (im < n && jm < n && jm > im) om[(int64_t)r * ld] = widen(t[tx][r]);
  }
}

// Where row r of the unique table lives.  One GPU: row r.  The sharded NW route all-gathers row blocks of cyclic 128-row units
// (rank p computed units p, p + world, ...; every block holds rows_local rows), so row r sits in block (r / 128) % world.
struct TableRows { int world; int64_t rows_local; };
__device__ __forceinline__ int64_t table_row(int r, const TableRows &tr) {
  if (tr.world <= 1) return r;
  const int t = r >> 7;
  return (int64_t)(t % tr.world) * tr.rows_local + (int64_t)(t / tr.world) * 128 + (r & 127);
}

// 64 x 64 tile of block b when the interior 128-tiles were taken by a 128 x 128 kernel: quarter (b & 3) of 128-tile b >> 2, the
// T128 diagonal tiles first, then the last tile column when n is not a multiple of 128 (grid: leftover_blocks64)
__device__ __forceinline__ TileId leftover_tile64(unsigned b, int n, int TB) {
  const int T128 = (n + 127) >> 7, t = (int)(b >> 2), sub = (int)(b & 3);
  const int ti128 = t < T128 ? t : t - T128, tj128 = t < T128 ? t : T128 - 1;
  TileId o;
  o.ti = 2 * ti128 + (sub >> 1); o.tj = 2 * tj128 + (sub & 1);
  o.valid = o.ti <= o.tj && o.ti < TB && o.tj < TB;
  return o;
}
__device__ __forceinline__ TileId decode_tile_xcd(unsigned b, int64_t per_xcd, int64_t ntiles, int TB) {
  const int64_t L = (int64_t)(b & 7) * per_xcd + (b >> 3);
  if (L >= ntiles) { TileId o; o.ti = o.tj = 0; o.valid = false; return o; }
  return decode_tile(L, TB, TB, true);
}
static unsigned leftover_blocks64(int64_t n) {
  const int T128 = (int)ceil_div(n, 128);
  return (unsigned)(4 * (T128 + ((n & 127) ? T128 - 1 : 0)));
}
constexpr int ER_STRIDE = 272;   // bytes per staged tile row: 256 + 16 -- the lanes' 4-byte reads (row 2 ty, word tx) fall into 64 distinct banks
// A 128 x 128 tile of uint16 codes staged in LDS (row stride ER_STRIDE) -> out[I0.., J0..] and its mirror image, float64.  Lane
// (tx, ty) of the 16 x 16 lane grid owns rows 32 g + 2 ty + {0, 1} x columns 32 q + 2 tx + {0, 1} (k_mh_compare's epilogue):
// direct rows go out as 16-byte stores of two adjacent columns, mirrored rows as 16-byte stores of two adjacent ROWS at one column.
template <typename WIDEN>
__device__ __forceinline__ void er_store_tile(const unsigned char *er_lds, WIDEN widen, double *__restrict__ out, int64_t ld, int64_t I0,
                                              int64_t J0, int tx, int ty) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    double v0[8], v1[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t w0 = *reinterpret_cast<const uint32_t *>(er_lds + (32 * g + 2 * ty) * ER_STRIDE + (32 * q + 2 * tx) * 2);
      const uint32_t w1 = *reinterpret_cast<const uint32_t *>(er_lds + (32 * g + 2 * ty + 1) * ER_STRIDE + (32 * q + 2 * tx) * 2);
      v0[2 * q] = widen(w0 & 0xffffu); v0[2 * q + 1] = widen(w0 >> 16);
      v1[2 * q] = widen(w1 & 0xffffu); v1[2 * q + 1] = widen(w1 >> 16);
    }
    double *orow = out + (I0 + 32 * g + 2 * ty) * ld + (J0 + 2 * tx);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      nt_store2(orow + 32 * q, v0[2 * q], v0[2 * q + 1]);
      nt_store2(orow + ld + 32 * q, v1[2 * q], v1[2 * q + 1]);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c)
      nt_store2(out + (J0 + 32 * (c >> 1) + 2 * tx + (c & 1)) * ld + (I0 + 32 * g + 2 * ty), v0[c], v1[c]);
  }
}
__device__ __forceinline__ bool expand_fast_takes(int ti128, int tj128, int n, int64_t ld, const void *out) {
  return ti128 != tj128 && (ti128 + 1) * 128 <= n && (tj128 + 1) * 128 <= n && (ld & 1) == 0 &&
         (reinterpret_cast<uintptr_t>(out) & 15) == 0;
}

// Gathered shards -> final matrix, interior off-diagonal 128 x 128 tiles (geom.tile = 128: a tile lives in ONE rank's block):
// 128 row pieces of 128 codes are staged in LDS with 16-byte loads (uint16 blocks) or 8 low bytes + the bit planes' byte per
// 8 columns (packed blocks), then er_store_tile writes both halves with 16-byte streaming stores -- the pattern that reaches
// 5.7-6.4 TB/s in k_expand_rows (the 64 x 64 LDS-transpose kernels below: 3.4-3.9 TB/s; they keep diagonal / border tiles).
template <bool IS_NW, bool PACKED>
__global__ __launch_bounds__(256, 4) void k_finalize_rows(const void *__restrict__ G_v, int64_t ld_g_or_block_bytes, ShardGeom geom, int nhi,
                                                          int n_hash, int tab_entries, double *__restrict__ out, int64_t ld, int T128,
                                                          int64_t ntiles, int64_t per_xcd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char er_lds[];   // 128 x ER_STRIDE tile, then the value table (MH)
  double *tab = reinterpret_cast<double *>(er_lds + 128 * ER_STRIDE);
  const int n = (int)geom.n;
  const int64_t L = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (L >= ntiles) return;
  const TileId tt = decode_tile(L, T128, T128, true);
  if (!tt.valid || !expand_fast_takes(tt.ti, tt.tj, n, ld, out)) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tx = ((wave & 1) << 3) + (lane & 7), ty = ((wave >> 1) << 3) + (lane >> 3);
  const int64_t I0 = (int64_t)tt.ti * 128, J0 = (int64_t)tt.tj * 128;
  const int tr = tt.ti, q_loc = tr / geom.world, owner = tr - q_loc * geom.world;
  const bool front = q_loc <= geom.Q - 1 - q_loc;
  const int64_t lrow0 = (int64_t)(front ? q_loc : geom.Q - 1 - q_loc) * 128;               // row inside the owner's block
  const int64_t col0 = (front ? -(int64_t)tr * 128 : shard_back(geom.W, geom.n)) + J0;     // a multiple of 8
  const int unit = tid & 15, row0 = tid >> 4;
  uint4 st[8];
  if (!PACKED) {
    const uint16_t *G = static_cast<const uint16_t *>(G_v);
    const int64_t ld_g = ld_g_or_block_bytes;
    const uint16_t *src = G + ((int64_t)owner * geom.rows + lrow0 + row0) * ld_g + col0 + 8 * unit;
#pragma unroll
    for (int q = 0; q < 8; ++q) st[q] = *reinterpret_cast<const uint4 *>(src + (int64_t)(16 * q) * ld_g);
  } else {
    const uint8_t *blk = static_cast<const uint8_t *>(G_v) + (int64_t)owner * ld_g_or_block_bytes;
    const int64_t W = geom.W, groups = W >> 3, rows = geom.rows, col = col0 + 8 * unit;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int64_t row = lrow0 + row0 + 16 * q;
      const uint2 lo = *reinterpret_cast<const uint2 *>(blk + row * W + col);
      uint32_t hi[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = 0; k < nhi; ++k) {
        const uint32_t b = blk[rows * W + ((int64_t)k * rows + row) * groups + (col >> 3)];
#pragma unroll
        for (int e = 0; e < 8; ++e) hi[e] |= ((b >> e) & 1u) << (8 + k);
      }
      st[q].x = ((lo.x & 255u) | hi[0]) | ((((lo.x >> 8) & 255u) | hi[1]) << 16);
      st[q].y = (((lo.x >> 16) & 255u) | hi[2]) | (((lo.x >> 24) | hi[3]) << 16);
      st[q].z = ((lo.y & 255u) | hi[4]) | ((((lo.y >> 8) & 255u) | hi[5]) << 16);
      st[q].w = (((lo.y >> 16) & 255u) | hi[6]) | (((lo.y >> 24) | hi[7]) << 16);
    }
  }
  if (!IS_NW)
    for (int e = tid; e < tab_entries; e += 256) tab[e] = (double)e / (double)n_hash;              // src/minHash.cpp:174
#pragma unroll
  for (int q = 0; q < 8; ++q) *reinterpret_cast<uint4 *>(er_lds + (row0 + 16 * q) * ER_STRIDE + unit * 16) = st[q];
  __syncthreads();
  auto widen = [&](uint32_t x) -> double {
    if (!IS_NW) return tab_entries ? tab[x] : (double)x / (double)n_hash;
    const uint32_t ln = x & 255u;
    if (ln == 0) return __longlong_as_double(0xFFF8000000000000ULL);     // 0/0 as on the reference's host
    return (double)(x >> 8) / (double)ln;                                // src/pairwiseSeqAlign.cpp:311
  };
  er_store_tile(er_lds, widen, out, ld, I0, J0, tx, ty);
}
// the interior tiles go to k_finalize_rows when the geometry allows it; the 64 x 64 kernels then visit only what is left
static bool finalize_rows_ok(const ShardGeom &geom, const void *d_g, int64_t ld_g, const double *d_out, int64_t ld, bool packed) {
  return geom.tile == 128 && geom.n >= 256 && (ld & 1) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(d_g) & 15) == 0 && (packed || (ld_g & 7) == 0);
}

// Gathered shards -> final matrix.  G holds, for every rank p, its folded local block
// (ShardGeom); out[i][j] = widen(G[entry of (min(i,j), max(i,j))]).  One workgroup per 64 x 64
// tile on or above the diagonal, enumerated like the compare kernel's tiles (bands of 8 tile rows,
// column-major inside a band, one id range per XCD): the mirrored 512-byte pieces of 8 consecutive
// workgroups are 4 KiB of one output row and the direct pieces of a band's next column follow on
// (row-major tile order: 23.5 ms, this order: 21 ms at N = 100k; a 128 x 128 variant with the
// compare kernel's 16-byte store pattern and 140 VGPRs ran 30 ms -- too few waves to hide the
// read -> LDS -> store chain; a low-register 128 x 128 variant whose wave-wide stores cover 1 KiB
// row pieces: 26.7 ms; persistent workgroups: 23.6 ms).  64 divides the shard tile (128 / 64), so which rank's block and
// which folded row a tile reads from is workgroup-uniform integer arithmetic.  MH widens through
// an LDS table built with the reference's divide.
// uint16 tile as it is (the sharded duplicate route rebuilds the symmetric table of the unique strings from the gathered shards)
template <int FT>
__device__ __forceinline__ void store_tile_u16(const uint16_t (&t)[FT][FT + 2], uint16_t *__restrict__ out, int64_t ld, int n, int i0, int j0) {
  constexpr int RS = 256 / FT;
  const int tx = threadIdx.x & (FT - 1), ty = threadIdx.x / FT;
  const int j = j0 + tx;
  uint16_t *o = out + (int64_t)i0 * ld + j;
  for (int r = ty; r < FT; r += RS) {
    const int i = i0 + r;
    if (i < n && j < n && j >= i) o[(int64_t)r * ld] = t[r][tx];
  }
  const int im = i0 + tx;
  uint16_t *om = out + (int64_t)j0 * ld + im;
  for (int r = ty; r < FT; r += RS) {
    const int jm = j0 + r;
    if (im < n && jm < n && jm > im) om[(int64_t)r * ld] = t[tx][r];
  }
}
template <bool IS_NW, bool U16 = false>
__global__ __launch_bounds__(256) void k_finalize_sharded(const uint16_t *__restrict__ G, int64_t ld_g, ShardGeom geom,
                                                          int n_hash, void *__restrict__ out_v, int64_t ld, int TB,
                                                          int64_t ntiles, int64_t per_xcd, int skip_fast) {
  double *out = static_cast<double *>(out_v);
  constexpr int FT = 64, TABLE = 2048;
  __shared__ uint16_t t[FT][FT + 2];
  __shared__ double ratio[IS_NW ? 1 : TABLE];
  const int n = (int)geom.n;
  const TileId tt = skip_fast ? leftover_tile64(blockIdx.x, n, TB) : decode_tile_xcd(blockIdx.x, per_xcd, ntiles, TB);
  if (!tt.valid) return;
  const int i0 = tt.ti * FT, j0 = tt.tj * FT;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  const bool use_table = !IS_NW && n_hash < TABLE;
  if (use_table)
    for (int c = threadIdx.x; c <= n_hash; c += 256) ratio[c] = (double)c / (double)n_hash;   // src/minHash.cpp:174
  auto widen = [&](uint32_t v) -> double {
    if (!IS_NW) return use_table ? ratio[v] : (double)v / (double)n_hash;
    const uint32_t ln = v & 255u;
    if (ln == 0) return __longlong_as_double(0xFFF8000000000000ULL);     // 0/0 as on the reference's host
    return (double)(v >> 8) / (double)ln;                                // src/pairwiseSeqAlign.cpp:311
  };
  // where rows i0 .. i0+63 live: global tile row tr, owner tr % world, folded local row
  const int tile = geom.tile;
  const int tr = i0 / tile;
  const int q = tr / geom.world, owner = tr - q * geom.world;
  const bool front = q <= geom.Q - 1 - q;
  const int64_t lrow0 = (int64_t)owner * geom.rows + (int64_t)(front ? q : geom.Q - 1 - q) * tile + (i0 - tr * tile);
  const int64_t coff = front ? -(int64_t)tr * tile : shard_back(geom.W, geom.n);
  const uint16_t *src = G + lrow0 * ld_g + coff + j0 + tx;
  const int j = j0 + tx;
  for (int r = ty; r < FT; r += 4) {
    const int i = i0 + r;
    if (i < n && j < n && j >= i) t[r][tx] = src[(int64_t)r * ld_g];
  }
  __syncthreads();
  if (U16) store_tile_u16<64>(t, static_cast<uint16_t *>(out_v), ld, n, i0, j0);
  else store_tile_f64<64>(t, widen, out, ld, n, i0, j0);
}

// Dense symmetric n x n result from the square table D of UNIQUE sequences (NW dedupe, nw_kernels.hip):
//   out[i][j] = value(D[uidx[min(i,j)]][uidx[max(i,j)]])      (D is ordered: row = sequence1)
// 64 x 64 output tiles on or above the diagonal in the banded XCD order of k_finalize_sharded, gathered through LDS so that
// the mirrored half is written as row pieces too.  Single-copy strings are numbered in input order, so most of a row's
// gather hits consecutive table columns; the multi-copy ones sit in the first few KiB of every table row.
template <bool F64, bool IS_NW, int FT>
__global__ __launch_bounds__(256) void k_expand_unique(const uint16_t *__restrict__ D, int64_t ld_d, const int32_t *__restrict__ uidx,
                                                       int n, int n_hash, void *__restrict__ out_v, int64_t ld, int TB,
                                                       int64_t ntiles, int64_t per_xcd, int skip_fast, TableRows trows) {
  __shared__ uint16_t t[FT][FT + 2];
  __shared__ int32_t ur[FT], uc[FT];
  const TileId tt = skip_fast ? leftover_tile64(blockIdx.x, n, TB) : decode_tile_xcd(blockIdx.x, per_xcd, ntiles, TB);
  if (!tt.valid) return;
  const int i0 = tt.ti * FT, j0 = tt.tj * FT;
  constexpr int RS = 256 / FT;
  const int tx = threadIdx.x & (FT - 1), ty = threadIdx.x / FT;  // FT x (256 / FT)
  for (int q = threadIdx.x; q < 2 * FT; q += 256) {
    if (q < FT) ur[q] = (i0 + q < n) ? uidx[i0 + q] : 0;
    else uc[q - FT] = (j0 + q - FT < n) ? uidx[j0 + q - FT] : 0;
  }
  __syncthreads();
  auto widen = [&](uint32_t v) -> double {
    if (!IS_NW) return (double)v / (double)n_hash;
    const uint32_t ln = v & 255u;
    if (ln == 0) return __longlong_as_double(0xFFF8000000000000ULL);     // 0/0 as on the reference's host
    return (double)(v >> 8) / (double)ln;                                // src/pairwiseSeqAlign.cpp:311
  };
  const int j = j0 + tx;
  const int32_t cj = uc[tx];
  for (int r = ty; r < FT; r += RS) {
    const int i = i0 + r;
    if (i < n && j < n && j >= i) t[r][tx] = D[table_row(ur[r], trows) * ld_d + cj];
  }
  __syncthreads();
  if (F64) {
    store_tile_f64<FT>(t, widen, reinterpret_cast<double *>(out_v), ld, n, i0, j0);
  } else {
    uint16_t *out = reinterpret_cast<uint16_t *>(out_v);
    uint16_t *o = out + (int64_t)i0 * ld + j;
    for (int r = ty; r < FT; r += RS) {
      const int i = i0 + r;
      if (i < n && j < n && j >= i) o[(int64_t)r * ld] = t[r][tx];
    }
    const int im = i0 + tx;
    uint16_t *om = out + (int64_t)j0 * ld + im;
    for (int r = ty; r < FT; r += RS) {
      const int jm = j0 + r;
      if (im < n && jm < n && jm > im) om[(int64_t)r * ld] = t[tx][r];
    }
  }
}

// The same expansion for the interior off-diagonal 128 x 128 tiles (99 % of the result at N = 100k), as two streaming passes.
// The one-kernel form above gathers every output entry by its own 2-byte read -- 64 load instructions per lane and tile, each
// touching up to 64 lines: 24 ms for the 80 GB result (3.4 TB/s).  Here
//   k_gather_columns:  F[r][j] = D[r][uidx[j]]  for every unique row r and every column j right of r's first tile -- the row of D
//                      sits in LDS, so the random access costs LDS cycles, and both global sides are coalesced;
//   k_expand_rows:     out[i][j] = out[j][i] = value(F[uidx[i]][j]) -- a tile is 128 row pieces of 256 consecutive bytes (16-byte
//                      loads -> LDS), every lane then owns an 8 x 8 block exactly like k_mh_compare's epilogue: no transpose, the
//                      mirrored 16-byte store of rows r, r+1 at one column uses values the lane already holds.
// The tiles they take (expand_fast_takes) are skipped by k_expand_unique, which keeps diagonal and border tiles.
constexpr int GC_THREADS = 1024;
// one resident workgroup per CU walks the unique rows r = blockIdx.x, + gridDim.x, ...: the NEXT row is already on its way into
// registers while the current one is gathered out of LDS (U <= 65536 -> at most 8 16-byte units per thread)
__global__ __launch_bounds__(GC_THREADS) void k_gather_columns(const uint16_t *__restrict__ D, int64_t ld_d, const int32_t *__restrict__ uidx,
                                                               const int32_t *__restrict__ ufirst, int n, int U, uint16_t *__restrict__ F,
                                                               int64_t ld_f, TableRows trows, int from_first_tile, int row_begin) {
  extern __shared__ __attribute__((aligned(16))) uint16_t gc_row[];   // ld_d entries; the rows [row_begin, U) are gathered
  const int units = (int)(ld_d >> 3);
  uint4 pre[8];
#define GC_FETCH(row)                                                                                   \
  {                                                                                                     \
    const uint4 *src_ = reinterpret_cast<const uint4 *>(D + table_row((row), trows) * ld_d);            \
    _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                     \
      const int u = (int)threadIdx.x + q * GC_THREADS;                                                  \
      pre[q] = u < units ? src_[u] : make_uint4(0, 0, 0, 0);                                            \
    }                                                                                                   \
  }
  int r = row_begin + (int)blockIdx.x;
  if (r < U) GC_FETCH(r)
  const int2 *u2 = reinterpret_cast<const int2 *>(uidx);
  const int j2_end = n >> 1;                                         // column pairs; an odd last column is written by itself below
  for (; r < U; r += gridDim.x) {
    __syncthreads();                                                 // the previous row's gather has left the LDS row
    uint4 *dst = reinterpret_cast<uint4 *>(gc_row);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int u = (int)threadIdx.x + q * GC_THREADS;
      if (u < units) dst[u] = pre[q];
    }
    __syncthreads();
    if (r + (int)gridDim.x < U) GC_FETCH(r + (int)gridDim.x)
    const int j_begin = ((ufirst[r] >> 7) + (from_first_tile ? 0 : 1)) << 7;   // first column of the tile right of r's first occurrence:
    uint32_t *frow = reinterpret_cast<uint32_t *>(F + (int64_t)r * ld_f);   // interior tiles never read this row left of it
    for (int j2 = (j_begin >> 1) + threadIdx.x; j2 < j2_end; j2 += GC_THREADS) {
      const int2 c = u2[j2];
      __builtin_nontemporal_store((uint32_t)gc_row[c.x] | ((uint32_t)gc_row[c.y] << 16), frow + j2);
    }
    if ((n & 1) && threadIdx.x == 0 && n - 1 >= j_begin) F[(int64_t)r * ld_f + n - 1] = gc_row[uidx[n - 1]];   // odd n: the last column
  }
#undef GC_FETCH
}

template <bool IS_NW>
__global__ __launch_bounds__(256, 4) void k_expand_rows(const uint16_t *__restrict__ F, int64_t ld_f, const int32_t *__restrict__ uidx,
                                                        int n, int n_hash, int tab_stride, int tab_entries, double *__restrict__ out,
                                                        int64_t ld, int T128, int64_t ntiles, int64_t per_xcd, int64_t tile_begin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char er_lds[];   // 128 x ER_STRIDE tile, then the value table
  double *tab = reinterpret_cast<double *>(er_lds + 128 * ER_STRIDE);
  const int64_t L = tile_begin + (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);   // the ids [tile_begin, ntiles) in 8 runs
  if (L >= ntiles || L >= tile_begin + (int64_t)((blockIdx.x & 7) + 1) * per_xcd) return;
  const TileId tt = decode_tile(L, T128, T128, true);
  if (!tt.valid || !expand_fast_takes(tt.ti, tt.tj, n, ld, out)) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tx = ((wave & 1) << 3) + (lane & 7), ty = ((wave >> 1) << 3) + (lane >> 3);
  const int64_t I0 = (int64_t)tt.ti * 128, J0 = (int64_t)tt.tj * 128;
  // stage: thread t moves 16-byte unit (t & 15) of rows (t >> 4) + 16 q
  uint4 st[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int row = (tid >> 4) + 16 * q;
    st[q] = *(reinterpret_cast<const uint4 *>(F + (int64_t)uidx[I0 + row] * ld_f + J0) + (tid & 15));
  }
  if (IS_NW) {
    for (int e = tid; e < tab_entries; e += 256) {
      const int mt = e / tab_stride, ln = e - mt * tab_stride;
      tab[e] = ln == 0 ? __longlong_as_double(0xFFF8000000000000ULL) : (double)mt / (double)ln;   // src/pairwiseSeqAlign.cpp:311
    }
  } else {
    for (int e = tid; e < tab_entries; e += 256) tab[e] = (double)e / (double)n_hash;              // src/minHash.cpp:174
  }
#pragma unroll
  for (int q = 0; q < 8; ++q)
    *reinterpret_cast<uint4 *>(er_lds + ((tid >> 4) + 16 * q) * ER_STRIDE + (tid & 15) * 16) = st[q];
  __syncthreads();
  auto widen = [&](uint32_t x) -> double {
    if (IS_NW) return tab[(x >> 8) * (uint32_t)tab_stride + (x & 255u)];
    return tab[x];
  };
  er_store_tile(er_lds, widen, out, ld, I0, J0, tx, ty);
}

// ---- ROW expansion: the N x N float64 matrix straight from the U x U count table, no gathered copy (round 3) -----------------
// k_gather_columns + k_expand_rows move the table twice more (9 GB written, 9 GB read) and the tiles' LDS images keep the gather out of
// the CUs.  Here one workgroup holds row r of the (symmetric) table in LDS -- like the gather -- and writes the OUTPUT rows of r's copies
// itself: for every column pair (j, j + 1) the two counts D[r][u(j)], D[r][u(j + 1)] come out of LDS, go through the count -> double table
// (also LDS, built with the reference's divide) and leave as one 16-byte streaming store per copy of r.  Every element of the result is
// written exactly once (diagonal and borders included: no second kernel), a wave-instruction covers 1 KiB of one output row, and the
// only global reads are the table rows (4 GB) and the id map (L2-resident).  Work items are (unique row, up to ES_COPIES of its copies),
// listed by k_es_items so that a string with thousands of copies is spread over many workgroups.
constexpr int ES_THREADS = 1024, ES_COPIES = 4, ES_TICKETS = 128;   // (ticket counters: one per launch of a call, launch_expand_stream_rows' `launch_no`)
__global__ __launch_bounds__(256) void k_es_count(const int32_t *__restrict__ uidx, int n, uint32_t *__restrict__ cnt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) atomicAdd(&cnt[uidx[i]], 1u);
}
// exclusive scans of cnt[] (-> cstart: where the positions of id u start) and of ceil(cnt / ES_COPIES) (-> istart: its work items), one
// workgroup, both sums in the halves of a 64-bit word; cstart[U] = n, istart[U] = number of items
__global__ __launch_bounds__(1024) void k_es_scan(const uint32_t *__restrict__ cnt, int U, uint32_t *__restrict__ cstart, uint32_t *__restrict__ istart) {
  __shared__ uint64_t wsum[16];
  __shared__ uint64_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < U; base += 1024) {
    const int u = base + tid;
    const uint32_t c = u < U ? cnt[u] : 0u;
    const uint64_t v = (uint64_t)c | ((uint64_t)((c + ES_COPIES - 1) / ES_COPIES) << 32);
    uint64_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint64_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const uint64_t carry = carry_s;
    if (u < U) {
      const uint64_t e = carry + woff + x - v;
      cstart[u] = (uint32_t)e;
      istart[u] = (uint32_t)(e >> 32);
    }
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + x;
    __syncthreads();
  }
  if (tid == 0) { cstart[U] = (uint32_t)carry_s; istart[U] = (uint32_t)(carry_s >> 32); }
}
// positions of every id's copies (any order inside an id: the copies receive identical values) and the item list
__global__ __launch_bounds__(256) void k_es_fill(const int32_t *__restrict__ uidx, int n, int U, const uint32_t *__restrict__ cstart,
                                                 const uint32_t *__restrict__ istart, uint32_t *__restrict__ cursor, int32_t *__restrict__ cpos,
                                                 int2 *__restrict__ items) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const int u = uidx[i];
    cpos[cstart[u] + atomicAdd(&cursor[u], 1u)] = i;
  }
  if (i < U) {
    const uint32_t first = istart[i], cnt = istart[i + 1] - first;
    for (uint32_t q = 0; q < cnt; ++q) items[first + q] = make_int2(i, (int)(q * ES_COPIES));
  }
}
// count / n_hash without a table and without the division sequence: with r = RN(1 / b), q0 = RN(a r), e = a - q0 b (exact in one FMA),
// RN(q0 + e r) IS the correctly rounded quotient a / b (Markstein's final step; checked exhaustively for every 0 <= a <= b <= 65535 on the
// host, and on the device by da_debug_ratio_check / tests/test_gpu_mh_dedup.py) -- what src/minHash.cpp:174 computes with a divide.
// The table lookups were the row expansion's heaviest LDS traffic, and LDS bandwidth is what the co-running compare kernel needs.
__device__ __forceinline__ double es_ratio(uint32_t c, double dn, double rcp) {
  const double dc = (double)c, q0 = __dmul_rn(dc, rcp), e = __fma_rn(-q0, dn, dc);
  return __fma_rn(e, rcp, q0);
}
__global__ __launch_bounds__(256) void k_es_ratio_check(int n_hash, double *__restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const double dn = (double)n_hash, rcp = 1.0 / dn;
  if (c <= n_hash) out[c] = es_ratio((uint32_t)c, dn, rcp);
}
// PACKED (counts <= 511, i.e. n_hash <= 511): the LDS row holds the low byte of every count + one bit plane for bit 8 -- 9/16 of the bytes (54.6 KB
// instead of 93.9 KB at U = 44 931), which lets TWO rings of the persistent compare kernel stay resident beside it in the pipelined form
template <bool PACKED, int NQ>                // NQ: 16-byte units of a table row per thread, ceil(U / 8192) rounded up to 2, 4, 6 or 8
__global__ __launch_bounds__(ES_THREADS, 8)   // <= 64 VGPRs: the 4 waves per SIMD of one workgroup leave 256 VGPRs = two K2 waves
void k_expand_stream(const uint16_t *__restrict__ D, int64_t ld_d, const int32_t *__restrict__ uidx,
                                                              const uint32_t *__restrict__ cstart, const int32_t *__restrict__ cpos,
                                                              const int2 *__restrict__ items, const uint32_t *__restrict__ istart, int row_begin,
                                                              int row_end, int n, int n_hash, double *__restrict__ out, int64_t ld,
                                                              uint32_t *__restrict__ ticket) {
  extern __shared__ __attribute__((aligned(16))) unsigned char es_lds[];   // the table row
#ifdef ES_PRIO
  __builtin_amdgcn_s_setprio(ES_PRIO);   // (experiment: issue priority of the storing waves beside the compare's, which run their loop at priority 2)
#endif
  uint16_t *row = reinterpret_cast<uint16_t *>(es_lds);
  unsigned char *row_lo = es_lds, *row_hi = row_lo + ld_d;           // (PACKED) ld_d low bytes, then ld_d / 8 bytes of bit 8
  const int tid = threadIdx.x;
  const double dn = (double)n_hash, rcp = 1.0 / dn;                  // src/minHash.cpp:174 (es_ratio)
  auto count_of = [&](int c) -> uint32_t {
    if (!PACKED) return row[c];
    return (uint32_t)row_lo[c] | ((((uint32_t)row_hi[c >> 3] >> (c & 7)) & 1u) << 8);
  };
  const int n_items = (int)istart[row_end];                          // the items of the table rows [row_begin, row_end)
  const int units = (int)(ld_d >> 3);
  uint4 pre[NQ];
#define ES_FETCH(r_)                                                                                    \
  {                                                                                                     \
    const uint4 *src_ = reinterpret_cast<const uint4 *>(D + (int64_t)(r_) * ld_d);                      \
    _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                    \
      const int u = tid + q * ES_THREADS;                                                               \
      pre[q] = u < units ? src_[u] : make_uint4(0, 0, 0, 0);                                            \
    }                                                                                                   \
  }
  // a workgroup's first item is its block index, the following ones come from a ticket counter: items carry 1 ... ES_COPIES output rows, a
  // static deal leaves the slowest workgroup ~8 % behind the mean
  __shared__ int s_next;
  const int base = (int)istart[row_begin];
  int k = base + (int)blockIdx.x;
  int2 it = k < n_items ? items[k] : make_int2(0, 0);
  if (k < n_items) ES_FETCH(it.x)
  const int2 *u2 = reinterpret_cast<const int2 *>(uidx);
  const int j2_end = n >> 1;
  while (k < n_items) {
    __syncthreads();                                                 // the previous item's reads have left the LDS row
    if (tid == 0) s_next = base + (int)gridDim.x + (int)atomicAdd(ticket, 1u);
    uint4 *dst = reinterpret_cast<uint4 *>(row);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int u = tid + q * ES_THREADS;
      if (u >= units) continue;
      if (!PACKED) { dst[u] = pre[q]; continue; }
      const uint4 v = pre[q];                                        // counts 8 u ... 8 u + 7, two per word
      uint2 lo;
      lo.x = __builtin_amdgcn_perm(v.y, v.x, 0x06040200u);           // bytes 0 and 2 of x, then of y
      lo.y = __builtin_amdgcn_perm(v.w, v.z, 0x06040200u);
      *reinterpret_cast<uint2 *>(row_lo + 8 * u) = lo;
      row_hi[u] = (unsigned char)(((v.x >> 8) & 1u) | ((v.x >> 23) & 2u) | ((v.y >> 6) & 4u) | ((v.y >> 21) & 8u) | ((v.z >> 4) & 16u) |
                                  ((v.z >> 19) & 32u) | ((v.w >> 2) & 64u) | ((v.w >> 17) & 128u));
    }
    __syncthreads();
    // this item's output rows (wave-uniform), then the next item's table row on its way while this one is expanded
    const uint32_t c0 = cstart[it.x] + (uint32_t)it.y, c_end = cstart[it.x + 1];
    const int ncop = (int)min((uint32_t)ES_COPIES, c_end - c0);
    double *orow[ES_COPIES];
#pragma unroll
    for (int q = 0; q < ES_COPIES; ++q) orow[q] = out + (int64_t)cpos[c0 + (uint32_t)min(q, ncop - 1)] * ld;
    const int kn = __builtin_amdgcn_readfirstlane(s_next);          // wave-uniform: the item, its output rows and the next table row's address stay scalar
    int2 itn = make_int2(0, 0);
    if (kn < n_items) { itn = items[kn]; ES_FETCH(itn.x) }
#pragma unroll 4
    for (int j2 = tid; j2 < j2_end; j2 += ES_THREADS) {
      const int2 c = u2[j2];
      const double v0 = es_ratio(count_of(c.x), dn, rcp), v1 = es_ratio(count_of(c.y), dn, rcp);
      nt_store2(orow[0] + 2 * j2, v0, v1);
      if (ncop > 1) nt_store2(orow[1] + 2 * j2, v0, v1);
      if (ncop > 2) nt_store2(orow[2] + 2 * j2, v0, v1);
      if (ncop > 3) nt_store2(orow[3] + 2 * j2, v0, v1);
    }
    if ((n & 1) && tid == 0) {                                       // odd n: the last column
      const double v = es_ratio(count_of(uidx[n - 1]), dn, rcp);
      for (int q = 0; q < ncop; ++q) orow[q][n - 1] = v;
    }
    it = itn;
    k = kn;
  }
#undef ES_FETCH
}

bool expand_stream_ok(int64_t n, int64_t U, int n_hash, const void *d_out, int64_t ld) {
  return n >= 2 && n <= 0x7fffffffLL && U >= 1 && U <= 65536 && n_hash >= 1 && n_hash <= 2047 && (ld & 1) == 0 &&
         (reinterpret_cast<uintptr_t>(d_out) & 15) == 0;
}
extern "C" int da_debug_ratio_check(int n_hash, double *d_out, void *stream) {   // tests: es_ratio(c) for c = 0 .. n_hash into d_out
  if (n_hash < 1 || !d_out) return fail(DA_ERR_BAD_ARG, "ratio check: bad arguments");
  hipLaunchKernelGGL(k_es_ratio_check, dim3((unsigned)(n_hash / 256 + 1)), dim3(256), 0, static_cast<hipStream_t>(stream), n_hash, d_out);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}
bool expand_stream_packed(int n_hash) { return n_hash <= 511; }   // counts fit 9 bits: byte + bit plane in LDS
size_t expand_stream_scratch_bytes(int64_t n, int64_t U) {
  // cnt[U] + cursor[U] (zeroed together), cstart[U + 1], istart[U + 1], cpos[n], items[U + n / ES_COPIES + 1]
  return ((size_t)(4 * U + 8 + ES_TICKETS) * 4 + (size_t)n * 4 + (size_t)(U + n / ES_COPIES + 2) * 8 + 1024);
}
struct EsLists { uint32_t *cnt, *cursor, *ticket, *cstart, *istart; int32_t *cpos; int2 *items; };
static EsLists es_layout(void *d_scratch, int64_t n, int64_t U) {
  uint32_t *w = static_cast<uint32_t *>(d_scratch);
  EsLists L;
  L.cnt = w; L.cursor = w + U; L.ticket = w + 2 * U; L.cstart = L.ticket + ES_TICKETS; L.istart = L.cstart + (U + 1);   // cnt, cursor, ticket: zeroed together
  L.cpos = reinterpret_cast<int32_t *>(L.istart + (U + 1));
  L.items = reinterpret_cast<int2 *>((reinterpret_cast<uintptr_t>(L.cpos + n) + 15) & ~(uintptr_t)15);
  return L;
}
// the copy lists of the row expansion (positions of every unique id's copies, work items), stream-ordered
int launch_expand_stream_lists(const int32_t *d_uidx, int64_t n, int64_t U, void *d_scratch, hipStream_t stream) {
  const EsLists L = es_layout(d_scratch, n, U);
  DA_HIP_TRY(hipMemsetAsync(L.cnt, 0, ((size_t)U * 2 + ES_TICKETS) * 4, stream));
  const unsigned nb = (unsigned)ceil_div(std::max(n, U), 256);
  hipLaunchKernelGGL(k_es_count, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, stream, d_uidx, (int)n, L.cnt);
  hipLaunchKernelGGL(k_es_scan, dim3(1), dim3(1024), 0, stream, L.cnt, (int)U, L.cstart, L.istart);
  hipLaunchKernelGGL(k_es_fill, dim3(nb), dim3(256), 0, stream, d_uidx, (int)n, (int)U, L.cstart, L.istart, L.cursor, L.cpos, L.items);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}
// k_expand_stream on the table rows [row_begin, row_end) (lists from launch_expand_stream_lists on the same scratch)
int launch_expand_stream_rows(const uint16_t *d_D, int64_t ld_d, const int32_t *d_uidx, int64_t n, int64_t U, int n_hash, double *d_out, int64_t ld,
                              void *d_scratch, int64_t row_begin, int64_t row_end, hipStream_t stream, int launch_no) {
  if (!expand_stream_ok(n, U, n_hash, d_out, ld) || (ld_d & 7) || ld_d > 65536 || (reinterpret_cast<uintptr_t>(d_D) & 15) || launch_no < 0 ||
      launch_no >= ES_TICKETS)
    return fail(DA_ERR_UNSUPPORTED, "row expansion: shape not covered");
  if (row_end > U) row_end = U;
  if (row_begin >= row_end) return DA_OK;
  const EsLists L = es_layout(d_scratch, n, U);
  static std::atomic<int> es_cus;
  static std::atomic<uint64_t> es_attr_done;
  int dev = 0;
  DA_HIP_TRY(hipGetDevice(&dev));
  if (!((es_attr_done.load() >> (dev & 63)) & 1u)) {
    hipDeviceProp_t prop;
    DA_HIP_TRY(hipGetDeviceProperties(&prop, dev));
#define DA_ES_ATTR(P, Q) DA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_expand_stream<P, Q>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 * 2))
    DA_ES_ATTR(false, 2); DA_ES_ATTR(false, 4); DA_ES_ATTR(false, 6); DA_ES_ATTR(false, 8);
    DA_ES_ATTR(true, 2); DA_ES_ATTR(true, 4); DA_ES_ATTR(true, 6); DA_ES_ATTR(true, 8);
#undef DA_ES_ATTR
    es_cus.store(prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
    es_attr_done.fetch_or(1ull << (dev & 63));
  }
  const bool packed = expand_stream_packed(n_hash);
  size_t lds = packed ? (size_t)ld_d + (size_t)ld_d / 8 : (size_t)ld_d * 2;
  // never two of these workgroups on one CU (the pipelined form's consecutive launches would otherwise fill the VGPR file with 8 waves per SIMD
  // and lock the compare kernel out until the older launch has drained): ask for more than half of the CU's 160 KB
  lds = std::max<size_t>(lds, 82 * 1024);
  // one resident workgroup (16 waves) per CU: every workgroup of the grid must be resident from the start (first items are dealt by block index)
  int64_t grid = (int64_t)es_cus.load();
  grid = std::min<int64_t>(grid, (row_end - row_begin) + n / ES_COPIES + 1);
#define DA_ES(P, Q) hipLaunchKernelGGL((k_expand_stream<P, Q>), dim3((unsigned)grid), dim3(ES_THREADS), lds, stream, d_D, ld_d, d_uidx, L.cstart, L.cpos, L.items, \
                                       L.istart, (int)row_begin, (int)row_end, (int)n, n_hash, d_out, ld, L.ticket + launch_no)
  const int64_t nq = ceil_div(ld_d >> 3, ES_THREADS);                 // 16-byte units of a row per thread
  if (packed) { if (nq <= 2) DA_ES(true, 2); else if (nq <= 4) DA_ES(true, 4); else if (nq <= 6) DA_ES(true, 6); else DA_ES(true, 8); }
  else { if (nq <= 2) DA_ES(false, 2); else if (nq <= 4) DA_ES(false, 4); else if (nq <= 6) DA_ES(false, 6); else DA_ES(false, 8); }
#undef DA_ES
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}
int launch_expand_stream(const uint16_t *d_D, int64_t ld_d, const int32_t *d_uidx, int64_t n, int64_t U, int n_hash, double *d_out, int64_t ld,
                         void *d_scratch, hipStream_t stream, hipEvent_t after_lists) {
  if (!expand_stream_ok(n, U, n_hash, d_out, ld)) return fail(DA_ERR_UNSUPPORTED, "row expansion: shape not covered");
  int rc = launch_expand_stream_lists(d_uidx, n, U, d_scratch, stream);
  if (rc != DA_OK) return rc;
  if (after_lists) DA_HIP_TRY(hipEventRecord(after_lists, stream));
  return launch_expand_stream_rows(d_D, ld_d, d_uidx, n, U, n_hash, d_out, ld, d_scratch, 0, U, stream, 0);
}

// device bytes of the column-gathered table the two-pass expansion wants (0: the shape is not covered, pass NULL)
size_t expand_rows_workspace_bytes(int64_t n, int64_t U, int kind, bool is_nw, int n_hash, int nw_max_len) {
  if (kind != DA_OUT_F64 || n < 256 || U > 65536 || U < 1) return 0;
  if (is_nw ? (nw_max_len < 1 || (int64_t)(nw_max_len + 1) * (2 * nw_max_len + 1) > 2048) : (n_hash < 1 || n_hash + 1 > 2048)) return 0;
  return (size_t)U * (size_t)(ceil_div(n, 8) * 8) * 2;
}

int launch_gather_columns(const uint16_t *d_D, int64_t ld_d, const int32_t *d_uidx, const int32_t *d_ufirst, int64_t n, int64_t U,
                          uint16_t *d_F, int64_t ld_f, bool from_first_tile, hipStream_t stream, int table_world, int64_t table_rows_local,
                          int64_t row_begin, int64_t row_end) {
  if (row_end < 0 || row_end > U) row_end = U;                         // the unique rows [row_begin, row_end): everything by default
  if (row_begin < 0) row_begin = 0;
  if (U >= 1 && row_begin >= row_end) return DA_OK;
  if (U < 1 || U > 65536 || (ld_d & 7) || ld_d > 65536 || (reinterpret_cast<uintptr_t>(d_D) & 15) || (ld_f & 1))
    return fail(DA_ERR_UNSUPPORTED, "column gather: at most 65536 unique strings, 16-byte aligned table rows");
  const TableRows trows{table_world, table_rows_local};
  const size_t row_bytes = (size_t)ld_d * 2;
  static std::atomic<int> gc_cus;
  static std::atomic<uint64_t> gc_attr_done;                           // per device: the kernel may use a 128 KiB dynamic LDS row
  int dev = 0;
  DA_HIP_TRY(hipGetDevice(&dev));
  if (!((gc_attr_done.load() >> (dev & 63)) & 1u)) {
    hipDeviceProp_t prop;
    DA_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    DA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gather_columns), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 * 2));
    gc_cus.store(prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
    gc_attr_done.fetch_or(1ull << (dev & 63));
  }
  const int gc_wg = row_bytes <= 40 * 1024 ? 2 : 1;                    // resident workgroups per CU the LDS row allows (of 2 x 16 waves)
  const int64_t gc_grid = std::min<int64_t>(row_end - row_begin, (int64_t)gc_cus.load() * gc_wg);
  hipLaunchKernelGGL(k_gather_columns, dim3((unsigned)gc_grid), dim3(GC_THREADS), row_bytes, stream, d_D, ld_d, d_uidx, d_ufirst, (int)n,
                     (int)row_end, d_F, ld_f, trows, from_first_tile ? 1 : 0, (int)row_begin);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

// k_expand_rows on the output tile-row bands [band_begin, band_end) (8 x 128 rows each, decode_tile's numbering of the n x n tiles);
// band_end < 0: all of them
int launch_expand_rows(const uint16_t *d_F, int64_t ld_f, const int32_t *d_uidx, int64_t n, bool is_nw, int n_hash, int nw_max_len,
                       double *d_out, int64_t ld, int64_t band_begin, int64_t band_end, hipStream_t stream) {
  const int T128 = (int)ceil_div(n, 128);
  if (band_end < 0) band_end = mh_sym_bands(n);
  const int64_t t0 = mh_sym_band_prefix(n, band_begin), t1 = mh_sym_band_prefix(n, band_end);
  if (t1 <= t0) return DA_OK;
  const int64_t px = ceil_div(t1 - t0, 8);
  const int stride = is_nw ? 2 * nw_max_len + 1 : 1;
  const int entries = is_nw ? (nw_max_len + 1) * stride : n_hash + 1;
  const size_t lds = 128 * ER_STRIDE + (size_t)entries * 8;
  if (is_nw) hipLaunchKernelGGL(k_expand_rows<true>, dim3((unsigned)(px * 8)), dim3(256), lds, stream, d_F, ld_f, d_uidx, (int)n, n_hash,
                                stride, entries, d_out, ld, T128, t1, px, t0);
  else hipLaunchKernelGGL(k_expand_rows<false>, dim3((unsigned)(px * 8)), dim3(256), lds, stream, d_F, ld_f, d_uidx, (int)n, n_hash,
                          stride, entries, d_out, ld, T128, t1, px, t0);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

int launch_expand_unique(const uint16_t *d_D, int64_t ld_d, const int32_t *d_uidx, int64_t n, int kind, bool is_nw, int n_hash,
                         void *d_out, int64_t ld, hipStream_t stream, int nw_max_len, uint16_t *d_F, const int32_t *d_ufirst, int64_t U,
                         hipEvent_t after_gather, hipEvent_t after_rows, int table_world, int64_t table_rows_local, bool only_leftover) {
  const TableRows trows{table_world, table_rows_local};
  if (n <= 0) return DA_OK;
  if (n > 0x7fffffffLL) return fail(DA_ERR_UNSUPPORTED, "expand: matrix too large");
  if (kind != DA_OUT_F64 && kind != DA_OUT_COMPACT) return fail(DA_ERR_BAD_ARG, "expand: bad output kind");
  const int TB = (int)ceil_div(n, 64);
  const int64_t tiles = (int64_t)TB * (TB + 1) / 2;
  if (tiles > 0x7fffffffLL) return fail(DA_ERR_UNSUPPORTED, "matrix too large for one launch");
  const int64_t per_xcd = ceil_div(tiles, 8);
  dim3 grid((unsigned)(per_xcd * 8));
  // float64 with a column-gathered table from the caller: interior off-diagonal 128 x 128 tiles by the two streaming passes,
  // the rest (diagonal, borders) by the 64 x 64 kernel below
  const bool fast = d_F != nullptr && expand_rows_workspace_bytes(n, U, kind, is_nw, n_hash, nw_max_len) != 0 && (ld & 1) == 0 &&
                    (reinterpret_cast<uintptr_t>(d_out) & 15) == 0;
  if (only_leftover && !fast) return fail(DA_ERR_BAD_ARG, "expand: the streaming passes do not cover this shape");
  if (fast && !only_leftover) {
    const int64_t ld_f = ceil_div(n, 8) * 8;
    int rc_g = launch_gather_columns(d_D, ld_d, d_uidx, d_ufirst, n, U, d_F, ld_f, false, stream, table_world, table_rows_local);
    if (rc_g != DA_OK) return rc_g;
    if (after_gather) DA_HIP_TRY(hipEventRecord(after_gather, stream));
    rc_g = launch_expand_rows(d_F, ld_f, d_uidx, n, is_nw, n_hash, nw_max_len, static_cast<double *>(d_out), ld, 0, -1, stream);
    if (rc_g != DA_OK) return rc_g;
  }
  else if (after_gather) DA_HIP_TRY(hipEventRecord(after_gather, stream));
  if (after_rows) DA_HIP_TRY(hipEventRecord(after_rows, stream));
  const int skip_fast = fast ? 1 : 0;
  if (fast) grid.x = leftover_blocks64(n);   // the 64 x 64 kernel then only visits the diagonal 128-tiles and the border column
#define DA_EXP(F, W, T) hipLaunchKernelGGL((k_expand_unique<F, W, T>), grid, dim3(256), 0, stream, d_D, ld_d, d_uidx, (int)n, n_hash, d_out, ld, TB, tiles, per_xcd, skip_fast, trows)
  if (kind == DA_OUT_F64) { if (is_nw) DA_EXP(true, true, 64); else DA_EXP(true, false, 64); }
  else { if (is_nw) DA_EXP(false, true, 64); else DA_EXP(false, false, 64); }
#undef DA_EXP
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

int launch_finalize_sharded(const uint16_t *d_g, int64_t ld_g, const ShardGeom &geom, bool is_nw, int n_hash,
                            double *d_out, int64_t ld, hipStream_t stream) {
  if (geom.n <= 0) return DA_OK;
  if (geom.n > 0x7fffffffLL || (geom.tile % 64) != 0) return fail(DA_ERR_UNSUPPORTED, "finalize: unsupported geometry");
  const int TB = (int)ceil_div(geom.n, 64);
  const int64_t tiles = (int64_t)TB * (TB + 1) / 2;
  if (tiles > 0x7fffffffLL) return fail(DA_ERR_UNSUPPORTED, "matrix too large for one launch");
  const int64_t per_xcd = ceil_div(tiles, 8);
  dim3 grid((unsigned)(per_xcd * 8));
  const bool fast = finalize_rows_ok(geom, d_g, ld_g, d_out, ld, false);
  if (fast) {
    const int T128 = (int)ceil_div(geom.n, 128);
    const int64_t t128 = (int64_t)T128 * (T128 + 1) / 2, px = ceil_div(t128, 8);
    const int entries = (!is_nw && n_hash + 1 <= 2048) ? n_hash + 1 : 0;
    const size_t lds = 128 * ER_STRIDE + (size_t)entries * 8;
    if (is_nw) hipLaunchKernelGGL((k_finalize_rows<true, false>), dim3((unsigned)(px * 8)), dim3(256), lds, stream, d_g, ld_g, geom, 0, n_hash, entries,
                                  d_out, ld, T128, t128, px);
    else hipLaunchKernelGGL((k_finalize_rows<false, false>), dim3((unsigned)(px * 8)), dim3(256), lds, stream, d_g, ld_g, geom, 0, n_hash, entries,
                            d_out, ld, T128, t128, px);
    grid.x = leftover_blocks64(geom.n);
  }
  if (is_nw)
    hipLaunchKernelGGL(k_finalize_sharded<true>, grid, dim3(256), 0, stream, d_g, ld_g, geom, n_hash, d_out, ld, TB, tiles, per_xcd, fast ? 1 : 0);
  else
    hipLaunchKernelGGL(k_finalize_sharded<false>, grid, dim3(256), 0, stream, d_g, ld_g, geom, n_hash, d_out, ld, TB, tiles, per_xcd, fast ? 1 : 0);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

// ---- b-bit packing of the MH shard blocks for the exchange -------------------------------------
// A match count needs value_bits = bits(n_hash) <= 16 bits (9 at n_hash = 500), the gather is what
// the multi-GPU MH step waits for, so a rank's uint16 block [rows][W] travels as a byte plane
// (low 8 bits) followed by value_bits - 8 bit planes ([rows][W/8] bytes each, bit c&7 of byte c>>3).
__global__ __launch_bounds__(256) void k_pack_shard(const uint16_t *__restrict__ local, int64_t ld, int64_t rows, int64_t W,
                                                    int nhi, uint8_t *__restrict__ packed) {
  const int64_t groups = W >> 3;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * groups) return;
  const int64_t row = idx / groups, g = idx - row * groups;
  const uint16_t *src = local + row * ld + 8 * g;
  uint32_t v[8];
  if (((ld & 7) == 0) && ((reinterpret_cast<uintptr_t>(local) & 15) == 0)) {
    const uint4 q = *reinterpret_cast<const uint4 *>(src);
    v[0] = q.x & 0xffffu; v[1] = q.x >> 16; v[2] = q.y & 0xffffu; v[3] = q.y >> 16;
    v[4] = q.z & 0xffffu; v[5] = q.z >> 16; v[6] = q.w & 0xffffu; v[7] = q.w >> 16;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[e];
  }
  uint2 lo;
  lo.x = (v[0] & 255u) | ((v[1] & 255u) << 8) | ((v[2] & 255u) << 16) | ((v[3] & 255u) << 24);
  lo.y = (v[4] & 255u) | ((v[5] & 255u) << 8) | ((v[6] & 255u) << 16) | ((v[7] & 255u) << 24);
  *reinterpret_cast<uint2 *>(packed + row * W + 8 * g) = lo;
  uint8_t *hi = packed + rows * W;
  for (int k = 0; k < nhi; ++k) {
    uint32_t b = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) b |= ((v[e] >> (8 + k)) & 1u) << e;
    hi[((int64_t)k * rows + row) * groups + g] = (uint8_t)b;
  }
}

// k_finalize_sharded reading packed blocks (MH only): G = world blocks of block_bytes each
template <bool U16 = false>
__global__ __launch_bounds__(256) void k_finalize_packed(const uint8_t *__restrict__ G, int64_t block_bytes, ShardGeom geom,
                                                         int nhi, int n_hash, void *__restrict__ out_v, int64_t ld, int TB,
                                                         int64_t ntiles, int64_t per_xcd, int skip_fast) {
  double *out = static_cast<double *>(out_v);
  constexpr int FT = 64, TABLE = 2048;
  __shared__ uint16_t t[FT][FT + 2];
  __shared__ double ratio[TABLE];
  const int n = (int)geom.n;
  const TileId tt = skip_fast ? leftover_tile64(blockIdx.x, n, TB) : decode_tile_xcd(blockIdx.x, per_xcd, ntiles, TB);
  if (!tt.valid) return;
  const int i0 = tt.ti * FT, j0 = tt.tj * FT;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  const bool use_table = n_hash < TABLE;
  if (use_table)
    for (int c = threadIdx.x; c <= n_hash; c += 256) ratio[c] = (double)c / (double)n_hash;   // src/minHash.cpp:174
  auto widen = [&](uint32_t v) -> double { return use_table ? ratio[v] : (double)v / (double)n_hash; };
  const int tile = geom.tile;
  const int tr = i0 / tile;
  const int q = tr / geom.world, owner = tr - q * geom.world;
  const bool front = q <= geom.Q - 1 - q;
  const int64_t lrow0 = (int64_t)(front ? q : geom.Q - 1 - q) * tile + (i0 - tr * tile);   // row inside the owner's block
  const int64_t coff = front ? -(int64_t)tr * tile : shard_back(geom.W, geom.n);
  const uint8_t *blk = G + (int64_t)owner * block_bytes;
  const int64_t W = geom.W, groups = W >> 3, rows = geom.rows;
  const int j = j0 + tx;
  const int64_t col = coff + j;
  for (int r = ty; r < FT; r += 4) {
    const int i = i0 + r;
    if (i < n && j < n && j >= i) {
      const int64_t row = lrow0 + r;
      uint32_t v = blk[row * W + col];
      for (int k = 0; k < nhi; ++k)
        v |= (uint32_t)((blk[rows * W + ((int64_t)k * rows + row) * groups + (col >> 3)] >> (col & 7)) & 1u) << (8 + k);
      t[r][tx] = (uint16_t)v;
    }
  }
  __syncthreads();
  if (U16) store_tile_u16<64>(t, static_cast<uint16_t *>(out_v), ld, n, i0, j0);
  else store_tile_f64<64>(t, widen, out, ld, n, i0, j0);
}

int64_t shard_packed_bytes(const ShardGeom &g, int value_bits) {
  const int nhi = value_bits > 8 ? value_bits - 8 : 0;
  return g.rows * g.W + (int64_t)nhi * g.rows * (g.W >> 3);
}

int launch_pack_shard(const uint16_t *d_local, int64_t ld, const ShardGeom &geom, int value_bits, uint8_t *d_packed,
                      hipStream_t stream) {
  if (geom.n <= 0) return DA_OK;
  const int nhi = value_bits > 8 ? value_bits - 8 : 0;
  const int64_t work = geom.rows * (geom.W >> 3);
  hipLaunchKernelGGL(k_pack_shard, dim3((unsigned)ceil_div(work, 256)), dim3(256), 0, stream, d_local, ld, geom.rows, geom.W,
                     nhi, d_packed);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

int launch_finalize_packed(const uint8_t *d_g, const ShardGeom &geom, int value_bits, int n_hash, double *d_out, int64_t ld,
                           hipStream_t stream) {
  if (geom.n <= 0) return DA_OK;
  if (geom.n > 0x7fffffffLL || (geom.tile % 64) != 0) return fail(DA_ERR_UNSUPPORTED, "finalize: unsupported geometry");
  const int TB = (int)ceil_div(geom.n, 64);
  const int64_t tiles = (int64_t)TB * (TB + 1) / 2;
  if (tiles > 0x7fffffffLL) return fail(DA_ERR_UNSUPPORTED, "matrix too large for one launch");
  const int64_t per_xcd = ceil_div(tiles, 8);
  const int nhi = value_bits > 8 ? value_bits - 8 : 0;
  unsigned blocks = (unsigned)(per_xcd * 8);
  const bool fast = finalize_rows_ok(geom, d_g, 0, d_out, ld, true);
  if (fast) {
    const int T128 = (int)ceil_div(geom.n, 128);
    const int64_t t128 = (int64_t)T128 * (T128 + 1) / 2, px = ceil_div(t128, 8);
    const int entries = n_hash + 1 <= 2048 ? n_hash + 1 : 0;
    hipLaunchKernelGGL((k_finalize_rows<false, true>), dim3((unsigned)(px * 8)), dim3(256), 128 * ER_STRIDE + (size_t)entries * 8, stream, d_g,
                       shard_packed_bytes(geom, value_bits), geom, nhi, n_hash, entries, d_out, ld, T128, t128, px);
    blocks = leftover_blocks64(geom.n);
  }
  hipLaunchKernelGGL(k_finalize_packed<false>, dim3(blocks), dim3(256), 0, stream, d_g, shard_packed_bytes(geom, value_bits),
                     geom, nhi, n_hash, d_out, ld, TB, tiles, per_xcd, fast ? 1 : 0);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

// gathered MinHash shards (uint16 folded blocks: value_bits = 0, d_g leading dimension ld_g; packed blocks: value_bits > 0) ->
// symmetric uint16 count table of geom.n rows (the unique strings of the sharded duplicate route)
int launch_shards_to_table(const void *d_g, int64_t ld_g, const ShardGeom &geom, int value_bits, uint16_t *d_table, int64_t ld,
                           hipStream_t stream) {
  if (geom.n <= 0) return DA_OK;
  if (geom.n > 0x7fffffffLL || (geom.tile % 64) != 0) return fail(DA_ERR_UNSUPPORTED, "shards -> table: unsupported geometry");
  const int TB = (int)ceil_div(geom.n, 64);
  const int64_t tiles = (int64_t)TB * (TB + 1) / 2;
  if (tiles > 0x7fffffffLL) return fail(DA_ERR_UNSUPPORTED, "matrix too large for one launch");
  const int64_t per_xcd = ceil_div(tiles, 8);
  const dim3 grid((unsigned)(per_xcd * 8));
  if (value_bits > 0)
    hipLaunchKernelGGL(k_finalize_packed<true>, grid, dim3(256), 0, stream, static_cast<const uint8_t *>(d_g), shard_packed_bytes(geom, value_bits),
                       geom, value_bits > 8 ? value_bits - 8 : 0, 1, d_table, ld, TB, tiles, per_xcd, 0);
  else
    hipLaunchKernelGGL((k_finalize_sharded<false, true>), grid, dim3(256), 0, stream, static_cast<const uint16_t *>(d_g), ld_g, geom, 1, d_table, ld,
                       TB, tiles, per_xcd, 0);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

// ---- SPARSE route of the symmetric float64 compare (round 3) --------------------------------------------------------------
// For inputs whose signatures rarely agree (uniform random peptides at N = 100k: 1.4*10^8 matching (pair, hash function)
// incidences against 2.5*10^12 compared ones) the bit-sliced compare spends 31 ms finding that almost every count is zero.  The
// dictionary codes of K1b already say which sequences share a value in a column; so, exactly:
//   k_sp_count   per column: multiplicity of every repeated value (LDS histogram) -> E = sum m (m - 1) / 2 and the largest m; the host
//                takes the route when E and m are small (da_dev_similarity_mh; DYNAALIGN_MH_SPARSE_MAX_PAIRS)
//   k_sp_link    per (column, sequence): a linked list per repeated value (atomicExch on the value's head)
//   k_sp_walk    per (column, sequence): walk the list behind the sequence = the partners inserted before it -> every matching
//                incidence (i < j, column) once; pass 0 counts them per 128 x 128 output tile, pass 1 (after a scan) drops them into
//                the tile's bucket as 14-bit local coordinates
//   k_sp_tiles   per output tile on or above the diagonal: the incidences are added into a 128 x 128 uint16 image in LDS (integer
//                atomics), then the tile and its mirror image are written as float64 through the count -> double table -- the same
//                streaming-store epilogue as k_expand_rows (er_store_tile); diagonal / border tiles element by element.
// Every element of the result is written exactly once; counts are integers until the table lookup: bit-identical to the dense kernels.
constexpr int SP_MAX_IDS = 32768;            // LDS histogram of a column's repeated values (uint32 each: 128 KiB)
__global__ __launch_bounds__(1024) void k_sp_count(const uint16_t *__restrict__ idsT, int64_t ld_ids, int n, int max_ids,
                                                   unsigned long long *__restrict__ stats) {
  extern __shared__ uint32_t sp_hist[];
  const uint16_t *ids = idsT + (int64_t)blockIdx.x * ld_ids;
  for (int c = threadIdx.x; c < max_ids; c += 1024) sp_hist[c] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 1024) {
    const uint32_t c = ids[i];
    if (c != 0xFFFFu) atomicAdd(&sp_hist[c], 1u);
  }
  __syncthreads();
  unsigned long long e = 0, mx = 0;
  for (int c = threadIdx.x; c < max_ids; c += 1024) {
    const unsigned long long m = sp_hist[c];
    e += m * (m - (m ? 1 : 0)) / 2;
    mx = m > mx ? m : mx;
  }
  for (int o = 32; o > 0; o >>= 1) {
    e += __shfl_down(e, o);
    const unsigned long long other = __shfl_down(mx, o);
    mx = other > mx ? other : mx;
  }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&stats[0], e); atomicMax(&stats[1], mx); }
}
// Per column: the members of every repeated value, class after class (a counting sort by code in LDS -- no global atomics):
//   cstart[h][c] .. cstart[h][c + 1]  = where class c's members sit in member[h][.]
__global__ __launch_bounds__(1024) void k_sp_classes(const uint16_t *__restrict__ idsT, int64_t ld_ids, int n, int max_ids, int cs_ld,
                                                     uint32_t *__restrict__ cstart, uint32_t *__restrict__ member) {
  extern __shared__ uint32_t sp_hist[];        // max_ids counters, then 1024 partial sums
  uint32_t *part = sp_hist + max_ids;
  const uint16_t *ids = idsT + (int64_t)blockIdx.x * ld_ids;
  uint32_t *cs = cstart + (int64_t)blockIdx.x * cs_ld, *mem = member + (int64_t)blockIdx.x * ld_ids;
  for (int c = threadIdx.x; c < max_ids; c += 1024) sp_hist[c] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 1024) {
    const uint32_t c = ids[i];
    if (c != 0xFFFFu) atomicAdd(&sp_hist[c], 1u);
  }
  __syncthreads();
  const int per = (max_ids + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < max_ids ? lo + per : max_ids;
  uint32_t sum = 0;
  for (int c = lo; c < hi; ++c) sum += sp_hist[c];
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x < 64) {                      // exclusive scan of the 1024 partial sums by one wave (16 each)
    uint32_t v[16], tot = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { v[q] = part[threadIdx.x * 16 + q]; tot += v[q]; }
    uint32_t incl = tot;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if ((int)threadIdx.x >= o) incl += up; }
    uint32_t run = incl - tot;
#pragma unroll
    for (int q = 0; q < 16; ++q) { part[threadIdx.x * 16 + q] = run; run += v[q]; }
  }
  __syncthreads();
  uint32_t run = part[threadIdx.x];
  for (int c = lo; c < hi; ++c) { const uint32_t v = sp_hist[c]; cs[c] = run; sp_hist[c] = run; run += v; }
  if (hi == max_ids && lo < hi) cs[max_ids] = run;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 1024) {
    const uint32_t c = ids[i];
    if (c != 0xFFFFu) mem[atomicAdd(&sp_hist[c], 1u)] = (uint32_t)i;
  }
}
__device__ __forceinline__ int64_t sp_tile_index(int ti, int tj, int T) { return (int64_t)ti * T - (int64_t)ti * (ti - 1) / 2 + (tj - ti); }
// Per (column, sequence i): the class partners j > i = the matching incidences (i, j, column), each once, emitted from the side of the
// SMALLER index -- so a block of 256 consecutive i feeds exactly two 128-row bands of the output and needs one global atomic per band
// to reserve its space.  PASS 0 sizes the bands, PASS 1 (after a scan of the 782 band sizes) writes (i & 127) << 17 | j.
template <int PASS>
__global__ __launch_bounds__(256) void k_sp_emit(const uint16_t *__restrict__ idsT, int64_t ld_ids, int n, int cs_ld, const uint32_t *__restrict__ cstart,
                                                 const uint32_t *__restrict__ member, uint32_t *__restrict__ band_cnt,
                                                 const uint32_t *__restrict__ band_start, uint32_t *__restrict__ entries32,
                                                 uint16_t *__restrict__ saved_cnt) {
  __shared__ uint32_t blk[4];
  const int i = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y, bl = threadIdx.x >> 7;
  if (threadIdx.x < 4) blk[threadIdx.x] = 0;
  __syncthreads();
  uint32_t lo = 0, hi = 0, cnt = 0;
  const uint32_t *mem = member + (int64_t)h * ld_ids;
  if (i < n) {
    const uint32_t c = idsT[(int64_t)h * ld_ids + i];
    if (c != 0xFFFFu) { lo = cstart[(int64_t)h * cs_ld + c]; hi = cstart[(int64_t)h * cs_ld + c + 1]; }
    if (PASS == 0) {
      for (uint32_t k = lo; k < hi; ++k) cnt += mem[k] > (uint32_t)i;
      saved_cnt[(int64_t)h * ld_ids + i] = (uint16_t)cnt;          // (a class has at most 4096 members: the route's admission rule)
    } else {
      cnt = saved_cnt[(int64_t)h * ld_ids + i];
    }
  }
  const uint32_t off = cnt ? atomicAdd(&blk[bl], cnt) : 0u;
  __syncthreads();
  const int band = 2 * blockIdx.x + bl;
  if (PASS == 0) {
    if ((threadIdx.x & 127) == 0 && blk[bl]) atomicAdd(&band_cnt[band], blk[bl]);
    return;
  }
  if ((threadIdx.x & 127) == 0 && blk[bl]) blk[2 + bl] = band_start[band] + atomicAdd(&band_cnt[band], blk[bl]);   // band_cnt: cursors (zeroed again)
  __syncthreads();
  if (!cnt) return;
  uint32_t w = blk[2 + bl] + off;
  for (uint32_t k = lo; k < hi; ++k) {
    const uint32_t j = mem[k];
    if (j > (uint32_t)i) entries32[w++] = ((uint32_t)(i & 127) << 17) | j;
  }
}
// exclusive scan of `count` values by one workgroup: start[t] (and cnt[t] = 0 for its second life as a cursor), start[count] = total
__global__ __launch_bounds__(1024) void k_sp_scan(uint32_t *__restrict__ cnt, uint32_t *__restrict__ start, int64_t count) {
  __shared__ uint32_t part[1024];
  const int64_t per = (count + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < count ? lo + per : count;
  uint32_t s = 0;
  for (int64_t t = lo; t < hi; ++t) s += cnt[t];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { uint32_t run = 0; for (int k = 0; k < 1024; ++k) { const uint32_t v = part[k]; part[k] = run; run += v; } }
  __syncthreads();
  uint32_t run = part[threadIdx.x];
  for (int64_t t = lo; t < hi; ++t) { const uint32_t v = cnt[t]; start[t] = run; cnt[t] = 0; run += v; }
  if (threadIdx.x == 1023) start[count] = run;
}
// Per band (128 output rows): its incidences sorted by tile column in LDS (histogram, scan, cursors: LDS atomics only) -> the tiles' start
// offsets and the final 14-bit entries (row & 127) << 7 | (column & 127)
__global__ __launch_bounds__(1024) void k_sp_band(const uint32_t *__restrict__ band_start, const uint32_t *__restrict__ entries32, int T,
                                                  uint32_t *__restrict__ start, uint16_t *__restrict__ entries) {
  __shared__ uint32_t hist[1024];
  const int ti = blockIdx.x;
  const uint32_t lo = band_start[ti], hi = band_start[ti + 1];
  hist[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t e = lo + threadIdx.x; e < hi; e += 4096) {     // four loads in flight per thread: the loop is latency-bound otherwise
    uint32_t v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = e + q * 1024 < hi ? entries32[e + q * 1024] : 0xffffffffu;
#pragma unroll
    for (int q = 0; q < 4; ++q) if (v[q] != 0xffffffffu) atomicAdd(&hist[(v[q] & 0x1ffffu) >> 7], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 64) {                      // exclusive scan of the 1024 counters by one wave (16 each)
    uint32_t v[16], tot = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { v[q] = hist[threadIdx.x * 16 + q]; tot += v[q]; }
    uint32_t incl = tot;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if ((int)threadIdx.x >= o) incl += up; }
    uint32_t run = incl - tot;
#pragma unroll
    for (int q = 0; q < 16; ++q) { hist[threadIdx.x * 16 + q] = run; run += v[q]; }
  }
  __syncthreads();
  const int tj = threadIdx.x;
  if (tj >= ti && tj < T) start[sp_tile_index(ti, tj, T)] = lo + hist[tj];
  if (ti == T - 1 && threadIdx.x == 0) start[sp_tile_index(T - 1, T - 1, T) + 1] = hi;
  __syncthreads();
  for (uint32_t e = lo + threadIdx.x; e < hi; e += 4096) {
    uint32_t v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = e + q * 1024 < hi ? entries32[e + q * 1024] : 0xffffffffu;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (v[q] != 0xffffffffu) {
        const uint32_t j = v[q] & 0x1ffffu;
        entries[lo + atomicAdd(&hist[j >> 7], 1u)] = (uint16_t)(((v[q] >> 17) << 7) | (j & 127u));
      }
  }
}
__global__ __launch_bounds__(256, 4) void k_sp_tiles(const uint32_t *__restrict__ start, const uint16_t *__restrict__ entries, int n, int n_hash,
                                                     int tab_entries, double *__restrict__ out, int64_t ld, int T, int64_t ntiles, int64_t per_xcd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char er_lds[];   // 128 x ER_STRIDE uint16 image, then the count -> double table
  double *tab = reinterpret_cast<double *>(er_lds + 128 * ER_STRIDE);
  const int64_t L = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (L >= ntiles) return;
  const TileId tt = decode_tile(L, T, T, true);
  if (!tt.valid) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  uint32_t *img = reinterpret_cast<uint32_t *>(er_lds);
  for (int w = tid; w < 128 * ER_STRIDE / 4; w += 256) img[w] = 0u;
  for (int e = tid; e < tab_entries; e += 256) tab[e] = (double)e / (double)n_hash;            // src/minHash.cpp:174
  __syncthreads();
  const int64_t t = sp_tile_index(tt.ti, tt.tj, T);
  const uint32_t e0 = start[t], e1 = start[t + 1];
  for (uint32_t e = e0 + tid; e < e1; e += 256) {
    const uint32_t v = entries[e], r = v >> 7, c = v & 127u;
    atomicAdd(&img[(r * ER_STRIDE + (c >> 1) * 4) / 4], 1u << (16 * (c & 1u)));                // two uint16 counts per word: no carry below 65536
  }
  __syncthreads();
  const int64_t I0 = (int64_t)tt.ti * 128, J0 = (int64_t)tt.tj * 128;
  if (expand_fast_takes(tt.ti, tt.tj, n, ld, out)) {
    const int tx = ((wave & 1) << 3) + (lane & 7), ty = ((wave >> 1) << 3) + (lane >> 3);
    er_store_tile(er_lds, [&](uint32_t x) -> double { return tab[x]; }, out, ld, I0, J0, tx, ty);
    return;
  }
  // diagonal / border tiles: element by element (the image holds i < j only: the diagonal tile's lower half comes from its upper)
  const uint16_t *img16 = reinterpret_cast<const uint16_t *>(er_lds);
  for (int q = tid; q < 128 * 128; q += 256) {
    const int r = q >> 7, c = q & 127;
    const int64_t i = I0 + r, j = J0 + c;
    if (i >= n || j >= n) continue;
    if (tt.ti == tt.tj) {
      const uint32_t v = r == c ? (uint32_t)n_hash : img16[(r < c ? r : c) * (ER_STRIDE / 2) + (r < c ? c : r)];   // diagonal: src/minHash.cpp:161
      out[i * ld + j] = tab[v];
    } else {
      const double v = tab[img16[r * (ER_STRIDE / 2) + c]];
      out[i * ld + j] = v;
      out[j * ld + i] = v;
    }
  }
}

size_t mh_sparse_pairs_limit() {
  return (size_t)config().mh_sparse_max_pairs;   // DYNAALIGN_MH_SPARSE_MAX_PAIRS (400 000 000; at most 0xfffffff0: 32-bit entry offsets)
}
// stats[0] = matching (pair, hash function) incidences, stats[1] = largest multiplicity of a value in a column; both zeroed here
int launch_mh_sparse_count(const uint16_t *d_idsT, int64_t ld_ids, int64_t n, int n_hash, int max_ids, unsigned long long *d_stats, hipStream_t stream) {
  if (max_ids < 1 || max_ids > SP_MAX_IDS || n > 0x7fffffffLL) return fail(DA_ERR_UNSUPPORTED, "sparse count: too many repeated values per column");
  static std::atomic<uint64_t> attr_done;
  int dev = 0;
  DA_HIP_TRY(hipGetDevice(&dev));
  if (!((attr_done.load() >> (dev & 63)) & 1u)) {
    DA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sp_count), hipFuncAttributeMaxDynamicSharedMemorySize, SP_MAX_IDS * 4));
    attr_done.fetch_or(1ull << (dev & 63));
  }
  DA_HIP_TRY(hipMemsetAsync(d_stats, 0, 16, stream));
  hipLaunchKernelGGL(k_sp_count, dim3((unsigned)n_hash), dim3(1024), (size_t)max_ids * 4, stream, d_idsT, ld_ids, (int)n, max_ids, d_stats);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}
// scratch the sparse route needs besides the two entry lists (uint32 words): class starts [n_hash][cs_ld], members [n_hash][ld_ids], band
// counters + starts, tile starts
size_t mh_sparse_scratch_words(int64_t n, int n_hash, int max_ids, int64_t ld_ids) {
  const int64_t T = ceil_div(n, 128), ntiles = T * (T + 1) / 2;
  return (size_t)n_hash * (size_t)((max_ids + 1 + 63) / 64 * 64) + (size_t)n_hash * (size_t)ld_ids + 2 * (size_t)(T + 64) + (size_t)(ntiles + 64) +
         (size_t)n_hash * (size_t)ld_ids / 2 + 64;      // ... + the per-(column, sequence) partner counts pass 0 leaves for pass 1 (uint16)
}
// carve-up of the route's scratch (mh_sparse_scratch_words)
struct SparseScratch { uint32_t *cstart, *member, *band_cnt, *band_start, *start; uint16_t *saved; int cs_ld, T; int64_t ntiles; };
static SparseScratch sparse_scratch(uint32_t *d_scratch, int64_t n, int n_hash, int max_ids, int64_t ld_ids) {
  SparseScratch w;
  w.T = (int)ceil_div(n, 128);
  w.ntiles = (int64_t)w.T * (w.T + 1) / 2;
  w.cs_ld = (max_ids + 1 + 63) / 64 * 64;
  w.cstart = d_scratch;
  w.member = w.cstart + (size_t)n_hash * w.cs_ld;
  w.band_cnt = w.member + (size_t)n_hash * ld_ids;
  w.band_start = w.band_cnt + w.T + 64;
  w.start = w.band_start + w.T + 64;
  w.saved = reinterpret_cast<uint16_t *>(w.start + w.ntiles + 64);
  return w;
}
// the LIST phase: d_entries32: `pairs` uint32, d_entries: `pairs` uint16; afterwards start[sp_tile_index(ti, tj)] .. of the scratch delimit tile (ti, tj)'s entries
int launch_mh_sparse_lists(const uint16_t *d_idsT, int64_t ld_ids, int64_t n, int n_hash, int max_ids, uint64_t pairs, uint32_t *d_scratch,
                           uint32_t *d_entries32, uint16_t *d_entries, hipStream_t stream) {
  if (n > 131072 || n_hash + 1 > 2048 || pairs > 0xfffffff0ull || max_ids < 1 || max_ids > SP_MAX_IDS)
    return fail(DA_ERR_UNSUPPORTED, "sparse route: shape not covered");
  const SparseScratch w = sparse_scratch(d_scratch, n, n_hash, max_ids, ld_ids);
  const int T = w.T, cs_ld = w.cs_ld;
  uint32_t *cstart = w.cstart, *member = w.member, *band_cnt = w.band_cnt, *band_start = w.band_start, *start = w.start;
  uint16_t *saved = w.saved;
  static std::atomic<uint64_t> attr_done;
  int dev = 0;
  DA_HIP_TRY(hipGetDevice(&dev));
  if (!((attr_done.load() >> (dev & 63)) & 1u)) {
    DA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sp_classes), hipFuncAttributeMaxDynamicSharedMemorySize, (SP_MAX_IDS + 1024) * 4));
    attr_done.fetch_or(1ull << (dev & 63));
  }
  DA_HIP_TRY(hipMemsetAsync(band_cnt, 0, (size_t)(T + 64) * 4, stream));
  hipLaunchKernelGGL(k_sp_classes, dim3((unsigned)n_hash), dim3(1024), (size_t)(max_ids + 1024) * 4, stream, d_idsT, ld_ids, (int)n, max_ids, cs_ld, cstart, member);
  const dim3 grid((unsigned)ceil_div(n, 256), (unsigned)n_hash);
  hipLaunchKernelGGL(k_sp_emit<0>, grid, dim3(256), 0, stream, d_idsT, ld_ids, (int)n, cs_ld, cstart, member, band_cnt, band_start, d_entries32, saved);
  hipLaunchKernelGGL(k_sp_scan, dim3(1), dim3(1024), 0, stream, band_cnt, band_start, (int64_t)T);
  hipLaunchKernelGGL(k_sp_emit<1>, grid, dim3(256), 0, stream, d_idsT, ld_ids, (int)n, cs_ld, cstart, member, band_cnt, band_start, d_entries32, saved);
  hipLaunchKernelGGL(k_sp_band, dim3((unsigned)T), dim3(1024), 0, stream, band_start, d_entries32, T, start, d_entries);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}
int launch_mh_sparse(const uint16_t *d_idsT, int64_t ld_ids, int64_t n, int n_hash, int max_ids, uint64_t pairs, uint32_t *d_scratch,
                     uint32_t *d_entries32, uint16_t *d_entries, double *d_out, int64_t ld, hipStream_t stream, hipEvent_t after_buckets) {
  const int rc = launch_mh_sparse_lists(d_idsT, ld_ids, n, n_hash, max_ids, pairs, d_scratch, d_entries32, d_entries, stream);
  if (rc != DA_OK) return rc;
  const SparseScratch w = sparse_scratch(d_scratch, n, n_hash, max_ids, ld_ids);
  if (after_buckets) DA_HIP_TRY(hipEventRecord(after_buckets, stream));
  const int64_t px = ceil_div(w.ntiles, 8);
  const int entries = n_hash + 1;
  hipLaunchKernelGGL(k_sp_tiles, dim3((unsigned)(px * 8)), dim3(256), 128 * ER_STRIDE + (size_t)entries * 8, stream, w.start, d_entries, (int)n, n_hash,
                     entries, d_out, ld, w.T, w.ntiles, px);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

// ---- heavy / rare split: the rare values' incidences ADDED to a finished dense result (dict_kernels.hip k_hy_split) ------------------------
// Tiles with more than HY_SMALL entries -- one workgroup per (tile row, column phase): the entries are counted into a 128 x 128 LDS image (atomics), then every entry's
// thread clears its cell: whoever gets a non-zero count m back owns the pair and applies it -- element (i, j) and its mirror image:
//   float64:  the dense kernel stored count / n_hash (correctly rounded): count = (int)(v * n_hash + 0.5) exactly (|v n_hash - count| < 2^-36),
//             new value (count + m) / n_hash -- the same IEEE division as every other kernel (src/minHash.cpp:174);
//   uint16:   count += m.
// The image is zero again after the swaps.  Diagonal tiles hold i < j only, like every entry list.
constexpr uint32_t HY_SMALL = 64;       // tiles with at most that many entries: one wavefront each, no LDS (k_hy_fixup_small); the rest: k_hy_fixup
template <bool F64>
__device__ __forceinline__ void hy_apply(void *__restrict__ out_v, int64_t ld, int n_hash, int64_t i, int64_t j, uint32_t m) {
  if (F64) {
    double *out = reinterpret_cast<double *>(out_v);
    const double nh = (double)n_hash;
    const uint32_t cnt = (uint32_t)(out[i * ld + j] * nh + 0.5);
    const double nv = (double)(cnt + m) / nh;
    out[i * ld + j] = nv;
    out[j * ld + i] = nv;
  } else {
    uint16_t *out = reinterpret_cast<uint16_t *>(out_v);
    const uint16_t nv = (uint16_t)(out[i * ld + j] + m);
    out[i * ld + j] = nv;
    out[j * ld + i] = nv;
  }
}
// The usual tile has a few dozen entries (7.2e6 over 3.1e5 tiles on the h3n2-like 100k set): a wavefront takes a tile, a lane an entry; how often the
// lane's cell occurs, and whether the lane is the first to hold it, come from a loop of readlanes over the tile's entries -- no LDS, no barrier, so the
// kernel is bound by the three dependent global accesses per tile at 32 waves per CU.
template <bool F64>
__global__ __launch_bounds__(256) void k_hy_fixup_small(const uint32_t *__restrict__ start, const uint16_t *__restrict__ entries, int n_hash,
                                                        void *__restrict__ out_v, int64_t ld, int T, int tile_row_begin) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ti = tile_row_begin + (int)blockIdx.x;
  for (int tj = ti + (int)blockIdx.y * 4 + wave; tj < T; tj += (int)gridDim.y * 4) {
    const int64_t t = sp_tile_index(ti, tj, T);
    const uint32_t e0 = __builtin_amdgcn_readfirstlane(start[t]), cnt = __builtin_amdgcn_readfirstlane(start[t + 1]) - e0;
    if (cnt == 0 || cnt > HY_SMALL) continue;                        // (wave-uniform)
    const uint32_t key = (uint32_t)lane < cnt ? (uint32_t)entries[e0 + lane] : 0xffffffffu;
    uint32_t m = 0;
    bool first = true;
    for (uint32_t q = 0; q < cnt; ++q) {
      const uint32_t kq = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)q);
      m += kq == key;
      first = first && !(kq == key && q < (uint32_t)lane);
    }
    if ((uint32_t)lane < cnt && first) hy_apply<F64>(out_v, ld, n_hash, (int64_t)ti * 128 + (key >> 7), (int64_t)tj * 128 + (key & 127u), m);
  }
}
template <bool F64>
__global__ __launch_bounds__(256) void k_hy_fixup(const uint32_t *__restrict__ start, const uint16_t *__restrict__ entries, int n_hash, void *__restrict__ out_v,
                                                  int64_t ld, int T, int tile_row_begin) {
  __shared__ uint32_t img[128 * 128 / 2];          // two uint16 counts per word (a count is <= n_hash <= 2047: no carry); 32 KiB fits beside the row expansion
  const int tid = threadIdx.x, ti = tile_row_begin + (int)blockIdx.x;
  bool zeroed = false;
  for (int tj = ti + (int)blockIdx.y; tj < T; tj += (int)gridDim.y) {
    const int64_t t = sp_tile_index(ti, tj, T);
    const uint32_t e0 = start[t], e1 = start[t + 1];
    if (e1 - e0 <= HY_SMALL) continue;                               // (uniform; those tiles: k_hy_fixup_small)
    if (!zeroed) {
      for (int w = tid; w < 128 * 128 / 2; w += 256) img[w] = 0u;
      __syncthreads();
      zeroed = true;
    }
    for (uint32_t e = e0 + tid; e < e1; e += 256) {                  // entry = (row & 127) << 7 | (column & 127)
      const uint32_t v = entries[e];
      atomicAdd(&img[v >> 1], 1u << (16 * (v & 1u)));
    }
    __syncthreads();
    for (uint32_t e = e0 + tid; e < e1; e += 256) {
      const uint32_t v = entries[e], sh = 16 * (v & 1u);
      const uint32_t m = (atomicAnd(&img[v >> 1], ~(0xffffu << sh)) >> sh) & 0xffffu;     // the first thread to clear the cell owns the pair
      if (!m) continue;
      hy_apply<F64>(out_v, ld, n_hash, (int64_t)ti * 128 + (v >> 7), (int64_t)tj * 128 + (v & 127u), m);
    }
    __syncthreads();
  }
}
// tile rows [tile_row_begin, tile_row_end) of the lists launch_mh_sparse_lists left in d_scratch / d_entries; kind: DA_OUT_F64 or DA_OUT_COMPACT
int launch_mh_sparse_fixup(const uint32_t *d_scratch, const uint16_t *d_entries, int64_t n, int n_hash, int max_ids, int64_t ld_ids, int kind, void *d_out,
                           int64_t ld, int64_t tile_row_begin, int64_t tile_row_end, hipStream_t stream) {
  const SparseScratch w = sparse_scratch(const_cast<uint32_t *>(d_scratch), n, n_hash, max_ids, ld_ids);
  if (tile_row_end > w.T) tile_row_end = w.T;
  if (tile_row_end <= tile_row_begin) return DA_OK;
  // column phases per tile row: enough workgroups to fill the chip even when a call covers one band of the table (pipelined duplicate route) -- a workgroup
  // walks its tiles one after the other, three dependent global accesses each
  const int64_t rows = tile_row_end - tile_row_begin;
  const unsigned phases = (unsigned)std::max<int64_t>(4, std::min<int64_t>(ceil_div(w.T, 4), ceil_div(4096, rows)));
  const dim3 grid((unsigned)rows, phases);
  if (kind == DA_OUT_F64) {
    hipLaunchKernelGGL(k_hy_fixup_small<true>, grid, dim3(256), 0, stream, w.start, d_entries, n_hash, d_out, ld, w.T, (int)tile_row_begin);
    hipLaunchKernelGGL(k_hy_fixup<true>, grid, dim3(256), 0, stream, w.start, d_entries, n_hash, d_out, ld, w.T, (int)tile_row_begin);
  } else {
    hipLaunchKernelGGL(k_hy_fixup_small<false>, grid, dim3(256), 0, stream, w.start, d_entries, n_hash, d_out, ld, w.T, (int)tile_row_begin);
    hipLaunchKernelGGL(k_hy_fixup<false>, grid, dim3(256), 0, stream, w.start, d_entries, n_hash, d_out, ld, w.T, (int)tile_row_begin);
  }
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

int launch_symmetrize(void *d_mat, int64_t n, int64_t ld, int kind, hipStream_t stream) {
  if (n <= 1) return DA_OK;
  const unsigned t = (unsigned)ceil_div(n, 32);
  dim3 grid(t, t), block(256);
  if (kind == DA_OUT_F64)
    hipLaunchKernelGGL(k_symmetrize<double>, grid, block, 0, stream, (double *)d_mat, n, ld);
  else
    hipLaunchKernelGGL(k_symmetrize<uint16_t>, grid, block, 0, stream, (uint16_t *)d_mat, n, ld);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

int launch_widen(const uint16_t *d_in, double *d_out, int64_t count, bool is_nw, int n_hash,
                 hipStream_t stream) {
  if (count <= 0) return DA_OK;
  int64_t blocks = ceil_div(count, 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(k_widen, dim3((unsigned)blocks), dim3(256), 0, stream, d_in, d_out, count, is_nw ? 1 : 0, n_hash);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

}  // namespace da
