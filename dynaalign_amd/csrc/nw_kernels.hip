// nw_kernels.hip -- gfx950 kernels for the similarityNW hot path.
//
//   K0  k_nw_encode : residue byte -> BLOSUM row index 0..23, flag anything
//                     else (reference src/pairwiseSeqAlign.cpp:15-21, :240-250)
//   K3  k_nw_short  : all-pairs affine-gap global alignment identity for
//                     peptides up to NMAX residues (reference :209-313, driver
//                     :331-365)
//
// K3 design (CDNA4, integer VALU; no MFMA -- a DP recurrence is not a
// contraction):
//   * one LANE per pair: a wavefront owns one row i of the pair space at a
//     time and 64 consecutive columns j, so sequence1 = seq[i] is
//     wave-uniform and sequence2 = seq[j] is lane-private.  The whole
//     (m+1) x (n+1) DP of a pair lives in the lane's registers as one
//     rolling row (M-goe, Ix, packed matches/len per column); there is no
//     cross-lane traffic and every lane is busy on every step.
//   * the traceback (reference :284-308) is replaced by carrying
//     (matches, length) forward along the chosen predecessor -- SURVEY A.2,
//     identical by construction because the reference's traceback reads only
//     the move chosen at fill time.
//   * scores come from a 24x24 table staged in LDS as {score + gapOpen + gapExt,
//     1 + (a==b)<<16}: one conflict-free ds_read_b64 per cell (all lanes of a
//     wave read inside one 192-byte table row).
//   * only pairs with i <= j are evaluated, as calc(seq[i], seq[j])
//     (reference :340-346; the function is not symmetric); the
//     mirrored element is stored from the same lane (:349-350).
#include <algorithm>
#include <climits>
#include <cstdlib>

#include <hipcub/hipcub.hpp>

#include "da_common.hpp"

namespace da {
namespace {

#include "blosum_data.inc"

struct ScoreTable { signed char s[576]; };  // passed by value in the kernarg segment

// residue byte -> index (reference src/pairwiseSeqAlign.cpp:15-21)
__device__ __forceinline__ int aa_index(uint32_t c) {
  switch (c) {
    case 'A': return 0;  case 'R': return 1;  case 'N': return 2;  case 'D': return 3;
    case 'C': return 4;  case 'Q': return 5;  case 'E': return 6;  case 'G': return 7;
    case 'H': return 8;  case 'I': return 9;  case 'L': return 10; case 'K': return 11;
    case 'M': return 12; case 'F': return 13; case 'P': return 14; case 'S': return 15;
    case 'T': return 16; case 'W': return 17; case 'Y': return 18; case 'V': return 19;
    case 'B': return 20; case 'Z': return 21; case 'X': return 22; case '*': return 23;
    default: return -1;
  }
}

__global__ __launch_bounds__(256) void k_nw_encode(const uint8_t *__restrict__ res, int64_t total,
                                                   uint8_t *__restrict__ codes, int32_t *__restrict__ bad) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += stride) {
    const int ix = aa_index(res[p]);
    codes[p] = (uint8_t)(ix < 0 ? 0 : ix);
    if (ix < 0) {
      const int64_t q = p < (int64_t)INT_MAX - 1 ? p : (int64_t)INT_MAX - 1;
      atomicMax(bad, (int32_t)(INT_MAX - q));  // smallest position wins; 0 = all valid
    }
  }
}

constexpr int K3_THREADS = 256;
constexpr int K3_ROWS_PER_WAVE = 16;
constexpr int K3_TILE = 64;  // 4 waves x 16 rows, 64 columns

struct Cell { int32_t s_goe; uint32_t inc; };  // LDS table entry (8 bytes)

// One DP row for a lane: columns 1..NMAX, left to right.
//   MG[c] : M[r-1][c] - (go+ge) on entry, M[r][c] - (go+ge) on exit  (c = 0..NMAX-1 <-> column c+1)
//   X[c]  : Ix[r-1][c]  ->  Ix[r][c]
//   P[c]  : (matches<<16 | len) of cell (r-1,c) -> (r,c)
// FIRST: row 1, where M[0][c] = Ix[0][c] = NEG (reference :230-235), so
// Ix[1][c] = max(NEG-goe, NEG-ge) for every c while the diagonal term still
// sees max(M,Ix,Iy)[0][c-1] = Iy[0][c-1] (kept in MG by the initialisation).
template <int NMAX, bool FIRST>
__device__ __forceinline__ void nw_row(int32_t (&MG)[NMAX], int32_t (&X)[NMAX], uint32_t (&P)[NMAX],
                                       const uint32_t (&boff)[NMAX], const char *tab_row,
                                       int32_t mg_diag0, uint32_t p_diag0, uint32_t p_left0,
                                       int32_t mg_left0, int32_t y_left0, int32_t ge, int32_t goe,
                                       int32_t ix_first, int32_t *y_last = nullptr) {
  int32_t mgd = mg_diag0;   // (max(M,Ix,Iy)[r-1][c-1]) - goe
  uint32_t pd = p_diag0;
  int32_t mgl = mg_left0;   // M[r][c-1] - goe
  int32_t yl = y_left0;     // Iy[r][c-1]
  uint32_t pl = p_left0;
#pragma unroll
  for (int c = 0; c < NMAX; ++c) {
    const Cell e = *reinterpret_cast<const Cell *>(tab_row + boff[c]);
    const int32_t ix = FIRST ? ix_first : max(MG[c], X[c] - ge);          // reference :255-257
    const int32_t iy = max(mgl, yl - ge);                                   // :260-262
    const int32_t d = mgd + e.s_goe;                                        // :265-268
    const int32_t gap = max(ix, iy);
    const bool take_d = d >= gap;                                           // :271
    const bool up_over_left = ix >= iy;                                     // :273
    const int32_t m = max(d, gap);                                          // :272-278 (M overwrite)
    const uint32_t p_gap = (up_over_left ? P[c] : pl) + 1u;
    const uint32_t p_new = take_d ? pd + e.inc : p_gap;
    mgd = MG[c];
    pd = P[c];
    mgl = m - goe;
    yl = iy;
    pl = p_new;
    MG[c] = mgl;
    X[c] = ix;
    P[c] = p_new;
  }
  if (y_last) *y_last = yl;
}

// matches / length exactly as the reference divides them (:311); 0/0 gives the x86 default NaN
__device__ __forceinline__ double nw_ratio(uint32_t mt, uint32_t ln) {
  return ln == 0 ? __longlong_as_double(0xFFF8000000000000ULL) : (double)mt / (double)ln;
}

// ---- "combined key" cell update (fast path) ---------------------------------
// A cell's state is ONE int32:  score' * 2^15 + priority * 2^13 + payload, payload =
// matches * 128 + D, D = number of diagonal moves on the path.  Signed comparison orders by
// score first, then by priority (diagonal 2 > up 1 > left 0 -- exactly the reference's
// tie-break, :271-279), and the payload of the winner rides along, so the choose-and-propagate
// step is one v_max3 instead of two half-rate compares and two v_cndmask (gfx950 issues
// add/sub/and/bitop3 at twice the rate of cmp/cndmask/max).
// Scores are kept in a frame that moves with the cell: score'(r,c) = score(r,c) + (r+c)*gapExt.
// All three candidates of a cell are in the same frame, so every decision is unchanged, but a
// gap EXTENSION costs nothing any more (Ix' = max(M' + (ge - goe), Ix'): add + max instead of
// sub + sub + max) and the diagonal's 2*ge goes into the score table.  Likewise the alignment
// length is not carried: every move adds one column, a diagonal move consumes two indices, so
// length = r + c - D and only diagonal moves touch the payload (through the table).  10 VALU
// ops + 1 LDS read per cell (12 + 1 before the moving frame).
// Valid while scores stay within 17 bits: the launcher uses it for 0 <= gapOpen, gapExt and
// gapOpen + 64*gapExt <= 7000 with the "minus infinity" sentinel at -24000 (any value below every
// reachable score gives the same decisions as the reference's INT_MIN/2; the frame moves it by at
// most 64*gapExt; the int32 kernel covers everything else).
// Bit budget per length class (the launcher checks the penalty limit):
//   <= 32 residues: D 7 bits, matches 6 bits -> score 17 bits, sentinel -24000, gapOpen + 64*gapExt <= 7000
//   <= 64 residues: D 8 bits, matches 7 bits -> score 15 bits, sentinel  -6000, gapOpen + 128*gapExt <= 2500
template <int NMAX> struct CKBits {
  static constexpr int LB = NMAX <= 32 ? 7 : 8;              // bits of the alignment length
  static constexpr int S = NMAX <= 32 ? 13 : 15;             // payload bits (matches << LB | length)
  static constexpr int S2 = S + 2;                           // score starts here (2 priority bits)
  static constexpr int32_t LOW = (1 << S2) - 1;              // priority + payload bits
  static constexpr int32_t PRI = 3 << S;
  static constexpr int32_t NEG = NMAX <= 32 ? -24000 : -6000;
  static constexpr int GAP_LIMIT = NMAX <= 32 ? 7000 : 2500; // gapOpen + 2*NMAX*gapExt must stay below
};

//   VM[c] : combined best'[r-1][c] (priority cleared)     XP[c] : Ix'[r-1][c]*2^15 | priority 1 | payload of cell (r-1,c)
// PACKB: sequence2's table offsets (<= 92 bytes each) four to a register in boff[0 .. NMAX / 4) -- one v_bfe_u32 more per cell, off the dependent chain
// (the row is bound by that chain, profiles/r04_c_*), fifteen VGPRs less: what lets the prefix-sharing kernel run five waves per SIMD
template <int NMAX, bool FIRST, bool PACKB = false>
__device__ __forceinline__ void nw_row_ck(int32_t (&VM)[NMAX], int32_t (&XP)[NMAX], const uint32_t (&boff)[NMAX],
                                          const char *tab_row, int32_t vm_diag0, int32_t vm_left0, int32_t yp_left0,
                                          int32_t kx, int32_t ky, int32_t ixf_first, int32_t pay_mask, int32_t pri_clear) {
  int32_t vmd = vm_diag0, vml = vm_left0, ypl = yp_left0;
#pragma unroll
  for (int c = 0; c < NMAX; ++c) {
    int32_t e;
    if constexpr (PACKB) {
      // byte (c & 3) of the packed offsets + the row's table address in ONE SDWA add, as an asm statement: written in C++ the unpacking is loop-invariant
      // and hoisted out of the row loop -- twenty registers again
      typedef __attribute__((address_space(3))) const char lds_char_t;
      typedef __attribute__((address_space(3))) const int32_t lds_i32_t;
      const uint32_t base = (uint32_t)(uintptr_t)(lds_char_t *)tab_row;
      uint32_t addr;
      if ((c & 3) == 0) asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(addr) : "v"(boff[c >> 2]), "v"(base));
      else if ((c & 3) == 1) asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(addr) : "v"(boff[c >> 2]), "v"(base));
      else if ((c & 3) == 2) asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(addr) : "v"(boff[c >> 2]), "v"(base));
      else asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(addr) : "v"(boff[c >> 2]), "v"(base));
      e = *(lds_i32_t *)(uintptr_t)addr;
    } else {
      e = *reinterpret_cast<const int32_t *>(tab_row + boff[c]);
    }
    // up: max(M-goe, Ix-ge), priority 1, payload of the cell above            (reference :255-257, :273-275)
    const int32_t ixf = FIRST ? ixf_first : max(VM[c] + kx, XP[c]);
    // left: max(M-goe, Iy-ge), priority 0, payload of the cell to the left             (:260-262, :276-278)
    const int32_t iyf = max(vml + ky, ypl);
    const int32_t vd = vmd + e;                       // diagonal, priority 2, D + 1 (+128 on a match)  (:265-271)
    const int32_t w = max(max(vd, ixf), iyf);         // v_max3_i32
    const int32_t vmn = w & pri_clear;
    vmd = VM[c];
    VM[c] = vmn;
    // (vmn & PAY) | (ixf & ~PAY) as one full-rate v_bitop3 (S0=0xF0,S1=0xCC,S2=0xAA: (S1&S2)|(S0&~S2) = 0xD8):
    // the gap state's score (and its own priority bits) with the payload of the cell it leaves
    XP[c] = __builtin_amdgcn_bitop3_b32(ixf, vmn, pay_mask, 0xD8);
    ypl = __builtin_amdgcn_bitop3_b32(iyf, vmn, pay_mask, 0xD8);
    vml = vmn;
  }
}

// Hand-scheduled rows (round 4; tools/gen_nw_asm.py -> tools/experiments/nw_rows_p<NMAX>.inc): all DP rows of one sequence1 as ONE asm
// statement -- two rows per sweep skewed by a column (no copy of the diagonal neighbour), every per-cell operand a VGPR, table reads
// through a ring.  Bit-exact (CPU model: tests/test_nw_asm_model.py; GPU: the test suite run on the experiment library) and NOT faster
// than the compiled row -- ordered DP 102 vs 97.5 ms, direct sweep 424 vs 417 ms at 100k: the row is bound by its three max-class
// instructions, not by scheduling (profiles/r04_c_*) -- so the product library does not carry it: ASM = true instances exist only in
// the experiment twin of the library (tools/experiments/build.sh, -DDA_K2_EXPERIMENTS; DYNAALIGN_NW_ASM=1 selects them there).
#ifdef DA_K2_EXPERIMENTS
#include "nw_rows_p12_bind.inc"
#include "nw_rows_p20_bind.inc"
template <int NMAX> constexpr bool nw_has_asm_rows() { return NMAX == 12 || NMAX == 20; }
#else
template <int NMAX> constexpr bool nw_has_asm_rows() { return false; }
#endif
// PFX (round 4, ordered mode): PREFIX SHARING.  The DP state after row r depends only on sequence1[0 .. r) and sequence2, so two
// sequence1 strings with a common prefix of p residues share the first p DP rows of every pair.  The rows of the ordered table are
// therefore PROCESSED in lexicographic order of the unique strings (ord_perm: sorted position -> unique id; the table itself keeps its
// layout -- a wave stores row id ord_perm[pos]) and a wave walking its 16 sorted rows keeps ONE checkpoint per lane (the state
// of the 2 NMAX row registers -- in LDS -- at the depth the next needed row shares with the current one): that row resumes there instead of at row 0.  ord_lcp[pos] =
// common prefix of the strings at sorted positions pos - 1 and pos.  Same cells, same arithmetic: bit-identical; 30 % fewer DP rows on
// the h3n2-like headline set (mean shared prefix of sorted neighbours 8.5 of 20), 12 % on uniform peptides.
template <int NMAX, bool CK, bool ORD, bool ASM = false, bool PFX = false>
__global__ __launch_bounds__(K3_THREADS, (NMAX <= 24 ? (CK && ORD && !ASM ? 5 : 4) : 1)) void k_nw_short(   // <= 24 residues: 4 waves per SIMD (128 VGPRs); the ordered mode FIVE (96 VGPRs; PFX: 30 KB of LDS) since its table offsets are packed (round 4b; the direct sweep: 422 ms that way, 404 at four)
    const uint8_t *__restrict__ codes, const int64_t *__restrict__ offsets, int64_t n,
    ScoreTable table, int32_t go, int32_t ge, int64_t row_begin, int64_t row_end, int symmetric,
    int kind, void *__restrict__ out_v, int64_t ld, int32_t *__restrict__ score_out,
    int64_t ld_score, int64_t ntiles, int T, int shard_rank, int shard_world, int fold_q, int64_t fold_w,
    const int32_t *__restrict__ ord_first, const int32_t *__restrict__ ord_minfirst, const int32_t *__restrict__ ord_maxlast,
    const int32_t *__restrict__ ord_perm, const uint8_t *__restrict__ ord_lcp) {
  // ord_first != NULL: ORDERED mode on a table of UNIQUE sequences (nw_dedup below): element (p, q) of the full square is
  // calc(U_p, U_q) with U_p as sequence1 whatever the order of p and q, computed only where some pair i < j of the original
  // input maps to it: first(p) < last(q) (or p == q).
  constexpr bool ordered = ORD;               // (a template parameter: the two modes show up as two kernels in profiles)
  const bool f64_out = kind == DA_OUT_F64;
  __shared__ __attribute__((aligned(16))) Cell tab[CK ? 1 : 24 * 24];
  __shared__ int32_t tabk[CK ? 24 * 24 : 1];
  __shared__ uint8_t rowcodes[K3_TILE][NMAX];
  __shared__ int32_t rowlen[K3_TILE];
  __shared__ int32_t rowid[PFX ? K3_TILE : 1];      // PFX: unique id of the row at this sorted position (-1 past the end)
  __shared__ uint8_t rowlcp[PFX ? K3_TILE : 1];     // ... its common prefix with the previous sorted row (0 at a wave's first row)
  __shared__ uint8_t rowneed[PFX ? K3_TILE : 1];    // ... whether this tile needs the row at all
  __shared__ uint32_t mirror_res[ORD ? 1 : K3_THREADS / 64][ORD ? 1 : K3_ROWS_PER_WAVE][64];   // (ordered mode never mirrors)
  // PFX: a lane's checkpoint lives in LDS, [word][thread]: NMAX words of VM + NMAX / 2 words of Ix' scores packed two by two (the rest of an
  // XP word is priority 1 + the payload of the same column's VM) -- 30 KB at NMAX = 20, four workgroups per CU, no extra VGPRs
  // round 4b: the Ix' part as BYTES: d = score(VM[c]) - score(XP[c]) is >= 0 (VM is the max of three that include Ix') and only matters up to `go` --
  // the next row takes max(VM[c] + kx, XP[c]) with score(VM[c] + kx) = score(VM[c]) - go, same priority, same payload: at d >= go the first operand wins or
  // ties bit for bit -- so min(d, go) <= 255 is stored, four columns per word: 25 words = 25.6 KB at NMAX = 20, FIVE workgroups per CU
  static_assert(!PFX || NMAX % 4 == 0, "the checkpoint packs four columns per word");
  __shared__ uint32_t cp_lds[PFX ? NMAX + NMAX / 4 : 1][PFX ? K3_THREADS : 1];

  // ---- tile decode (upper-triangular 64x64 tiles of the pair space)
  const int64_t L = blockIdx.x;
  if (L >= ntiles) return;
  int ti, tj;
  bool allow_direct = true, allow_mirror = true;
  int64_t row_shift = -row_begin;  // local output row = global row + row_shift
  int64_t col_shift = 0;           // local output column = global column + col_shift (direct stores)
  if (ordered) {
    ti = (int)(row_begin / K3_TILE) + (int)(L / T);              // (row_begin, row_end: whole 64-row tile rows of the unique table)
    tj = (int)(L % T);
    if (shard_world > 0) {     // one rank's cyclic 128-row units of the ordered table, stored back to back: local unit u = global unit u * world + rank
      const int q64 = (int)(L / T), u = q64 >> 1;
      ti = 2 * (u * shard_world + shard_rank) + (q64 & 1);
      if (ti >= T) return;
      row_shift = (int64_t)u * 128 + (int64_t)(q64 & 1) * K3_TILE - (int64_t)ti * K3_TILE;
    }
    if (!PFX && ti != tj && ord_minfirst[ti] >= ord_maxlast[tj]) return;   // no original pair i < j needs this tile (PFX: rows are in sorted order, decided per row)
    allow_mirror = false;
  } else if (symmetric) {
    // row-major over the upper triangle: row t holds T - t tiles
    const double Td = (double)T;
    int64_t t = (int64_t)(Td + 0.5 - sqrt((Td + 0.5) * (Td + 0.5) - 2.0 * (double)L));
    if (t < 0) t = 0;
    if (t > T - 1) t = T - 1;
    auto start = [&](int64_t r) { return r * T - r * (r - 1) / 2; };
    while (t > 0 && start(t) > L) --t;
    while (t + 1 <= T - 1 && start(t + 1) <= L) ++t;
    ti = (int)t;
    tj = (int)(t + (L - start(t)));
  } else if (shard_world > 0) {
    // cyclic shard of one rank: local unit u is global 128-row unit u*world + rank; only
    // upper tiles, direct store into local rows q*64 + [0,64) (the mirror is filled after
    // the all-gather by k_finalize_sharded)
    // ranks are dealt 128-row UNITS (two of this kernel's 64-row tile rows), the same unit and folded layout as the
    // MinHash shards -- so the histogram / edge-extraction kernels (graph_kernels.hip) read both kinds of block
    const int q64 = (int)(L / T);                         // local 64-row tile row
    const int u = q64 >> 1;                               // local unit
    const int gu = u * shard_world + shard_rank;          // global unit
    ti = 2 * gu + (q64 & 1);
    tj = (int)(L % T);
    if (ti >= T || tj < ti) return;
    allow_mirror = false;
    {  // folded shard layout (ShardGeom, tile = 128): units u and Q-1-u share a stored unit row
      const bool front = u <= fold_q - 1 - u;
      row_shift = (int64_t)(front ? u : fold_q - 1 - u) * 128 + (int64_t)(q64 & 1) * K3_TILE - (int64_t)ti * K3_TILE;
      col_shift = front ? -(int64_t)gu * 128 : shard_back(fold_w, n);
    }
  } else {
    // row-block request: tile row rt (inside the block) x every tile column tc.
    // tc >= rt is the upper tile itself (direct store); tc < rt is served by
    // the upper tile (tc, rt) through its mirrored store, so each element of
    // the block is produced exactly once.
    const int rt = (int)(row_begin / K3_TILE) + (int)(L / T);
    const int tc = (int)(L % T);
    ti = tc >= rt ? rt : tc;
    tj = tc >= rt ? tc : rt;
    allow_direct = tc >= rt;
    allow_mirror = tc <= rt;
  }
  const int64_t I0 = (int64_t)ti * K3_TILE, J0 = (int64_t)tj * K3_TILE;
  const int32_t goe = go + ge;
  const int32_t NEG = INT_MIN / 2;

  // ---- stage the score table and this tile's 64 row sequences in LDS
  for (int e = threadIdx.x; e < 576; e += K3_THREADS) {
    const int a = e / 24, b = e - a * 24;
    if (CK) {
      tabk[e] = (((int32_t)table.s[e] + 2 * ge) << CKBits<NMAX>::S2) + (2 << CKBits<NMAX>::S) + 1 + ((a == b) ? (1 << CKBits<NMAX>::LB) : 0);
    } else {
      tab[e].s_goe = (int32_t)table.s[e] + goe;
      tab[e].inc = 1u + ((a == b) ? 0x10000u : 0u);  // equal index <=> equal residue byte (:291-293)
    }
  }
  for (int r = threadIdx.x >> 2; r < K3_TILE; r += K3_THREADS / 4) {
    int64_t i = I0 + r;
    if (PFX) {
      const int64_t pos = i;
      i = pos < n ? (int64_t)ord_perm[pos] : n;                    // the row's unique id
      if ((threadIdx.x & 3) == 0) {
        rowid[r] = pos < n ? (int32_t)i : -1;
        rowlcp[r] = (pos < n && r != 0) ? ord_lcp[pos] : (uint8_t)0;
        rowneed[r] = (pos < n && ((int)(i / K3_TILE) == tj || ord_first[i] < ord_maxlast[tj])) ? (uint8_t)1 : (uint8_t)0;
      }
    }
    const int64_t b = i < n ? offsets[i] : 0;
    const int32_t len = i < n ? (int32_t)(offsets[i + 1] - b) : 0;
    if ((threadIdx.x & 3) == 0) rowlen[r] = len;
    for (int q = threadIdx.x & 3; q < NMAX; q += 4) rowcodes[r][q] = q < len ? codes[b + q] : 0;
  }

  // ---- lane-private sequence2
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t j = J0 + lane;
  const bool jvalid = j < n;
  int32_t nj = 0;
  uint32_t boff[NMAX];
  {
    const int64_t b = jvalid ? offsets[j] : 0;
    nj = jvalid ? (int32_t)(offsets[j + 1] - b) : 0;
#pragma unroll
    for (int c = 0; c < NMAX; ++c)
      boff[c] = (c < nj ? (uint32_t)codes[b + c] : 0u) * (uint32_t)(CK ? sizeof(int32_t) : sizeof(Cell));
    if (CK && ORD && NMAX <= 24) {                       // (ordered mode: four offsets per register, see nw_row_ck<.., PACKB>)
#pragma unroll
      for (int c = 0; c < NMAX; c += 4) boff[c >> 2] = boff[c] | (boff[c + 1] << 8) | (boff[c + 2] << 16) | (boff[c + 3] << 24);
    }
  }
  __syncthreads();

  const int32_t ix_first = max(NEG - goe, NEG - ge);
  const char *tab_bytes = CK ? reinterpret_cast<const char *>(tabk) : reinterpret_cast<const char *>(tab);

  // The mirrored element of row i lands in row j of the output: one lane, one row.  Storing it
  // per DP row would scatter 8-byte writes over 64 cache lines per instruction (3x write
  // amplification measured); instead a lane parks its 16 results in LDS (own slot, no sync
  // needed) and stores them as one contiguous, line-aligned run out[j][i0 .. i0+15] at the end.
  uint32_t *my_res = &mirror_res[ORD ? 0 : wave][0][lane];
  if (!ORD) {
#pragma unroll
    for (int rr = 0; rr < K3_ROWS_PER_WAVE; ++rr) my_res[rr * 64] = 0xffffffffu;  // = nothing to mirror
  }

  // (values read from LDS / derived from the wave id are wave-uniform, but the compiler cannot know: through v_readfirstlane they -- and the loop
  //  counters and addresses computed from them -- stay in SGPRs: without it the PFX kernel kept the row loop's counter, limit and residue address in
  //  VGPRs, ran the loop under an exec mask and, at five waves per SIMD, spilled the address inside the row loop)
  auto uni = [](int x) -> int { return __builtin_amdgcn_readfirstlane(x); };
  // PFX: the checkpoint (state of a row computed earlier in this wave's group at depth cp_depth) and the smallest common prefix met since
  int cp_depth = 0, since_min = 255, pfx_start = 0, pfx_save = 0;
  // the rows a wave takes: 16 consecutive ones -- or, PFX, a quarter of the tile's NEEDED rows by estimated cost, consecutive in sorted order, so that the four
  // waves of a workgroup finish together (with fixed groups of 16 -- or equal counts -- a workgroup kept its slot for its slowest wave:
  // 25 % fewer instructions gave 8 % less time)
  int it_begin = uni(wave) * K3_ROWS_PER_WAVE, it_end = it_begin + K3_ROWS_PER_WAVE;
  if (PFX) {
    // estimated DP rows of every needed row: its length minus what it shares with the previous needed row; a wave takes the consecutive
    // needed rows whose running cost falls into its quarter
    const unsigned long long need_mask = __ballot(rowneed[lane] != 0);
    int total = 0;
    {
      int mn = 255;
      bool have_prev = false;
      for (int r = 0; r < K3_TILE; ++r) {
        mn = min(mn, uni((int)rowlcp[r]));
        if ((need_mask >> r) & 1ull) {
          total += max(1, uni(rowlen[r]) - (have_prev ? mn : 0)) + 1;
          mn = 255;
          have_prev = true;
        }
      }
    }
    const int waves = K3_THREADS / 64;
    const int lo = uni(wave) * total / waves, hi = (uni(wave) + 1) * total / waves;   // this wave: rows whose running cost starts in [lo, hi)
    it_begin = it_end = 0;
    {
      int mn = 255, run = 0;
      bool have_prev = false, any = false;
      for (int r = 0; r < K3_TILE; ++r) {
        mn = min(mn, uni((int)rowlcp[r]));
        if ((need_mask >> r) & 1ull) {
          if (run >= lo && run < hi) {
            if (!any) { it_begin = r; any = true; }
            it_end = r + 1;
          }
          run += max(1, uni(rowlen[r]) - (have_prev ? mn : 0)) + 1;
          mn = 255;
          have_prev = true;
        }
      }
    }
  }
  for (int it = it_begin; it < it_end; ++it) {
    const int lr = it, rr = PFX ? 0 : it - wave * K3_ROWS_PER_WAVE;
    int64_t i = I0 + lr;
    if (i >= n) break;
    if (PFX) {
      since_min = min(since_min, uni((int)rowlcp[lr]));
      if (!uni((int)rowneed[lr])) continue;
      i = uni(rowid[lr]);
      pfx_start = (cp_depth > 0 && cp_depth <= since_min) ? cp_depth : 0;
      // the depth the NEXT needed row of this wave shares with this one: checkpoint there, if that row is computed now
      pfx_save = 0;
      {
        int mn = 255;
        for (int l2 = lr + 1; l2 < it_end; ++l2) {
          mn = min(mn, uni((int)rowlcp[l2]));
          if (uni((int)rowneed[l2])) { pfx_save = mn; break; }
        }
      }
      if (pfx_save <= pfx_start) pfx_save = 0;
    }
    if (!ordered && J0 + 63 < i) continue;  // the whole wave is below the diagonal
    if (!PFX && ordered && ti != tj && ord_first[i] >= ord_maxlast[tj]) continue;   // nothing in this row of the tile is needed
    const bool want_direct = allow_direct && i >= row_begin && i < row_end;
    const bool want_mirror_any = allow_mirror && J0 < row_end && J0 + 63 >= row_begin;
    if (!want_direct && !want_mirror_any) continue;
    const int32_t m = uni(rowlen[lr]);

    uint32_t mt, ln;
    int32_t sc;
    if constexpr (CK) {
      constexpr int CK_S = CKBits<NMAX>::S, CK_S2 = CKBits<NMAX>::S2, CK_LB = CKBits<NMAX>::LB;
      constexpr int32_t CK_PRI = CKBits<NMAX>::PRI, CK_NEG = CKBits<NMAX>::NEG;
      // row 0 (reference :222-235) in combined form: only its max(M,Ix,Iy) feeds row 1's diagonal.
      // Iy[0][c+1] = -go - c*ge, in the moving frame (+ (c+1)*ge) the constant ge - go; no diagonal moves yet
      int32_t VM[NMAX], XP[NMAX];
      if constexpr (!ASM) {                             // (the generated block initialises its own rows; with m == 0 VM is never read)
#pragma unroll
        for (int c = 0; c < NMAX; ++c) {
          VM[c] = (ge - go) << CK_S2;
          XP[c] = 0;                                    // Ix[0][.] = -inf is handled by FIRST
        }
      }
      typedef __attribute__((address_space(3))) const uint8_t lds_u8_t;
#ifdef DA_K2_EXPERIMENTS
      if constexpr (ASM) {
        static_assert(nw_has_asm_rows<NMAX>(), "no generated row block for this NMAX");
        if (m > 0) {
          // wave-uniform operands in SGPRs; the block moves the per-cell constants into VGPRs itself (an SGPR source halves the rate of
          // v_add / v_bitop3, profiles/r04_a_ubench_inst_rate.txt)
          const uint32_t m_s = __builtin_amdgcn_readfirstlane((uint32_t)m);
          const uint32_t rc_s = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8_t *)&rowcodes[lr][0]);
          const uint32_t tb_s = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8_t *)reinterpret_cast<const uint8_t *>(tabk));
          const int32_t kx_s = ((ge - goe) << CK_S2) + (1 << CK_S), ky_s = (ge - goe) << CK_S2, pm_s = (1 << CK_S) - 1, pc_s = ~CK_PRI;
          const int32_t vi_s = (ge - go) << CK_S2, l0_s = CK_NEG << CK_S2, xf_s = ((CK_NEG - min(goe, ge)) << CK_S2) + (1 << CK_S);
          if constexpr (NMAX == 12) {
            NW_ASM_DECL_12
            asm volatile(
#include "nw_rows_p12.inc"
                : NW_ASM_OUTS_12
                : [m] "s"(m_s), [rc] "s"(rc_s), [tb] "s"(tb_s), [kx] "s"(kx_s), [ky] "s"(ky_s), [pm] "s"(pm_s), [pc] "s"(pc_s),
                  [vi] "s"(vi_s), [l0] "s"(l0_s), [xf] "s"(xf_s), NW_ASM_INS_12
                : NW_ASM_CLOBBERS_12);
            NW_ASM_COPY_12
          } else {
            NW_ASM_DECL_20
            asm volatile(
#include "nw_rows_p20.inc"
                : NW_ASM_OUTS_20
                : [m] "s"(m_s), [rc] "s"(rc_s), [tb] "s"(tb_s), [kx] "s"(kx_s), [ky] "s"(ky_s), [pm] "s"(pm_s), [pc] "s"(pc_s),
                  [vi] "s"(vi_s), [l0] "s"(l0_s), [xf] "s"(xf_s), NW_ASM_INS_20
                : NW_ASM_CLOBBERS_20);
            NW_ASM_COPY_20
          }
        }
      } else
#endif
      {
      // wave-uniform constants are parked in VGPRs: an SGPR source halves v_bitop3's issue rate
      auto in_vgpr = [](int32_t x) { int32_t v; asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(x)); return v; };
      const int32_t kx = in_vgpr(((ge - goe) << CK_S2) + (1 << CK_S));   // open a gap from M': -goe, +ge of the frame, priority 1
      const int32_t ky = in_vgpr((ge - goe) << CK_S2);                   // same to the left, priority 0
      const int32_t pay_mask = in_vgpr((1 << CK_S) - 1), pri_clear = in_vgpr(~CK_PRI);
      // Ix[1][c] = max(NEG-goe, NEG-ge), priority 1, payload of row 0 (nothing)
      const int32_t ixf_first = ((CK_NEG - min(goe, ge)) << CK_S2) + (1 << CK_S);
      // DIRECT sweep: sequence1's residue of the NEXT row is fetched while this row's chain runs (round 3: 466 -> 417 ms; the row loop
      // otherwise opens with ds_read_u8 + s_waitcnt lgkmcnt(0) + v_mad in front of its 20 table reads, and hipcc rotates a plain C++
      // prefetch back to that shape).  The value is in flight ACROSS two asm statements (read in one, wait in the next), so it lives in
      // a FIXED register that nothing else in the kernel names (with an ordinary "=v" variable a compiler-inserted copy or
      // spill between the two statements would capture a stale value; tests/test_nw_asm_model.py checks the disassembly: v127 appears
      // only in these statements).  LDS operations return in order, so the extra outstanding read only makes the compiler's own
      // counted lgkmcnt waits stricter.  The ordered mode does not use it (measured slower there: 98 -> 102 ms).
      constexpr bool PREFETCH = !ORD && NMAX <= 24;
      const uint32_t rc_addr = (uint32_t)(uintptr_t)(lds_u8_t *)&rowcodes[lr][0];
      register uint32_t code_v asm("v127");
      if (PREFETCH && m > 0) asm volatile("ds_read_u8 %0, %1" : "=v"(code_v) : "v"(rc_addr) : "memory");
      if (PFX && pfx_start > 0) {                                    // resume: the checkpoint's state
#pragma unroll
        for (int c = 0; c < NMAX; ++c) VM[c] = (int32_t)cp_lds[c][threadIdx.x];
#pragma unroll
        for (int c = 0; c < NMAX; c += 4) {
          const uint32_t pk = cp_lds[NMAX + c / 4][threadIdx.x];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int32_t d = (int32_t)((pk >> (8 * q)) & 0xffu);
            XP[c + q] = (int32_t)((uint32_t)((VM[c + q] >> CK_S2) - d) << CK_S2) | (1 << CK_S) | (VM[c + q] & ((1 << CK_S) - 1));
          }
        }
      }
      // the DP rows (r_from, r_to]; PFX runs them in two pieces with the checkpoint in between -- kept OUT of the row loop: with the save block inside it the
      // loop did not fit the 96 VGPRs of five waves per SIMD (three scratch reloads per DP row)
      const int32_t r_cut = (PFX && pfx_save > 0) ? pfx_save : m;      // PFX: rows (pfx_start, pfx_save], checkpoint, rows (pfx_save, m] -- ONE copy of the row loop
#pragma nounroll
      for (int piece = 0; piece < (PFX ? 2 : 1); ++piece) {
      const int32_t r_from = piece == 0 ? (PFX ? pfx_start : 0) : r_cut, r_to = piece == 0 ? r_cut : m;
      for (int32_t r = r_from + 1; r <= r_to; ++r) {
        // (making this offset opaque to the compiler turns the per-cell v_mad into a v_add but lets it
        // hoist all 20 lookups: 141 VGPRs / 3 waves per SIMD and 15 % slower -- measured, not kept)
        uint32_t row_off;
        if (PREFETCH) {
          uint32_t code_s;
          asm volatile("s_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %0, %1" : "=s"(code_s), "+v"(code_v) :: "memory");
          row_off = code_s * (24u * (uint32_t)sizeof(int32_t));
          asm volatile("ds_read_u8 %0, %1" : "=v"(code_v) : "v"(rc_addr + (uint32_t)(r < m ? r : r - 1)) : "memory");
        } else {
          row_off = (uint32_t)rowcodes[lr][r - 1] * (24u * (uint32_t)sizeof(int32_t));
        }
        const char *tab_row = tab_bytes + row_off;
        // column 0 of rows r-1 and r (reference :224-229): max(M,Ix,Iy)[r-1][0] = Ix = -go - (r-2)*ge (frame:
        // ge - go) and M = Iy = -inf at (r,0)
        const int32_t vm_diag0 = (r == 1) ? 0 : ((ge - go) << CK_S2);
        const int32_t left0 = CK_NEG << CK_S2;
        if (r == 1)
          nw_row_ck<NMAX, true, (ORD && NMAX <= 24)>(VM, XP, boff, tab_row, vm_diag0, left0, left0, kx, ky, ixf_first, pay_mask, pri_clear);
        else
          nw_row_ck<NMAX, false, (ORD && NMAX <= 24)>(VM, XP, boff, tab_row, vm_diag0, left0, left0, kx, ky, ixf_first, pay_mask, pri_clear);
      }
        if (PFX && piece == 0 && pfx_save > 0) {                     // the next needed row of the group shares this many rows: checkpoint
#pragma unroll
          for (int c = 0; c < NMAX; ++c) cp_lds[c][threadIdx.x] = (uint32_t)VM[c];
#pragma unroll
          for (int c = 0; c < NMAX; c += 4) {
            uint32_t pk = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) pk |= (uint32_t)min((VM[c + q] >> CK_S2) - (XP[c + q] >> CK_S2), go) << (8 * q);   // (go <= 255: the launcher's condition)
            cp_lds[NMAX + c / 4][threadIdx.x] = pk;
          }
        }
      }
      if (PFX && pfx_save > 0) { cp_depth = pfx_save; since_min = 255; }
      }
      // ---- cell (m, nj): length = m + nj - D, score = score' - (m + nj)*ge
      mt = 0; ln = (uint32_t)m;                     // nj == 0: column-0 boundary
      sc = (m == 0) ? 0 : NEG;
      if (m == 0) {
        ln = (uint32_t)nj;
        sc = (nj == 0) ? 0 : NEG;
      } else {
#pragma unroll
        for (int c = 0; c < NMAX; ++c)
          if (nj == c + 1) {
            mt = ((uint32_t)VM[c] >> CK_LB) & ((1u << (CK_S - CK_LB)) - 1u);
            ln = (uint32_t)(m + nj) - ((uint32_t)VM[c] & ((1u << CK_LB) - 1u));
            sc = (VM[c] >> CK_S2) - (m + nj) * ge;
          }
      }
    } else {
    // row 0 of the DP (reference :222-235)
    int32_t MG[NMAX], X[NMAX];
    uint32_t P[NMAX];
#pragma unroll
    for (int c = 0; c < NMAX; ++c) {
      const int32_t iy0 = -go - c * ge;  // Iy[0][c+1] = -go - ((c+1)-1)*ge
      MG[c] = max(NEG, iy0) - goe;       // max(M,Ix,Iy)[0][c+1] - goe, feeds row 1's diagonal
      X[c] = NEG;
      P[c] = (uint32_t)(c + 1);          // 0 matches, length c+1
    }
    for (int32_t r = 1; r <= m; ++r) {
      const uint32_t a = rowcodes[lr][r - 1];
      const char *tab_row = tab_bytes + a * (24u * (uint32_t)sizeof(Cell));
      // column-0 boundary of rows r-1 and r (reference :224-229)
      const int32_t hb_prev = (r == 1) ? 0 : max(NEG, -go - (r - 2) * ge);  // max(M,Ix,Iy)[r-1][0]
      const int32_t mg_diag0 = hb_prev - goe;
      const uint32_t p_diag0 = (uint32_t)(r - 1), p_left0 = (uint32_t)r;
      if (r == 1)
        nw_row<NMAX, true>(MG, X, P, boff, tab_row, mg_diag0, p_diag0, p_left0, NEG - goe, NEG, ge, goe, ix_first);
      else
        nw_row<NMAX, false>(MG, X, P, boff, tab_row, mg_diag0, p_diag0, p_left0, NEG - goe, NEG, ge, goe, ix_first);
    }

    // ---- cell (m, nj)
    uint32_t p = (uint32_t)m;  // nj == 0: length m, 0 matches (column-0 boundary)
    sc = (m == 0) ? 0 : NEG;
    if (m == 0) {              // no rows were run: the arrays still hold DP row 0
      p = (uint32_t)nj;
      sc = (nj == 0) ? 0 : NEG;
    } else {
#pragma unroll
      for (int c = 0; c < NMAX; ++c)
        if (nj == c + 1) { p = P[c]; sc = MG[c] + goe; }
    }
    mt = p >> 16;
    ln = p & 0xffffu;
    }

    if (!jvalid || (!ordered && j < i)) continue;
    const bool do_direct = want_direct;
    const bool do_mirror = allow_mirror && (j != i) && j >= row_begin && j < row_end;
    if (!ORD && do_mirror) my_res[rr * 64] = (mt << 16) | ln;
    if (do_direct) {
      if (f64_out) {
        reinterpret_cast<double *>(out_v)[(i + row_shift) * ld + j + col_shift] = nw_ratio(mt, ln);
      } else if (kind == DA_OUT_PACK32) {
        reinterpret_cast<uint32_t *>(out_v)[(i + row_shift) * ld + j + col_shift] = (mt << 16) | ln;
      } else {
        reinterpret_cast<uint16_t *>(out_v)[(i + row_shift) * ld + j + col_shift] = (uint16_t)((mt << 8) | (ln & 0xffu));
      }
    }
    if (score_out) {
      if (do_direct) score_out[(i + row_shift) * ld_score + j] = sc;
      if (do_mirror) score_out[(j + row_shift) * ld_score + i] = sc;
    }
  }

  // ---- mirrored run of this lane: out[j][i0 + q], q = 0..15
  if (!ORD && jvalid && allow_mirror) {
    const int64_t i0 = I0 + wave * K3_ROWS_PER_WAVE;
    const int64_t base = (j + row_shift) * ld + i0;
    if (f64_out) {
      double *o = reinterpret_cast<double *>(out_v) + base;
      const bool aligned = (reinterpret_cast<uintptr_t>(o) & 15) == 0;
#pragma unroll 1
      for (int q = 0; q < K3_ROWS_PER_WAVE; q += 2) {
        const uint32_t r0 = my_res[q * 64], r1 = my_res[(q + 1) * 64];
        const bool v0 = r0 != 0xffffffffu, v1 = r1 != 0xffffffffu;
        if (v0 && v1 && aligned) {
          *reinterpret_cast<double2 *>(o + q) = make_double2(nw_ratio(r0 >> 16, r0 & 0xffffu), nw_ratio(r1 >> 16, r1 & 0xffffu));
        } else {
          if (v0) o[q] = nw_ratio(r0 >> 16, r0 & 0xffffu);
          if (v1) o[q + 1] = nw_ratio(r1 >> 16, r1 & 0xffffu);
        }
      }
    } else if (kind == DA_OUT_PACK32) {
      uint32_t *o = reinterpret_cast<uint32_t *>(out_v) + base;
#pragma unroll 1
      for (int q = 0; q < K3_ROWS_PER_WAVE; ++q) {
        const uint32_t r0 = my_res[q * 64];
        if (r0 != 0xffffffffu) o[q] = r0;
      }
    } else {
      uint16_t *o = reinterpret_cast<uint16_t *>(out_v) + base;
#pragma unroll 1
      for (int q = 0; q < K3_ROWS_PER_WAVE; ++q) {
        const uint32_t r0 = my_res[q * 64];
        if (r0 != 0xffffffffu) o[q] = (uint16_t)(((r0 >> 16) << 8) | (r0 & 0xffu));
      }
    }
  }
}


// ---------------------------------------------------------------- K4 --
// k_nw_long: sequences longer than the register-resident kernel takes (up to
// K4_MAXLEN residues).  ONE WAVEFRONT PER PAIR, anti-diagonal ("systolic")
// order: lane l owns W consecutive columns of the DP matrix and at step t
// works on row t - l + 1, so the 64 lanes sit on one anti-diagonal band of the
// matrix.  A lane's rolling state (M-goe, Ix, matches/len for its W columns)
// stays in registers; the only cross-lane traffic is the last column of lane
// l-1 handed to lane l once per step (three 32-bit wave shifts).  sequence1
// sits in LDS (one byte per residue, per wave), the score table in LDS.
// Same int32 recurrence, boundary and tie-break as k_nw_short<.., false>
// (reference src/pairwiseSeqAlign.cpp:222-281), same forward-propagated
// (matches, length) instead of the traceback (:284-308).
constexpr int K4_THREADS = 256;
constexpr int K4_MAXLEN = 1024;   // 64 lanes x W <= 16 columns
constexpr int K4_TILE = 8;        // pairs per tile side; a wave takes 2 rows x 8 columns

template <int W>
__global__ __launch_bounds__(K4_THREADS) void k_nw_long(
    const uint8_t *__restrict__ codes, const int64_t *__restrict__ offsets, int64_t n, ScoreTable table,
    int32_t go, int32_t ge, int64_t row_begin, int64_t row_end, int symmetric, int kind,
    void *__restrict__ out_v, int64_t ld, int32_t *__restrict__ score_out, int64_t ld_score,
    int64_t ntiles, int T) {
  __shared__ __attribute__((aligned(16))) Cell tab[24 * 24];
  __shared__ uint8_t seq1[K4_THREADS / 64][K4_MAXLEN];

  const int64_t L = blockIdx.x;
  if (L >= ntiles) return;
  int ti, tj;
  bool allow_direct = true, allow_mirror = true;
  if (symmetric) {  // row-major over the upper triangle of 8x8-pair tiles
    const double Td = (double)T;
    int64_t t = (int64_t)(Td + 0.5 - sqrt((Td + 0.5) * (Td + 0.5) - 2.0 * (double)L));
    if (t < 0) t = 0;
    if (t > T - 1) t = T - 1;
    auto start = [&](int64_t r) { return r * T - r * (r - 1) / 2; };
    while (t > 0 && start(t) > L) --t;
    while (t + 1 <= T - 1 && start(t + 1) <= L) ++t;
    ti = (int)t;
    tj = (int)(t + (L - start(t)));
  } else {          // row block: see k_nw_short
    const int rt = (int)(row_begin / K4_TILE) + (int)(L / T);
    const int tc = (int)(L % T);
    ti = tc >= rt ? rt : tc;
    tj = tc >= rt ? tc : rt;
    allow_direct = tc >= rt;
    allow_mirror = tc <= rt;
  }
  const int64_t I0 = (int64_t)ti * K4_TILE, J0 = (int64_t)tj * K4_TILE;
  const int32_t goe = go + ge;
  const int32_t NEG = INT_MIN / 2;
  const int32_t ix_first = max(NEG - goe, NEG - ge);

  for (int e = threadIdx.x; e < 576; e += K4_THREADS) {
    const int a = e / 24, b = e - a * 24;
    tab[e].s_goe = (int32_t)table.s[e] + goe;
    tab[e].inc = 1u + ((a == b) ? 0x10000u : 0u);
  }
  __syncthreads();
  const char *tab_bytes = reinterpret_cast<const char *>(tab);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint8_t *s1 = seq1[wave];

  for (int ri = 0; ri < 2; ++ri) {
    const int64_t i = I0 + wave * 2 + ri;
    if (i >= n) break;
    const int64_t b1 = offsets[i];
    const int32_t m = (int32_t)(offsets[i + 1] - b1);
    for (int q = lane; q < m; q += 64) s1[q] = codes[b1 + q];   // same wave reads it back: program order suffices
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    for (int cj = 0; cj < K4_TILE; ++cj) {
      const int64_t j = J0 + cj;
      if (j >= n || j < i) continue;
      const bool do_direct = allow_direct && i >= row_begin && i < row_end;
      const bool do_mirror = allow_mirror && j != i && j >= row_begin && j < row_end;
      if (!do_direct && !do_mirror) continue;
      const int64_t b2 = offsets[j];
      const int32_t nn = (int32_t)(offsets[j + 1] - b2);

      uint32_t p_res = 0;       // matches<<16 | len of cell (m, nn)
      int32_t sc_res = NEG;
      if (m == 0 || nn == 0) {  // boundary cells (reference :222-235)
        p_res = (uint32_t)(m == 0 ? nn : m);
        sc_res = (m == 0 && nn == 0) ? 0 : NEG;
      } else {
        const int la = (nn + W - 1) / W;          // lanes that own at least one real column
        const int c_first = lane * W;             // global column index (0-based) of this lane's first column - 1
        uint32_t boff[W];
        int32_t MG[W], X[W];
        uint32_t P[W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
          const int c = c_first + w;              // column c+1
          boff[w] = (c < nn ? (uint32_t)codes[b2 + c] : 0u) * (uint32_t)sizeof(Cell);
          MG[w] = max(NEG, -go - c * ge) - goe;   // max(M,Ix,Iy)[0][c+1] - goe
          X[w] = NEG;
          P[w] = (uint32_t)(c + 1);
        }
        // what lane l-1 hands over: state of column c_first (its last column) at the row this lane
        // is about to process; "prev" = the same for the row before (the diagonal neighbour)
        int32_t mg_prev = (lane == 0) ? -goe : (max(NEG, -go - (c_first - 1) * ge) - goe);  // max(M,Ix,Iy)[0][c_first] - goe
        uint32_t p_prev = (uint32_t)c_first;                                                // (0 matches, length c_first)
        int32_t mg_send = 0, y_send = 0;
        uint32_t p_send = 0;
        const int own_lane = (nn - 1) / W, own_w = (nn - 1) - own_lane * W;
        const int steps = m + la - 1;
        for (int t = 0; t < steps; ++t) {
          // shift the last-column state one lane up (lane 0 takes the column-0 boundary, :224-229)
          int32_t mg_in = __shfl_up(mg_send, 1);
          int32_t y_in = __shfl_up(y_send, 1);
          uint32_t p_in = (uint32_t)__shfl_up((int)p_send, 1);
          const int r = t - lane + 1;             // DP row of this lane in this step
          if (lane == 0) { mg_in = NEG - goe; y_in = NEG; p_in = (uint32_t)r; }
          if (lane < la && r >= 1 && r <= m) {
            const uint32_t a = s1[r - 1];
            const char *tab_row = tab_bytes + a * (24u * (uint32_t)sizeof(Cell));
            int32_t y_out;
            if (r == 1)
              nw_row<W, true>(MG, X, P, boff, tab_row, mg_prev, p_prev, p_in, mg_in, y_in, ge, goe, ix_first, &y_out);
            else
              nw_row<W, false>(MG, X, P, boff, tab_row, mg_prev, p_prev, p_in, mg_in, y_in, ge, goe, ix_first, &y_out);
            // column c_first at row r becomes the diagonal neighbour of row r+1
            mg_prev = (lane == 0) ? (max(NEG, -go - (r - 1) * ge) - goe) : mg_in;   // lane 0: max(M,Ix,Iy)[r][0] = Ix[r][0]
            p_prev = p_in;
            mg_send = MG[W - 1];
            y_send = y_out;
            p_send = P[W - 1];
            if (r == m && lane == own_lane) {
#pragma unroll
              for (int w = 0; w < W; ++w)
                if (w == own_w) { p_res = P[w]; sc_res = MG[w] + goe; }
            }
          }
        }
        p_res = (uint32_t)__shfl((int)p_res, own_lane);
        sc_res = __shfl(sc_res, own_lane);
      }

      if (lane == 0) {
        const uint32_t mt = p_res >> 16, ln = p_res & 0xffffu;
        const int64_t od = (i - row_begin) * ld + j, om = (j - row_begin) * ld + i;
        if (kind == DA_OUT_F64) {
          const double v = nw_ratio(mt, ln);
          if (do_direct) reinterpret_cast<double *>(out_v)[od] = v;
          if (do_mirror) reinterpret_cast<double *>(out_v)[om] = v;
        } else if (kind == DA_OUT_PACK32) {
          if (do_direct) reinterpret_cast<uint32_t *>(out_v)[od] = p_res;
          if (do_mirror) reinterpret_cast<uint32_t *>(out_v)[om] = p_res;
        } else {
          const uint16_t v = (uint16_t)((mt << 8) | (ln & 0xffu));
          if (do_direct) reinterpret_cast<uint16_t *>(out_v)[od] = v;
          if (do_mirror) reinterpret_cast<uint16_t *>(out_v)[om] = v;
        }
        if (score_out) {
          if (do_direct) score_out[(i - row_begin) * ld_score + j] = sc_res;
          if (do_mirror) score_out[(j - row_begin) * ld_score + i] = sc_res;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();   // next row overwrites s1
  }
}

// ---------------------------------------------------------------- K5 --
// k_nw_xlong: sequences beyond k_nw_long's 64 lanes x 16 columns (the reference is O(m n) for ANY length,
// src/pairwiseSeqAlign.cpp:216-219).  Same systolic sweep, tiled along sequence2: column blocks of 1024; the last column of a
// block (M - goe, Iy, matches/length for every row) is spilled to HBM by lane 63 and read back by lane 0 as the left boundary
// of the next block -- the "rolling kernel with HBM spill" of SURVEY 8(f)-4.  sequence1 sits in dynamic LDS (any length the
// 160 KiB hold: 4 waves x max_len bytes), workgroups walk the tile list with a grid stride so the spill area is per resident
// wave.  Alignment length is carried in 16 bits like everywhere else, hence m + n <= 65535 (max_len <= 32767).
constexpr int K5_W = 16, K5_CB = 64 * K5_W;
constexpr int K5_MAXLEN = 32767;
__global__ __launch_bounds__(K4_THREADS) void k_nw_xlong(
    const uint8_t *__restrict__ codes, const int64_t *__restrict__ offsets, int64_t n, ScoreTable table,
    int32_t go, int32_t ge, int64_t row_begin, int64_t row_end, int symmetric, int kind,
    void *__restrict__ out_v, int64_t ld, int32_t *__restrict__ score_out, int64_t ld_score,
    int64_t ntiles, int T, int32_t *__restrict__ bnd, int64_t bnd_stride, int s1cap) {
  constexpr int W = K5_W;
  __shared__ __attribute__((aligned(16))) Cell tab[24 * 24];
  extern __shared__ uint8_t seq1_dyn[];
  const int32_t goe = go + ge;
  const int32_t NEG = INT_MIN / 2;
  const int32_t ix_first = max(NEG - goe, NEG - ge);
  for (int e = threadIdx.x; e < 576; e += K4_THREADS) {
    const int a = e / 24, b = e - a * 24;
    tab[e].s_goe = (int32_t)table.s[e] + goe;
    tab[e].inc = 1u + ((a == b) ? 0x10000u : 0u);
  }
  __syncthreads();
  const char *tab_bytes = reinterpret_cast<const char *>(tab);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint8_t *s1 = seq1_dyn + (size_t)wave * s1cap;
  int32_t *bm = bnd + ((int64_t)blockIdx.x * 4 + wave) * 3 * bnd_stride, *by = bm + bnd_stride, *bp = by + bnd_stride;

  for (int64_t L = blockIdx.x; L < ntiles; L += gridDim.x) {
    int ti, tj;
    bool allow_direct = true, allow_mirror = true;
    if (symmetric) {  // row-major over the upper triangle of 8x8-pair tiles
      const double Td = (double)T;
      int64_t t = (int64_t)(Td + 0.5 - sqrt((Td + 0.5) * (Td + 0.5) - 2.0 * (double)L));
      if (t < 0) t = 0;
      if (t > T - 1) t = T - 1;
      auto start = [&](int64_t r) { return r * T - r * (r - 1) / 2; };
      while (t > 0 && start(t) > L) --t;
      while (t + 1 <= T - 1 && start(t + 1) <= L) ++t;
      ti = (int)t;
      tj = (int)(t + (L - start(t)));
    } else {          // row block: see k_nw_short
      const int rt = (int)(row_begin / K4_TILE) + (int)(L / T);
      const int tc = (int)(L % T);
      ti = tc >= rt ? rt : tc;
      tj = tc >= rt ? tc : rt;
      allow_direct = tc >= rt;
      allow_mirror = tc <= rt;
    }
    const int64_t I0 = (int64_t)ti * K4_TILE, J0 = (int64_t)tj * K4_TILE;
    for (int ri = 0; ri < 2; ++ri) {
      const int64_t i = I0 + wave * 2 + ri;
      if (i >= n) break;
      const int64_t b1 = offsets[i];
      const int32_t m = (int32_t)(offsets[i + 1] - b1);
      for (int q = lane; q < m; q += 64) s1[q] = codes[b1 + q];   // same wave reads it back: program order suffices
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int cj = 0; cj < K4_TILE; ++cj) {
        const int64_t j = J0 + cj;
        if (j >= n || j < i) continue;
        const bool do_direct = allow_direct && i >= row_begin && i < row_end;
        const bool do_mirror = allow_mirror && j != i && j >= row_begin && j < row_end;
        if (!do_direct && !do_mirror) continue;
        const int64_t b2 = offsets[j];
        const int32_t nn = (int32_t)(offsets[j + 1] - b2);
        uint32_t p_res = 0;       // matches<<16 | len of cell (m, nn)
        int32_t sc_res = NEG;
        if (m == 0 || nn == 0) {  // boundary cells (reference :222-235)
          p_res = (uint32_t)(m == 0 ? nn : m);
          sc_res = (m == 0 && nn == 0) ? 0 : NEG;
        } else {
          const int nblk = (nn + K5_CB - 1) / K5_CB;
          int own_lane = 0;
          for (int cb = 0; cb < nblk; ++cb) {
            const int c0 = cb * K5_CB;
            const int ncol = (nn - c0 < K5_CB) ? (nn - c0) : K5_CB;
            const int la = (ncol + W - 1) / W;          // lanes that own at least one real column of this block
            const int c_first = c0 + lane * W;          // 0-based index of this lane's first column
            const bool last_blk = cb + 1 == nblk;
            uint32_t boff[W];
            int32_t MG[W], X[W];
            uint32_t P[W];
#pragma unroll
            for (int w = 0; w < W; ++w) {
              const int c = c_first + w;                // column c+1
              boff[w] = (c < nn ? (uint32_t)codes[b2 + c] : 0u) * (uint32_t)sizeof(Cell);
              MG[w] = max(NEG, -go - c * ge) - goe;     // max(M,Ix,Iy)[0][c+1] - goe
              X[w] = NEG;
              P[w] = (uint32_t)(c + 1);
            }
            int32_t mg_prev = (c_first == 0) ? -goe : (max(NEG, -go - (c_first - 1) * ge) - goe);  // max(M,Ix,Iy)[0][c_first] - goe
            uint32_t p_prev = (uint32_t)c_first;
            int32_t mg_send = 0, y_send = 0;
            uint32_t p_send = 0;
            own_lane = (ncol - 1) / W;
            const int own_w = (ncol - 1) - own_lane * W;
            const int steps = m + la - 1;
            for (int t = 0; t < steps; ++t) {
              int32_t mg_in = __shfl_up(mg_send, 1);
              int32_t y_in = __shfl_up(y_send, 1);
              uint32_t p_in = (uint32_t)__shfl_up((int)p_send, 1);
              const int r = t - lane + 1;             // DP row of this lane in this step
              if (lane == 0) {
                if (cb == 0) { mg_in = NEG - goe; y_in = NEG; p_in = (uint32_t)r; }        // column 0 (reference :224-229)
                else if (r >= 1 && r <= m) {                                                 // last column of the previous block
                  mg_in = __hip_atomic_load(bm + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  y_in = __hip_atomic_load(by + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  p_in = (uint32_t)__hip_atomic_load(bp + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
              }
              if (lane < la && r >= 1 && r <= m) {
                const uint32_t a = s1[r - 1];
                const char *tab_row = tab_bytes + a * (24u * (uint32_t)sizeof(Cell));
                int32_t y_out;
                if (r == 1)
                  nw_row<W, true>(MG, X, P, boff, tab_row, mg_prev, p_prev, p_in, mg_in, y_in, ge, goe, ix_first, &y_out);
                else
                  nw_row<W, false>(MG, X, P, boff, tab_row, mg_prev, p_prev, p_in, mg_in, y_in, ge, goe, ix_first, &y_out);
                // column c_first at row r becomes the diagonal neighbour of row r+1
                mg_prev = (c_first == 0) ? (max(NEG, -go - (r - 1) * ge) - goe) : mg_in;   // column 0: max(M,Ix,Iy)[r][0] = Ix[r][0]
                p_prev = p_in;
                mg_send = MG[W - 1];
                y_send = y_out;
                p_send = P[W - 1];
                if (!last_blk && lane == 63) {          // a full block: lane 63 owns its last column
                  __hip_atomic_store(bm + r, mg_send, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  __hip_atomic_store(by + r, y_send, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  __hip_atomic_store(bp + r, (int32_t)p_send, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (last_blk && r == m && lane == own_lane) {
#pragma unroll
                  for (int w = 0; w < W; ++w)
                    if (w == own_w) { p_res = P[w]; sc_res = MG[w] + goe; }
                }
              }
            }
            if (!last_blk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the spilled column is complete before lane 0 reads it
          }
          p_res = (uint32_t)__shfl((int)p_res, own_lane);
          sc_res = __shfl(sc_res, own_lane);
        }
        if (lane == 0) {
          const uint32_t mt = p_res >> 16, ln = p_res & 0xffffu;
          const int64_t od = (i - row_begin) * ld + j, om = (j - row_begin) * ld + i;
          if (kind == DA_OUT_F64) {
            const double v = nw_ratio(mt, ln);
            if (do_direct) reinterpret_cast<double *>(out_v)[od] = v;
            if (do_mirror) reinterpret_cast<double *>(out_v)[om] = v;
          } else if (kind == DA_OUT_PACK32) {
            if (do_direct) reinterpret_cast<uint32_t *>(out_v)[od] = p_res;
            if (do_mirror) reinterpret_cast<uint32_t *>(out_v)[om] = p_res;
          }
          if (score_out) {
            if (do_direct) score_out[(i - row_begin) * ld_score + j] = sc_res;
            if (do_mirror) score_out[(j - row_begin) * ld_score + i] = sc_res;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();   // next row overwrites s1
    }
  }
}

// ---------------------------------------------------------------- duplicate sequences --
// Byte-identical sequences have identical rows and columns in the result, so the DP only has to run on the table of
// UNIQUE sequences and the N x N matrix is an index expansion of it (same spirit as the MinHash dictionary; exact).
// One subtlety: calculate_similarity is not symmetric and the reference evaluates calc(seq[i], seq[j])
// for i < j, so for two different unique strings A, B both calc(A, B) and calc(B, A) can be needed -- the first when some
// copy of A precedes some copy of B: first(A) < last(B).  The unique table is therefore compared as an ORDERED square
// and only where that condition can hold (k_nw_short's ordered mode); multi-copy strings are numbered before single-copy
// ones (each group in order of first occurrence), which makes the unneeded region whole tiles.
// h3n2-like 100k: 49 k unique strings, 3.5x fewer DP pairs; uniform peptides: nothing to collapse, the plan says so and
// the direct kernel runs.
constexpr uint32_t DD_EMPTY = 0xffffffffu;

__device__ __forceinline__ uint32_t dd_hash(const uint8_t *c, int64_t b, int32_t len) {
  uint32_t h = 0x9747b28cu ^ (uint32_t)len;
  for (int32_t q = 0; q < len; ++q) { h ^= c[b + q]; h *= 0x01000193u; h ^= h >> 15; }
  h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
  return h;
}
__device__ __forceinline__ bool dd_same(const uint8_t *c, const int64_t *off, int64_t a, int64_t b) {
  const int64_t ba = off[a], bb = off[b];
  const int64_t la = off[a + 1] - ba;
  if (la != off[b + 1] - bb) return false;
  for (int64_t q = 0; q < la; ++q)
    if (c[ba + q] != c[bb + q]) return false;
  return true;
}
// open-addressing table of sequence indices keyed by the sequence's bytes; a slot ends up holding the SMALLEST index of
// its string (= first occurrence).  Exact: keys are compared byte for byte, the hash only picks the start slot.
__global__ __launch_bounds__(256) void k_dd_insert(const uint8_t *__restrict__ codes, const int64_t *__restrict__ off, int32_t n,
                                                   uint32_t *__restrict__ table, uint32_t mask) {
  const int32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t slot = dd_hash(codes, off[i], (int32_t)(off[i + 1] - off[i])) & mask;
  for (;;) {
    uint32_t o = table[slot];
    if (o == DD_EMPTY) {
      o = atomicCAS(&table[slot], DD_EMPTY, (uint32_t)i);
      if (o == DD_EMPTY) return;
    }
    if (dd_same(codes, off, (int64_t)o, i)) { atomicMin(&table[slot], (uint32_t)i); return; }
    slot = (slot + 1) & mask;
  }
}
__global__ __launch_bounds__(256) void k_dd_lookup(const uint8_t *__restrict__ codes, const int64_t *__restrict__ off, int32_t n,
                                                   const uint32_t *__restrict__ table, uint32_t mask, int32_t *__restrict__ rep,
                                                   int32_t *__restrict__ mult, int32_t *__restrict__ last) {
  const int32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t slot = dd_hash(codes, off[i], (int32_t)(off[i + 1] - off[i])) & mask;
  for (;;) {
    const uint32_t o = table[slot];
    if (dd_same(codes, off, (int64_t)o, i)) {
      rep[i] = (int32_t)o;
      atomicAdd(&mult[o], 1);
      atomicMax(&last[o], i);
      return;
    }
    slot = (slot + 1) & mask;
  }
}
__global__ __launch_bounds__(256) void k_dd_flags(const int32_t *__restrict__ rep, const int32_t *__restrict__ mult, int32_t n,
                                                  int32_t *__restrict__ fm, int32_t *__restrict__ fs) {
  const int32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const bool is_rep = rep[i] == i;
  fm[i] = (is_rep && mult[i] > 1) ? 1 : 0;
  fs[i] = (is_rep && mult[i] == 1) ? 1 : 0;
}
// exclusive prefix sum of n int32 by ONE workgroup (n is ~1e5: not worth more); out[n] = total (int64 output type OUT)
template <typename OUT>
__global__ __launch_bounds__(1024) void k_dd_scan(const int32_t *__restrict__ in, OUT *__restrict__ out, int32_t n) {
  __shared__ int64_t wsum[16];
  __shared__ int64_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int32_t base = 0; base < n; base += 1024) {
    const int32_t i = base + tid;
    const int64_t v = i < n ? in[i] : 0;
    int64_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int64_t y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int64_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int64_t carry = carry_s;
    if (i < n) out[i] = (OUT)(carry + woff + x - v);
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + x;
    __syncthreads();
  }
  if (tid == 0) out[n] = (OUT)carry_s;
}
// the two flag arrays of the plan (multi-copy / single-copy representatives) scanned in ONE pass of one workgroup: the counts ride in
// the two halves of a 64-bit word (n < 2^31), so the cost is that of a single scan (0.14 ms at n = 100k, twice that as two launches)
__global__ __launch_bounds__(1024) void k_dd_scan_pair(const int32_t *__restrict__ fm, const int32_t *__restrict__ fs, int32_t *__restrict__ pm,
                                                       int32_t *__restrict__ ps, int32_t n) {
  __shared__ uint64_t wsum[16];
  __shared__ uint64_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int32_t base = 0; base < n; base += 1024) {
    const int32_t i = base + tid;
    const uint64_t v = i < n ? ((uint64_t)(uint32_t)fm[i] | ((uint64_t)(uint32_t)fs[i] << 32)) : 0;
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
    if (i < n) {
      const uint64_t e = carry + woff + x - v;
      pm[i] = (int32_t)(uint32_t)e;
      ps[i] = (int32_t)(uint32_t)(e >> 32);
    }
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + x;
    __syncthreads();
  }
  if (tid == 0) { pm[n] = (int32_t)(uint32_t)carry_s; ps[n] = (int32_t)(uint32_t)(carry_s >> 32); }
}
// The same two scans over many workgroups (n >= 8192): block sums, a scan of the block sums by one workgroup, then every block's own exclusive scan plus
// its offset -- three short launches (~0.03 ms) instead of one workgroup walking the whole array (0.15 ms at n = 100k: the plan sits in front of everything).
// MODE 0: (fm, fs) -> (pm, ps) packed in 64 bits;  MODE 1: in (int32) -> out (int64)
template <int MODE>
__device__ __forceinline__ uint64_t dd_scan_load(const int32_t *a, const int32_t *b, int32_t i) {
  if (MODE == 0) return (uint64_t)(uint32_t)a[i] | ((uint64_t)(uint32_t)b[i] << 32);
  return (uint64_t)(int64_t)a[i];
}
template <int MODE>
__global__ __launch_bounds__(1024) void k_dd_scan_sums(const int32_t *__restrict__ a, const int32_t *__restrict__ b, int32_t n, uint64_t *__restrict__ bsum) {
  __shared__ uint64_t wsum[16];
  const int32_t i = blockIdx.x * 1024 + threadIdx.x;
  uint64_t x = i < n ? dd_scan_load<MODE>(a, b, i) : 0;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = x;
  __syncthreads();
  if (threadIdx.x == 0) { uint64_t t = 0; for (int w = 0; w < 16; ++w) t += wsum[w]; bsum[blockIdx.x] = t; }
}
__global__ __launch_bounds__(1024) void k_dd_scan_offsets(uint64_t *__restrict__ bsum, int32_t nblocks) {   // in place: exclusive scan; bsum[nblocks] = total
  __shared__ uint64_t wsum[16];
  __shared__ uint64_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int32_t base = 0; base < nblocks; base += 1024) {
    const int32_t i = base + tid;
    const uint64_t v = i < nblocks ? bsum[i] : 0;
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
    if (i < nblocks) bsum[i] = carry + woff + x - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + x;
    __syncthreads();
  }
  if (tid == 0) bsum[nblocks] = carry_s;
}
template <int MODE>
__global__ __launch_bounds__(1024) void k_dd_scan_apply(const int32_t *__restrict__ a, const int32_t *__restrict__ b, int32_t n, const uint64_t *__restrict__ bsum,
                                                        int32_t *__restrict__ o0, int32_t *__restrict__ o1, int64_t *__restrict__ o64) {
  __shared__ uint64_t wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int32_t i = blockIdx.x * 1024 + tid;
  const uint64_t v = i < n ? dd_scan_load<MODE>(a, b, i) : 0;
  uint64_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  uint64_t woff = bsum[blockIdx.x];
  for (int w = 0; w < wave; ++w) woff += wsum[w];
  const uint64_t e = woff + x - v;
  if (i < n) {
    if (MODE == 0) { o0[i] = (int32_t)(uint32_t)e; o1[i] = (int32_t)(uint32_t)(e >> 32); }
    else o64[i] = (int64_t)e;
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) {
    const uint64_t t = bsum[gridDim.x];
    if (MODE == 0) { o0[n] = (int32_t)(uint32_t)t; o1[n] = (int32_t)(uint32_t)(t >> 32); }
    else o64[n] = (int64_t)t;
  }
}
__global__ __launch_bounds__(256) void k_dd_assign(const int32_t *__restrict__ rep, const int32_t *__restrict__ mult,
                                                   const int32_t *__restrict__ last, const int32_t *__restrict__ pm,
                                                   const int32_t *__restrict__ ps, const int64_t *__restrict__ off, int32_t n,
                                                   int32_t *__restrict__ uid_of, int32_t *__restrict__ ufirst,
                                                   int32_t *__restrict__ ulast, int32_t *__restrict__ ulen, int first_order) {
  const int32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || rep[i] != i) return;
  const int32_t M = pm[n];
  // first_order: one numbering by first occurrence (MinHash's pipelined route: rows [0, R) of the input only use ids < pm[R] + ps[R])
  const int32_t uid = first_order ? pm[i] + ps[i] : (mult[i] > 1 ? pm[i] : M + ps[i]);
  uid_of[i] = uid;
  ufirst[uid] = i;
  ulast[uid] = last[i];
  ulen[uid] = (int32_t)(off[i + 1] - off[i]);
}
__global__ __launch_bounds__(256) void k_dd_map(const int32_t *__restrict__ rep, const int32_t *__restrict__ uid_of, int32_t n,
                                                int32_t *__restrict__ uidx) {
  const int32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) uidx[i] = uid_of[rep[i]];
}
__global__ __launch_bounds__(256) void k_dd_gather(const uint8_t *__restrict__ codes, const int64_t *__restrict__ off,
                                                   const int32_t *__restrict__ ufirst, const int64_t *__restrict__ uoff, int32_t U,
                                                   uint8_t *__restrict__ ucodes) {
  const int32_t u = blockIdx.x * 256 + threadIdx.x;
  if (u >= U) return;
  const int64_t b = off[ufirst[u]], d = uoff[u], len = uoff[u + 1] - d;
  for (int64_t q = 0; q < len; ++q) ucodes[d + q] = codes[b + q];
}
// per 64-row tile block of the unique table: smallest first occurrence, largest last occurrence
__global__ __launch_bounds__(64) void k_dd_blocks(const int32_t *__restrict__ ufirst, const int32_t *__restrict__ ulast, int32_t U,
                                                  int32_t *__restrict__ minfirst, int32_t *__restrict__ maxlast) {
  const int32_t u = blockIdx.x * 64 + threadIdx.x;
  int32_t f = u < U ? ufirst[u] : 0x7fffffff, l = u < U ? ulast[u] : -1;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    f = min(f, __shfl_xor(f, d, 64));
    l = max(l, __shfl_xor(l, d, 64));
  }
  if (threadIdx.x == 0) { minfirst[blockIdx.x] = f; maxlast[blockIdx.x] = l; }
}

}  // namespace

// Layout of the dedupe plan inside one workspace (all device memory; see nw_dedup_workspace_bytes)
NwDedupPlan nw_dedup_layout(void *work, int64_t n, int64_t total) {
  NwDedupPlan p;
  char *w = static_cast<char *>(work);
  auto take = [&](size_t bytes) { char *r = w; w += (bytes + 255) / 256 * 256; return r; };
  uint32_t ts = 1024;
  while ((int64_t)ts < 2 * n) ts <<= 1;
  p.table_size = ts;
  p.table = reinterpret_cast<uint32_t *>(take((size_t)ts * 4));
  int32_t **arrs[] = {&p.rep, &p.mult, &p.last, &p.fm, &p.fs, &p.uid_of, &p.uidx, &p.ufirst, &p.ulast, &p.ulen};
  for (auto a : arrs) *a = reinterpret_cast<int32_t *>(take((size_t)(n + 1) * 4));
  p.pm = reinterpret_cast<int32_t *>(take((size_t)(n + 1) * 4));
  p.ps = reinterpret_cast<int32_t *>(take((size_t)(n + 1) * 4));
  p.uoff = reinterpret_cast<int64_t *>(take((size_t)(n + 1) * 8));
  p.ucodes = reinterpret_cast<uint8_t *>(take((size_t)(total > 0 ? total : 1)));
  p.minfirst = reinterpret_cast<int32_t *>(take((size_t)(n / 64 + 2) * 4));
  p.maxlast = reinterpret_cast<int32_t *>(take((size_t)(n / 64 + 2) * 4));
  p.bytes = (size_t)(w - static_cast<char *>(work));
  return p;
}
size_t nw_dedup_workspace_bytes(int64_t n, int64_t total) { return nw_dedup_layout(nullptr, n, total).bytes + 256; }

// Stage 1 (stream-ordered): everything up to the counts M (multi-copy strings) = pm[n] and S (single-copy) = ps[n].
int launch_nw_dedup_count(const uint8_t *d_codes, const int64_t *d_off, int64_t n, const NwDedupPlan &p, hipStream_t stream) {
  if (n <= 0 || n > 0x7ffffff0LL) return fail(DA_ERR_UNSUPPORTED, "dedupe plan: bad n");
  const unsigned nb = (unsigned)ceil_div(n, 256);
  DA_HIP_TRY(hipMemsetAsync(p.table, 0xff, (size_t)p.table_size * 4, stream));
  DA_HIP_TRY(hipMemsetAsync(p.mult, 0, (size_t)n * 4, stream));
  DA_HIP_TRY(hipMemsetAsync(p.last, 0, (size_t)n * 4, stream));
  hipLaunchKernelGGL(k_dd_insert, dim3(nb), dim3(256), 0, stream, d_codes, d_off, (int32_t)n, p.table, p.table_size - 1);
  hipLaunchKernelGGL(k_dd_lookup, dim3(nb), dim3(256), 0, stream, d_codes, d_off, (int32_t)n, p.table, p.table_size - 1, p.rep, p.mult, p.last);
  hipLaunchKernelGGL(k_dd_flags, dim3(nb), dim3(256), 0, stream, p.rep, p.mult, (int32_t)n, p.fm, p.fs);
  const bool wide_scan = true;
  if (n >= 8192 && wide_scan) {     // (the hash table is free again after k_dd_lookup: its words hold the block sums)
    const int32_t nblk = (int32_t)ceil_div(n, 1024);
    uint64_t *bsum = reinterpret_cast<uint64_t *>(p.table);
    hipLaunchKernelGGL(k_dd_scan_sums<0>, dim3((unsigned)nblk), dim3(1024), 0, stream, p.fm, p.fs, (int32_t)n, bsum);
    hipLaunchKernelGGL(k_dd_scan_offsets, dim3(1), dim3(1024), 0, stream, bsum, nblk);
    hipLaunchKernelGGL(k_dd_scan_apply<0>, dim3((unsigned)nblk), dim3(1024), 0, stream, p.fm, p.fs, (int32_t)n, bsum, p.pm, p.ps, (int64_t *)nullptr);
  } else
    hipLaunchKernelGGL(k_dd_scan_pair, dim3(1), dim3(1024), 0, stream, p.fm, p.fs, p.pm, p.ps, (int32_t)n);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}
// Stage 2 (stream-ordered, U = M + S known to the host): unique ids, the unique table's codes / offsets, tile-block bounds.
int launch_nw_dedup_build(const uint8_t *d_codes, const int64_t *d_off, int64_t n, int64_t U, const NwDedupPlan &p, hipStream_t stream,
                          bool first_order) {
  const unsigned nb = (unsigned)ceil_div(n, 256);
  hipLaunchKernelGGL(k_dd_assign, dim3(nb), dim3(256), 0, stream, p.rep, p.mult, p.last, p.pm, p.ps, d_off, (int32_t)n, p.uid_of,
                     p.ufirst, p.ulast, p.ulen, first_order ? 1 : 0);
  hipLaunchKernelGGL(k_dd_map, dim3(nb), dim3(256), 0, stream, p.rep, p.uid_of, (int32_t)n, p.uidx);
  if (U >= 8192) {
    const int32_t nblk = (int32_t)ceil_div(U, 1024);
    uint64_t *bsum = reinterpret_cast<uint64_t *>(p.table);
    hipLaunchKernelGGL(k_dd_scan_sums<1>, dim3((unsigned)nblk), dim3(1024), 0, stream, p.ulen, (const int32_t *)nullptr, (int32_t)U, bsum);
    hipLaunchKernelGGL(k_dd_scan_offsets, dim3(1), dim3(1024), 0, stream, bsum, nblk);
    hipLaunchKernelGGL(k_dd_scan_apply<1>, dim3((unsigned)nblk), dim3(1024), 0, stream, p.ulen, (const int32_t *)nullptr, (int32_t)U, bsum, (int32_t *)nullptr,
                       (int32_t *)nullptr, p.uoff);
  } else
    hipLaunchKernelGGL(k_dd_scan<int64_t>, dim3(1), dim3(1024), 0, stream, p.ulen, p.uoff, (int32_t)U);
  hipLaunchKernelGGL(k_dd_gather, dim3((unsigned)ceil_div(U, 256)), dim3(256), 0, stream, d_codes, d_off, p.ufirst, p.uoff, (int32_t)U, p.ucodes);
  hipLaunchKernelGGL(k_dd_blocks, dim3((unsigned)ceil_div(U, 64)), dim3(64), 0, stream, p.ufirst, p.ulast, (int32_t)U, p.minfirst, p.maxlast);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

int launch_nw_encode(const uint8_t *d_res, int64_t total, uint8_t *d_codes, int32_t *d_bad,
                     hipStream_t stream) {
  if (total <= 0) return DA_OK;
  int64_t blocks = ceil_div(total, 256);
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(k_nw_encode, dim3((unsigned)blocks), dim3(256), 0, stream, d_res, total, d_codes, d_bad);
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

const signed char *matrix_table_host(int id) {
  return (id >= 0 && id < DA_NUM_MATRICES) ? da_matrix_data[id] : nullptr;
}
const char *matrix_name_host(int id) {
  return (id >= 0 && id < DA_NUM_MATRICES) ? da_matrix_names[id] : nullptr;
}
int matrix_count_host() { return DA_NUM_MATRICES; }

int launch_nw(const uint8_t *d_codes, const int64_t *d_off, int64_t n, int64_t max_len,
              int matrix_id, int gap_open, int gap_ext, int64_t row_begin, int64_t row_end,
              bool symmetric, int kind, void *d_out, int64_t ld, int32_t *d_score,
              int64_t ld_score, hipStream_t stream, int shard_rank, int shard_world,
              const int32_t *ord_first, const int32_t *ord_minfirst, const int32_t *ord_maxlast,
              const int32_t *ord_perm, const uint8_t *ord_lcp) {
  if (n <= 0 || row_end <= row_begin) return DA_OK;
  const signed char *tab = matrix_table_host(matrix_id);
  if (!tab) return fail(DA_ERR_BAD_ARG, "matrix id %d out of range", matrix_id);
  if (max_len > K5_MAXLEN)
    return fail(DA_ERR_UNSUPPORTED,
                "similarityNW on gfx950 carries the alignment length in 16 bits: sequences up to %d residues (longest here: %lld)",
                K5_MAXLEN, (long long)max_len);
  if (max_len > K4_MAXLEN) {   // column-blocked sweep with the block boundary spilled to HBM (k_nw_xlong); synchronises the stream
    if (ord_first || shard_world > 0) return fail(DA_ERR_UNSUPPORTED, "sequences > %d residues: dense / row-block output only", K4_MAXLEN);
    if (kind == DA_OUT_COMPACT) return fail(DA_ERR_UNSUPPORTED, "uint16 NW output needs alignment length <= 255 (use the float64 or 32-bit packed kind)");
    ScoreTable st5;
    for (int e = 0; e < 576; ++e) st5.s[e] = tab[e];
    const int T8 = (int)ceil_div(n, K4_TILE);
    int64_t nt;
    if (symmetric) nt = (int64_t)T8 * (T8 + 1) / 2;
    else nt = ((row_end - 1) / K4_TILE - row_begin / K4_TILE + 1) * (int64_t)T8;
    const int s1cap = (int)ceil_div(max_len, 64) * 64;
    const size_t dyn = (size_t)4 * s1cap;
    const unsigned grid5 = (unsigned)std::min<int64_t>(nt, 1024);
    const int64_t bnd_stride = max_len + 1;
    int32_t *d_bnd = nullptr;
    DA_HIP_TRY(hipMalloc(&d_bnd, (size_t)grid5 * 4 * 3 * (size_t)bnd_stride * sizeof(int32_t)));
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_nw_xlong), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_nw_xlong, dim3(grid5), dim3(K4_THREADS), dyn, stream, d_codes, d_off, n, st5, (int32_t)gap_open, (int32_t)gap_ext,
                         row_begin, row_end, symmetric ? 1 : 0, kind, d_out, ld, d_score, ld_score, nt, T8, d_bnd, bnd_stride, s1cap);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_bnd);
    if (e != hipSuccess) return fail(DA_ERR_HIP, "k_nw_xlong failed: %s", hipGetErrorString(e));
    return DA_OK;
  }
  if (kind == DA_OUT_COMPACT && max_len > 127)
    return fail(DA_ERR_UNSUPPORTED, "uint16 NW output needs alignment length <= 255 (use the float64 or 32-bit packed kind)");
  // fast path: scores fit the score field of the combined key (see nw_row_ck / CKBits)
  const int64_t gap_span = (int64_t)gap_open + 2 * (max_len <= 32 ? 32 : 64) * (int64_t)gap_ext;
  const bool ck = gap_open >= 0 && gap_ext >= 0 && gap_span <= (max_len <= 32 ? 7000 : 2500) && !config().nw_int32;
  // <= 32 residues: register-resident lane-per-pair kernel (either cell update);
  // 33..64: the same kernel with the combined key only (the int32 form would need > 256 VGPRs);
  // otherwise the wavefront-per-pair anti-diagonal kernel
  if (max_len > 64 || (max_len > 32 && !ck)) {
    if (ord_first) return fail(DA_ERR_UNSUPPORTED, "ordered (deduplicated) NW is built for the lane-per-pair kernel (<= 64 residues)");
    if (shard_world > 0) return fail(DA_ERR_UNSUPPORTED, "row-sharded NW is built for sequences up to 64 residues (default-range gap penalties)");
    ScoreTable st4;
    for (int e = 0; e < 576; ++e) st4.s[e] = tab[e];
    const int T8 = (int)ceil_div(n, K4_TILE);
    int64_t nt;
    if (symmetric) nt = (int64_t)T8 * (T8 + 1) / 2;
    else nt = ((row_end - 1) / K4_TILE - row_begin / K4_TILE + 1) * (int64_t)T8;
    if (nt > 0x7fffffffLL) return fail(DA_ERR_UNSUPPORTED, "pair space too large for one launch");
    dim3 grid4((unsigned)nt), block4(K4_THREADS);
#define DA_K4(WW)                                                                                              \
  hipLaunchKernelGGL(k_nw_long<WW>, grid4, block4, 0, stream, d_codes, d_off, n, st4, (int32_t)gap_open,      \
                     (int32_t)gap_ext, row_begin, row_end, symmetric ? 1 : 0, kind, d_out, ld, d_score, ld_score, nt, T8)
    const int wneed = (int)ceil_div(max_len, 64);   // columns per lane so that 64 lanes cover the longest sequence
    if (wneed <= 1) DA_K4(1);
    else if (wneed <= 2) DA_K4(2);
    else if (wneed <= 3) DA_K4(3);
    else if (wneed <= 4) DA_K4(4);
    else if (wneed <= 6) DA_K4(6);
    else if (wneed <= 8) DA_K4(8);
    else if (wneed <= 9) DA_K4(9);
    else if (wneed <= 12) DA_K4(12);
    else DA_K4(16);
#undef DA_K4
    DA_HIP_TRY(hipGetLastError());
    return DA_OK;
  }
  ScoreTable st;
  for (int e = 0; e < 576; ++e) st.s[e] = tab[e];
  const int T = (int)ceil_div(n, K3_TILE);
  const ShardGeom sg = shard_geom(n, shard_world > 0 ? shard_world : 1, 128);
  const int fold_q = shard_world > 0 ? sg.Q : 0;
  const int64_t fold_w = sg.W;
  int64_t ntiles;
  if (ord_first) {
    if (row_begin % K3_TILE != 0 || (row_end % K3_TILE != 0 && row_end != n))
      return fail(DA_ERR_BAD_ARG, "ordered NW: the row range must consist of whole 64-row tile rows");
    if (shard_world > 0) ntiles = 2 * (int64_t)sg.Q * T;            // a rank's Q cyclic units (row_begin = 0, row_end = n)
    else ntiles = ((row_end - 1) / K3_TILE - row_begin / K3_TILE + 1) * (int64_t)T;
  }
  else if (symmetric) ntiles = (int64_t)T * (T + 1) / 2;
  else if (shard_world > 0) ntiles = 2 * (int64_t)sg.Q * T;          // 2 tile rows per 128-row unit, Q units per rank
  else ntiles = ((row_end - 1) / K3_TILE - row_begin / K3_TILE + 1) * (int64_t)T;
  if (ntiles > 0x7fffffffLL) return fail(DA_ERR_UNSUPPORTED, "pair space too large for one launch");
  dim3 grid((unsigned)ntiles), block(K3_THREADS);
#define DA_K3_ARGS d_codes, d_off, n, st, (int32_t)gap_open, (int32_t)gap_ext, row_begin, row_end, symmetric ? 1 : 0, kind, d_out, ld, \
                   d_score, ld_score, ntiles, T, shard_rank, shard_world, fold_q, fold_w, ord_first, ord_minfirst, ord_maxlast, ord_perm, ord_lcp
  // ordered mode on the whole table with the rows' sorted order given (launch_nw_sort_unique): prefix sharing (k_nw_short<.., PFX>)
  const bool pfx = ord_first && ord_perm && ord_lcp && ck && shard_world == 0 && row_begin == 0 && row_end == n && max_len <= 20 && gap_open <= 255;   // (NMAX = 24: 135 VGPRs, three waves; gap_open: the checkpoint's byte deltas)
#ifdef DA_K2_EXPERIMENTS
  const bool asm_rows = ck && getenv("DYNAALIGN_NW_ASM");     // experiment library only: the generated rows (tools/gen_nw_asm.py)
#else
  const bool asm_rows = false;
#endif
#define DA_K3(NM)                                                                                                     \
  do {                                                                                                                \
    if constexpr (NM <= 20) {                                                                                         \
      if (pfx) {                                                                                                      \
        hipLaunchKernelGGL((k_nw_short<NM, true, true, false, true>), grid, block, 0, stream, DA_K3_ARGS);            \
        break;                                                                                                        \
      }                                                                                                               \
    }                                                                                                                 \
    if constexpr (nw_has_asm_rows<NM>()) {                                                                            \
      if (asm_rows) {                                                                                                 \
        if (ord_first) hipLaunchKernelGGL((k_nw_short<NM, true, true, true>), grid, block, 0, stream, DA_K3_ARGS);    \
        else hipLaunchKernelGGL((k_nw_short<NM, true, false, true>), grid, block, 0, stream, DA_K3_ARGS);             \
        break;                                                                                                        \
      }                                                                                                               \
    }                                                                                                                 \
    if (ord_first) {                                                                                                  \
      if (ck) hipLaunchKernelGGL((k_nw_short<NM, true, true>), grid, block, 0, stream, DA_K3_ARGS);                   \
      else hipLaunchKernelGGL((k_nw_short<NM, false, true>), grid, block, 0, stream, DA_K3_ARGS);                     \
    } else {                                                                                                          \
      if (ck) hipLaunchKernelGGL((k_nw_short<NM, true, false>), grid, block, 0, stream, DA_K3_ARGS);                  \
      else hipLaunchKernelGGL((k_nw_short<NM, false, false>), grid, block, 0, stream, DA_K3_ARGS);                    \
    }                                                                                                                 \
  } while (0)
  if (max_len <= 8) DA_K3(8);
  else if (max_len <= 12) DA_K3(12);
  else if (max_len <= 16) DA_K3(16);
  else if (max_len <= 20) DA_K3(20);
  else if (max_len <= 24) DA_K3(24);
  else if (max_len <= 32) DA_K3(32);
  else {
#define DA_K3CK(NM)                                                                                                   \
  do {                                                                                                                \
    if (ord_first) hipLaunchKernelGGL((k_nw_short<NM, true, true>), grid, block, 0, stream, DA_K3_ARGS);              \
    else hipLaunchKernelGGL((k_nw_short<NM, true, false>), grid, block, 0, stream, DA_K3_ARGS);                       \
  } while (0)
    if (max_len <= 48) DA_K3CK(48); else DA_K3CK(64);
#undef DA_K3CK
  }
#undef DA_K3
#undef DA_K3_ARGS
  DA_HIP_TRY(hipGetLastError());
  return DA_OK;
}

// ---- sorted order of the unique strings for the ordered DP's prefix sharing (k_nw_short<.., PFX>) -----------------------------------------
// Residue codes 0..23 of up to 24 residues as two 60-bit keys (5 bits per position: code + 1, 0 = past the end, most significant first), two
// stable radix sorts (low key, then high key): perm[pos] = unique id at sorted position pos; lcp[pos] = common prefix of the strings at
// pos - 1 and pos (0 for pos = 0).  Any permutation is valid for the kernel (the order only decides how much is shared).
__global__ __launch_bounds__(256) void k_nw_sort_keys(const uint8_t *__restrict__ codes, const int64_t *__restrict__ off, int n, uint64_t *__restrict__ k_hi,
                                                      uint64_t *__restrict__ k_lo, int32_t *__restrict__ idx) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= n) return;
  const int64_t b = off[u];
  const int len = (int)(off[u + 1] - b);
  uint64_t hi = 0, lo = 0;
  for (int q = 0; q < 12; ++q) {
    hi = (hi << 5) | (uint64_t)(q < len ? codes[b + q] + 1 : 0);
    lo = (lo << 5) | (uint64_t)(q + 12 < len ? codes[b + q + 12] + 1 : 0);
  }
  k_hi[u] = hi; k_lo[u] = lo; idx[u] = u;
}
__global__ __launch_bounds__(256) void k_nw_gather_keys(const uint64_t *__restrict__ k_hi, const int32_t *__restrict__ idx, int n, uint64_t *__restrict__ out) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u < n) out[u] = k_hi[idx[u]];
}
__global__ __launch_bounds__(256) void k_nw_lcp(const uint8_t *__restrict__ codes, const int64_t *__restrict__ off, const int32_t *__restrict__ perm, int n,
                                                uint8_t *__restrict__ lcp) {
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= n) return;
  int l = 0;
  if (pos > 0) {
    const int a = perm[pos - 1], c = perm[pos];
    const int64_t ba = off[a], bc = off[c];
    const int la = (int)(off[a + 1] - ba), lc = (int)(off[c + 1] - bc), m = la < lc ? la : lc;
    while (l < m && l < 255 && codes[ba + l] == codes[bc + l]) ++l;
  }
  lcp[pos] = (uint8_t)l;
}
size_t nw_sort_unique_workspace_bytes(int64_t n) {
  size_t temp = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, temp, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const int32_t *)nullptr, (int32_t *)nullptr, (int)n);
  // perm[n], idx[n], idx2[n] (int32), lcp[n] (bytes), four key arrays (uint64), the sort's temporary storage
  return (size_t)n * (3 * 4 + 1 + 4 * 8) + temp + 4096;
}
int launch_nw_sort_unique(const uint8_t *d_codes, const int64_t *d_off, int64_t n, void *d_work, size_t work_bytes, const int32_t **perm_out,
                          const uint8_t **lcp_out, hipStream_t stream) {
  if (n < 1 || n > 0x7ffffff0LL || work_bytes < nw_sort_unique_workspace_bytes(n)) return fail(DA_ERR_BAD_ARG, "nw sort: workspace too small");
  auto align = [](uintptr_t x) { return (x + 255) & ~(uintptr_t)255; };
  uintptr_t w = align(reinterpret_cast<uintptr_t>(d_work));
  uint64_t *k_hi = reinterpret_cast<uint64_t *>(w); w = align(w + (size_t)n * 8);
  uint64_t *k_lo = reinterpret_cast<uint64_t *>(w); w = align(w + (size_t)n * 8);
  uint64_t *k_a = reinterpret_cast<uint64_t *>(w); w = align(w + (size_t)n * 8);
  uint64_t *k_b = reinterpret_cast<uint64_t *>(w); w = align(w + (size_t)n * 8);
  int32_t *idx = reinterpret_cast<int32_t *>(w); w = align(w + (size_t)n * 4);
  int32_t *idx2 = reinterpret_cast<int32_t *>(w); w = align(w + (size_t)n * 4);
  int32_t *perm = reinterpret_cast<int32_t *>(w); w = align(w + (size_t)n * 4);
  uint8_t *lcp = reinterpret_cast<uint8_t *>(w); w = align(w + (size_t)n);
  void *temp = reinterpret_cast<void *>(w);
  size_t temp_bytes = reinterpret_cast<uintptr_t>(d_work) + work_bytes - w;
  const unsigned nb = (unsigned)ceil_div(n, 256);
  hipLaunchKernelGGL(k_nw_sort_keys, dim3(nb), dim3(256), 0, stream, d_codes, d_off, (int)n, k_hi, k_lo, idx);
  DA_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, k_lo, k_a, idx, idx2, (int)n, 0, 60, stream));      // by residues 12..23
  hipLaunchKernelGGL(k_nw_gather_keys, dim3(nb), dim3(256), 0, stream, k_hi, idx2, (int)n, k_b);
  DA_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, k_b, k_a, idx2, perm, (int)n, 0, 60, stream));      // stable: then by residues 0..11
  hipLaunchKernelGGL(k_nw_lcp, dim3(nb), dim3(256), 0, stream, d_codes, d_off, perm, (int)n, lcp);
  DA_HIP_TRY(hipGetLastError());
  *perm_out = perm;
  *lcp_out = lcp;
  return DA_OK;
}

}  // namespace da
