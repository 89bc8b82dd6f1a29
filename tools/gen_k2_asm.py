#!/usr/bin/env python3
"""Generates dynaalign_amd/csrc/k2_loop_p12.inc: the hand-scheduled stage loop of the 12-plane compare kernel.

Why by hand: with 8-byte operands the loop needs 64 accumulators + 32 packed counters + 16 + 4 operand registers
= 116 VGPRs, which would allow 4 waves per SIMD (an even wave count and a fourth workgroup per CU), but hipcc needs
~30 more and spills the counters inside the loop.  The block below is ONE inline-asm statement with a
fixed register map; the surrounding C++ passes five per-lane values in v120..v124 and reads the counters back from LDS.

Register map (VGPR):  d[r][c] = v(8r+c) (0..63)   mis[r][c/2] = v(64+4r+c/2) (64..95)
                      a[r] = v(96+2r):v(97+2r)    b = v112:113 / v114:115 (double buffer)   v116 / v117 LDS read bases of the stage, v118:119 running DMA source, v125 scratch
inputs: v120 a_off (byte offset of the lane's first row operand inside a stage), v121 b_off, v122:123 DMA source of
        the wave's first instruction for stage 0, v124 write-back offset (4 * thread id)
        s[LB] LDS byte address of the ring, s[NS] number of stages, s[ST] source bytes per stage, s[WV] wave id
"""
import os
import sys

DUMP = int(os.environ.get("K2ASM_DUMP", "0"))   # debugging: leave the LDS ring as the loop left it (no counter write-back)
PRIO = int(os.environ.get("K2ASM_PRIO", "2"))       # wave priority inside the stage loop (0 = leave it alone): waves in their
# plane loop issue ahead of waves that are decoding a tile or storing one -- 26.6 -> 25.6 ms on the same box
SADDR = int(os.environ.get("K2ASM_SADDR", "0"))     # experiment: DMA with an SGPR base + 32-bit lane offset (no VALU, half the address registers)
SPREAD = int(os.environ.get("K2ASM_SPREAD", "0"))   # experiment: one DMA instruction before steps 0, 2 and 4 instead of three in a row
SDWA = int(os.environ.get("K2ASM_SDWA", "0"))       # experiment: add the odd column's popcount into the counter's high half with an SDWA add
EPRIO = int(os.environ.get("K2ASM_EPRIO", "0"))     # wave priority after the loop (tile epilogue)
SAFE = int(os.environ.get("K2ASM_SAFE", "0"))   # debugging: 1 = drain after every LDS read and DMA wait
REGOUT = int(os.environ.get("K2ASM_REGOUT", "1"))   # 1: the 32 packed counters leave the block in v64..v95 (asm outputs);
# 0: through LDS (barrier + 32 ds_write_b32 + wait, read back by the C++ epilogue -- the first version: +1 us per tile)

PERSIST = int(os.environ.get("K2ASM_PERSIST", "0"))   # 1: the block of the PERSISTENT kernel (k_mh_compare_p12): one tile of a
# workgroup's tile sequence -- the ring never drains between tiles.  Extra operands: %[fl] flags (bit 0: first tile of the
# workgroup = issue stages 0 and 1 here; bit 1: another tile follows = issue ITS stages 0 and 1 during the last two stages
# and wait for them before leaving), %[nl]:%[nh] the wave's source base of the next tile, %[sp] ring byte offset of stage 0.
# Always with the SGPR-base DMA form (the tile switch is two s_mov).
INLOOP = int(os.environ.get("K2ASM_INLOOP", "0"))     # 1: the block of k_mh_compare_q12 (round 3): the persistent block at THREE workgroups
# per CU (168 VGPRs), which stores the PREVIOUS tile as float64 from inside this tile's stage loop -- 16 pieces of four 16-byte
# streaming stores (a 2 x 2 block of the lane's 8 x 8 pairs, direct + mirrored), 16 / ns pieces per stage.  See gen_inloop().
if INLOOP:
    PERSIST = 1
if PERSIST:
    SADDR = 1
    REGOUT = 1
    if "K2ASM_PRIO" not in os.environ and not INLOOP:
        # the persistent block runs as the band kernel of the pipelined duplicate route, TWO workgroups per CU beside the store-bound row
        # expansion: at priority 2 its waves issue ahead of the expansion's store instructions and the step is 0.5 ms slower
        # (16.95 -> 16.45 ms at 100k, profiles/r04_d_pipeline_wave_priorities.txt; raising the expansion to 3 instead: 17.7)
        PRIO = 0
PLANES = int(os.environ.get("K2ASM_PLANES", "12"))   # 16: the block of k_mh_compare_a16 (uniform-like data: column dictionaries
# of up to 65 534 values).  Its operand is a PADDED copy of the 16 code planes: 80-byte LDS slots (64 bytes of planes + 16 of
# padding) make the loop's 8-byte operand reads bank-conflict-free with plain immediate offsets (the 64-byte slots of the
# compiled kernel need an XOR swizzle that an immediate cannot express), a wave's share of a stage is exactly five 1 KiB DMA
# pieces, and a ring of TWO 20 KiB stages keeps four workgroups per CU.  SGPR-base DMA form, counters out in registers.
# 14: the same block with SEVEN two-plane steps (K2ASM_PLANES=14, k2_loop_p14.inc): column dictionaries of up to 16 382 values need 14 code bits --
# uniform random 100k peptides: ~15 000 -- so the operand's planes 14 and 15 are zero and the eighth step would OR nothing in.
# 15: eight steps, the last one on plane 14 only (K2ASM_PLANES=15, k2_loop_p15.inc; uniform random 100k peptides: ~18 000 values per column).
if PLANES in (14, 15, 16):
    SADDR = 1
    REGOUT = 1
    assert not PERSIST
# 8: the persistent block on EIGHT code planes (K2ASM_PLANES=8 K2ASM_PERSIST=1 -> k2_loop_p8p.inc; with K2ASM_PRIO=2 -> k2_loop_p8.inc, the
# one-tile-per-workgroup form): the dense half of the heavy / rare split of the column dictionaries (round 4: the 254 most frequent
# values of a column keep dense codes, the rest is counted by the sparse route's incidence lists) -- 32-byte LDS slots, two 1 KiB DMA
# pieces per wave and stage, four two-plane steps, a 24 KiB ring.
if PLANES == 8:
    assert PERSIST and not INLOOP
SEGS, STEPS = (3, 6) if PLANES == 12 else (2, 4) if PLANES == 8 else (5, (PLANES + 1) // 2)   # 16-byte units per LDS slot; 2-plane steps per stage
HALF_LAST = PLANES == 15                                            # the last step's odd plane is zero in the operand: its instruction is left out
STAGE_BYTES = 256 * SEGS * 16  # 12288 (20480 with the padded 16-plane slots)
ROW = 16 * SEGS * 16          # byte distance between the lane's rows / columns in LDS: 768 (1280)
out = []
QTIMING = int(os.environ.get("K2ASM_QTIMING", "0"))   # in-loop block: s_memtime around the DMA wait and the stage barrier, sums in s68 / s69 (asm outputs %[tw] / %[tbr])
NOSTORE = int(os.environ.get("K2ASM_NOSTORE", "0"))   # experiments on the in-loop block: 1 = leave the global stores out, 2 = leave the table reads out
def e(x):
    if NOSTORE == 1 and x.startswith("global_store"):
        return
    if NOSTORE == 2 and x.startswith("ds_read_b64 v[16"):
        return
    out.append(x)
    if SAFE and x.startswith("ds_read"):
        out.append("s_waitcnt lgkmcnt(0)")
    if SAFE and x.startswith("s_waitcnt vmcnt(3)"):
        out.append("s_waitcnt vmcnt(0)")
    if SAFE and x.startswith("s_add_u32 m0"):
        out.append("s_nop 4")

def d(r, c): return "v%d" % (8 * r + c)
def mis(r, c2): return "v%d" % (64 + 4 * r + c2)
def a(r): return "v[%d:%d]" % (96 + 2 * r, 97 + 2 * r)
def ax(r): return "v%d" % (96 + 2 * r)
def ay(r): return "v%d" % (97 + 2 * r)
B = [("v[112:113]", "v112", "v113"), ("v[114:115]", "v114", "v115"), ("v[122:123]", "v122", "v123")]   # v122:123 is free once the DMA source has been copied

# scalar registers are operands: %[lb] %[ns] %[st]; we need a few scratch SGPRs -> use s-clobbers s40..s47
S_STAGE, S_SLOT, S_ISSUE_SLOT, S_TMP, S_M0, S_LEFT = "s40", "s41", "s42", "s43", "s44", "s45"

def issue(stage_reg_expr_comment):
    """DMA of one stage: 3 wave instructions of 1 KiB each.  v[118:119] holds the source of instruction 0 on entry
    and is advanced by 1 KiB in place for instructions 1 and 2 (the instruction's immediate offset is not used: it
    may move the LDS address as well), i.e. it is left 2 KiB past the stage's first byte.
    LDS destination in m0 = ring + slot*STAGE_BYTES + wave*3*1024 + q*1024 (S_M0 holds slot base + wave part)"""
    for q in range(3):
        e("s_add_u32 m0, %s, %d" % (S_M0, q * 1024))
        if SADDR:
            e("s_add_u32 s50, s48, %d" % (q * 1024))
            e("s_addc_u32 s51, s49, 0")
            e("global_load_lds_dwordx4 v118, s[50:51]")
            continue
        if q == 0:
            e("s_nop 0")
        else:
            e("v_add_co_u32 v118, vcc, 1024, v118")
            e("v_addc_co_u32 v119, vcc, 0, v119, vcc")
        e("global_load_lds_dwordx4 v[118:119], off")

def issue_piece(q):
    """instruction q of the DMA of stage st+2, guarded by the stage test (uniform branch)"""
    e("s_add_u32 %s, %s, 2" % (S_TMP, S_STAGE))
    e("s_cmp_lt_u32 %s, %%[ns]" % S_TMP)
    e("s_cbranch_scc0 7f")
    if q == 0:
        e("v_add_co_u32 v118, vcc, %s, v118" % S_LEFT)
        e("v_addc_co_u32 v119, vcc, 0, v119, vcc")
        e("s_add_u32 %s, s46, %s" % (S_M0, S_ISSUE_SLOT))
    else:
        e("v_add_co_u32 v118, vcc, 1024, v118")
        e("v_addc_co_u32 v119, vcc, 0, v119, vcc")
    e("s_add_u32 m0, %s, %d" % (S_M0, q * 1024))
    e("s_nop 0")
    e("global_load_lds_dwordx4 v[118:119], off")
    e("7:")

def count_group():
    for r in range(8):
        for c2 in range(4):
            e("v_bcnt_u32_b32 %s, %s, %s" % (mis(r, c2), d(r, 2 * c2), mis(r, c2)))
            e("v_bcnt_u32_b32 v125, %s, 0" % d(r, 2 * c2 + 1))
            if SDWA:
                e("v_add_u32_sdwa %s, v125, %s dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1" % (mis(r, c2), mis(r, c2)))
            else:
                e("v_lshl_add_u32 %s, v125, 16, %s" % (mis(r, c2), mis(r, c2)))

def b_read(buf, k, c):
    e("ds_read_b64 %s, v117 offset:%d" % (B[buf][0], c * ROW + k * 8))

def step(k, first, last, cur):
    """one 2-plane step.  On entry b(column 0) is in buffer `cur` and b(column 1) in the next buffer, both in flight
    or landed, and so are a[0..7]; column operands are fetched TWO columns ahead (three buffers), row operands of the
    next step right after their last use.  LDS reads return in order.  Returns the buffer holding the next step's b(0)."""
    for c in range(8):
        # keep two column operands in flight: fetch column c+2 (or the next step's column c+2-8) into the third buffer
        nxt2 = (cur + 2) % 3
        if c + 2 < 8:
            b_read(nxt2, k, c + 2); issued = True
        elif not last:
            b_read(nxt2, k + 1, c + 2 - 8); issued = True
        else:
            issued = False
        # b(cur) must have landed: the reads issued after it are b(c+1) [if any] and the one just issued
        younger = (1 if (c + 1 < 8 or not last) else 0) + (1 if issued else 0)
        if c == 0:
            younger = 1   # order so far: b(0), b(1), a[0..7], b(2): the row operands must have landed too
        e("s_waitcnt lgkmcnt(%d)" % younger)
        for r in range(8):
            if first:
                e("v_xor_b32 %s, %s, %s" % (d(r, c), ax(r), B[cur][2]))
            else:
                e("v_bitop3_b32 %s, %s, %s, %s bitop3:0xf6" % (d(r, c), d(r, c), ax(r), B[cur][2]))
            if not (HALF_LAST and last):
                e("v_bitop3_b32 %s, %s, %s, %s bitop3:0xf6" % (d(r, c), d(r, c), ay(r), B[cur][1]))
            if c == 7 and not last:
                e("ds_read_b64 %s, v116 offset:%d" % (a(r), r * ROW + (k + 1) * 8))
        cur = (cur + 1) % 3
    return cur


def wrap_slot(reg):
    """reg (a ring byte offset that was just advanced by one or two stages) back into [0, 3 stages)"""
    e("s_cmp_lt_u32 %s, %d" % (reg, 3 * STAGE_BYTES))
    e("s_cbranch_scc1 8f")
    e("s_sub_u32 %s, %s, %d" % (reg, reg, 3 * STAGE_BYTES))
    e("8:")

def issue_saddr():
    """DMA of one stage from the wave's running base s[48:49] (SEGS 1 KiB pieces), then advance the base by a stage"""
    for q in range(SEGS):
        e("s_add_u32 m0, %s, %d" % (S_M0, q * 1024))
        e("s_add_u32 s50, s48, %d" % (q * 1024))
        e("s_addc_u32 s51, s49, 0")
        e("global_load_lds_dwordx4 v118, s[50:51]")
    e("s_add_u32 s48, s48, %[st]")
    e("s_addc_u32 s49, s49, 0")

def gen_persistent():
    e("// generated by tools/gen_k2_asm.py (K2ASM_PERSIST=1%s) -- do not edit" % ("" if PLANES == 12 else " K2ASM_PLANES=%d" % PLANES))
    e("s_mov_b32 s47, m0")
    if PRIO:
        e("s_setprio %d" % PRIO)
    e("s_mov_b32 %s, 0" % S_STAGE)
    e("v_and_b32 v118, 0xfc, v124")          # 16 * lane: this lane's offset inside the wave's 1 KiB piece
    e("v_lshlrev_b32 v118, 2, v118")
    for r in range(8):
        for c2 in range(4):
            e("v_mov_b32 %s, 0" % mis(r, c2))
    e("s_mul_i32 %s, %%[wv], %d" % (S_TMP, SEGS * 1024))
    e("s_add_u32 s46, %[lb], " + S_TMP)                 # ring + this wave's part of a stage
    e("s_mov_b32 %s, %%[sp]" % S_SLOT)                  # ring slot (byte offset) of the stage being computed
    e("s_add_u32 %s, %%[sp], %d" % (S_ISSUE_SLOT, 2 * STAGE_BYTES))
    wrap_slot(S_ISSUE_SLOT)                              # ring slot the next issue goes to
    e("s_mov_b32 s48, %[sl]")
    e("s_mov_b32 s49, %[sh]")
    e("s_bitcmp1_b32 %[fl], 0")
    e("s_cbranch_scc0 10f")
    # first tile of the workgroup: its stages 0 and 1 are issued here
    e("s_add_u32 %s, s46, %s" % (S_M0, S_SLOT))
    issue_saddr()
    e("s_add_u32 %s, %s, %d" % (S_TMP, S_SLOT, STAGE_BYTES))
    wrap_slot(S_TMP)
    e("s_add_u32 %s, s46, %s" % (S_M0, S_TMP))
    issue_saddr()
    e("s_branch 11f")
    e("10:")
    # later tiles: the previous block issued (and waited for) stages 0 and 1; the running base starts at stage 2
    e("s_add_u32 s48, s48, %[st]")
    e("s_addc_u32 s49, s49, 0")
    e("s_add_u32 s48, s48, %[st]")
    e("s_addc_u32 s49, s49, 0")
    e("11:")
    # ---- stage loop
    e("2:")
    e("s_bitcmp1_b32 %[fl], 0")
    e("s_cbranch_scc1 20f")
    e("s_cmp_lt_u32 %s, 2" % S_STAGE)
    e("s_cbranch_scc1 4f")                   # stages 0 / 1 of a later tile landed before the previous tile's stores were issued
    e("20:")
    e("s_add_u32 %s, %s, 1" % (S_TMP, S_STAGE))
    e("s_cmp_lt_u32 %s, %%[ns]" % S_TMP)     # a younger stage of this tile in flight?
    e("s_cbranch_scc1 21f")
    e("s_bitcmp1_b32 %[fl], 1")              # ... or stage 0 of the next tile?
    e("s_cbranch_scc1 21f")
    e("s_waitcnt vmcnt(0)")
    e("s_branch 4f")
    e("21:")
    e("s_waitcnt vmcnt(%d)" % SEGS)
    e("4:")
    if QTIMING:
        e("s_memtime s[66:67]")
    e("s_barrier")
    if QTIMING:
        e("s_memtime s[70:71]")
        e("s_waitcnt lgkmcnt(0)")
        e("s_sub_u32 s64, s66, s64")          # cycles in the vmcnt wait (+ the compare chain)
        e("s_add_u32 s68, s68, s64")
        e("s_sub_u32 s66, s70, s66")          # cycles at the barrier
        e("s_add_u32 s69, s69, s66")
    e("v_add_u32 v116, %s, v120" % S_SLOT)
    e("v_add_u32 v117, %s, v121" % S_SLOT)
    b_read(0, 0, 0)
    b_read(1, 0, 1)
    for r in range(8):
        e("ds_read_b64 %s, v116 offset:%d" % (a(r), r * ROW))
    # issue stage + 2 (of this tile, or stage 0 / 1 of the next one) while those reads fly
    e("s_add_u32 %s, %s, 2" % (S_TMP, S_STAGE))
    e("s_cmp_lt_u32 %s, %%[ns]" % S_TMP)
    e("s_cbranch_scc1 30f")
    e("s_bitcmp1_b32 %[fl], 1")
    e("s_cbranch_scc0 5f")                   # last tile: nothing left to issue
    e("s_cmp_eq_u32 %s, %%[ns]" % S_TMP)
    e("s_cbranch_scc0 30f")                  # stage 1 of the next tile: the base was advanced by the stage-0 issue
    e("s_mov_b32 s48, %[nl]")                # stage 0 of the next tile: switch the running base
    e("s_mov_b32 s49, %[nh]")
    e("30:")
    e("s_add_u32 %s, s46, %s" % (S_M0, S_ISSUE_SLOT))
    issue_saddr()
    e("5:")
    e("s_cmp_eq_u32 %s, 0" % S_STAGE)
    e("s_cbranch_scc1 6f")
    count_group()
    e("6:")
    cur = 0
    for k in range(STEPS):
        cur = step(k, first=(k == 0), last=(k == STEPS - 1), cur=cur)
    assert cur == 0 or PLANES == 8   # (every stage preloads buffers 0 and 1 afresh: the loop body is emitted once and starts at buffer 0)
    e("s_add_u32 %s, %s, %d" % (S_SLOT, S_SLOT, STAGE_BYTES))
    e("s_cmp_lt_u32 %s, %d" % (S_SLOT, 3 * STAGE_BYTES))
    e("s_cselect_b32 %s, %s, 0" % (S_SLOT, S_SLOT))
    e("s_add_u32 %s, %s, %d" % (S_ISSUE_SLOT, S_ISSUE_SLOT, STAGE_BYTES))
    e("s_cmp_lt_u32 %s, %d" % (S_ISSUE_SLOT, 3 * STAGE_BYTES))
    e("s_cselect_b32 %s, %s, 0" % (S_ISSUE_SLOT, S_ISSUE_SLOT))
    e("s_add_u32 %s, %s, 1" % (S_STAGE, S_STAGE))
    e("s_cmp_lt_u32 %s, %%[ns]" % S_STAGE)
    e("s_cbranch_scc1 2b")
    count_group()
    if PRIO or EPRIO:
        e("s_setprio %d" % EPRIO)
    # the next tile's first two stages must have landed before this wave issues the tile's stores (the stage loop
    # of the next block does not wait for them: vmcnt counts loads and stores together, in order)
    e("s_bitcmp1_b32 %[fl], 1")
    e("s_cbranch_scc0 9f")
    e("s_waitcnt vmcnt(0)")
    e("9:")
    e("s_mov_b32 m0, s47")


def gen_inloop():
    """k_mh_compare_q12's block.  Same ring / DMA / plane-loop structure as gen_persistent(); differences:
      * v64..v95 come IN holding the previous tile's 32 packed mismatch counters (flag bit 2) and go OUT holding this tile's;
        the block parks the old ones in v128..v159 first;
      * every stage stores kk = 16 / ns PIECES of the previous tile: piece p = the lane's rows 2rp, 2rp+1 x column pair c2
        (rp = p >> 2, c2 = p & 3): two packed counters (picked with s_set_gpr_idx) -> four match counts -> four table reads
        (count -> double, the table is in LDS at %[tb]) -> two direct stores (rows 2rp, 2rp+1 at columns 2c2, 2c2+1) and, after two
        v_swap, two mirrored ones (the transposed tile: rows 2c2, 2c2+1 at columns 2rp, 2rp+1); addresses = a scalar base
        (tile + piece, s_mul / s_add) + a per-lane 32-bit offset (v119 direct, v127 mirrored);
      * vmcnt counts loads and stores together and in order, so the wait for a stage's DMA allows exactly the younger operations:
        the stores of the two previous stages (s53, s54; each 0 or 4 kk) + the three DMA instructions of the stage after
        (if one was issued) -- an immediate picked by a compare chain; stages 0 / 1 of a later tile wait the same way (the
        previous block's last two stages issued %[pw] stores each), so the block needs no drain at its end.
    Extra operands: %[tb] %[nn] %[odl]:%[odh] %[oml]:%[omh] (previous tile: &out[I0][J0], &out[J0][I0]) %[l8] = 8 ld, %[pw], %[kk]."""
    P, W1, W2, CNT, DO, LOOPC = "s52", "s53", "s54", "s55", "s60", "s61"
    e("// generated by tools/gen_k2_asm.py (K2ASM_INLOOP=1) -- do not edit")
    e("s_mov_b32 s47, m0")
    if PRIO:
        e("s_setprio %d" % PRIO)
    e("s_mov_b32 %s, 0" % S_STAGE)
    e("v_and_b32 v118, 0xfc, v124")
    e("v_lshlrev_b32 v118, 2, v118")
    for r in range(8):
        for c2 in range(4):
            e("v_mov_b32 v%d, %s" % (128 + 4 * r + c2, mis(r, c2)))     # park the previous tile's counters
    for r in range(8):
        for c2 in range(4):
            e("v_mov_b32 %s, 0" % mis(r, c2))
    if QTIMING:
        e("s_mov_b32 s68, 0")
        e("s_mov_b32 s69, 0")
    e("s_mov_b32 %s, 0" % P)
    e("s_mov_b32 %s, %%[pw]" % W1)
    e("s_mov_b32 %s, %%[pw]" % W2)
    e("s_mul_i32 %s, %%[wv], 3072" % S_TMP)
    e("s_add_u32 s46, %[lb], " + S_TMP)
    e("s_mov_b32 %s, %%[sp]" % S_SLOT)
    e("s_add_u32 %s, %%[sp], %d" % (S_ISSUE_SLOT, 2 * STAGE_BYTES))
    wrap_slot(S_ISSUE_SLOT)
    e("s_mov_b32 s48, %[sl]")
    e("s_mov_b32 s49, %[sh]")
    e("s_bitcmp1_b32 %[fl], 0")
    e("s_cbranch_scc0 10f")
    e("s_add_u32 %s, s46, %s" % (S_M0, S_SLOT))
    issue_saddr()
    e("s_add_u32 %s, %s, %d" % (S_TMP, S_SLOT, STAGE_BYTES))
    wrap_slot(S_TMP)
    e("s_add_u32 %s, s46, %s" % (S_M0, S_TMP))
    issue_saddr()
    e("s_branch 11f")
    e("10:")
    e("s_add_u32 s48, s48, %[st]")
    e("s_addc_u32 s49, s49, 0")
    e("s_add_u32 s48, s48, %[st]")
    e("s_addc_u32 s49, s49, 0")
    e("11:")

    def piece_a():
        e("s_lshr_b32 s58, %s, 2" % P)
        e("s_and_b32 s59, %s, 3" % P)
        e("s_lshl_b32 %s, s58, 3" % S_TMP)
        e("s_add_u32 %s, %s, s59" % (S_TMP, S_TMP))             # register index 8 rp + c2
        for half, (d0, d1) in enumerate(((160, 162), (164, 166))):
            e("s_set_gpr_idx_on %s, 0x1" % S_TMP)                  # SRC0 relative (clobbers m0: the DMA sets it before every use)
            e("s_nop 0")
            e("v_mov_b32 v126, v%d" % (128 + 4 * half))
            e("s_set_gpr_idx_off")
            e("v_sub_u32 v126, %[nn], v126")                       # two match counts: low half column 2 c2, high half column 2 c2 + 1
            e("v_lshlrev_b32 v%d, 3, v126" % d0)
            e("v_and_b32 v%d, 0x7fff8, v%d" % (d0, d0))
            e("v_add_u32 v%d, %%[tb], v%d" % (d0, d0))
            e("v_lshrrev_b32 v%d, 13, v126" % d1)
            e("v_and_b32 v%d, 0x7fff8, v%d" % (d1, d1))
            e("v_add_u32 v%d, %%[tb], v%d" % (d1, d1))
        for d in (160, 162, 164, 166):
            e("ds_read_b64 v[%d:%d], v%d" % (d, d + 1, d))

    def piece_b():
        e("s_waitcnt lgkmcnt(0)")
        e("s_lshr_b32 s58, %s, 2" % P)                             # rp
        e("s_and_b32 s59, %s, 3" % P)                              # c2
        e("s_lshl_b32 %s, %%[l8], 5" % LOOPC)                      # bytes per 32 rows (LOOPC is free here: reloaded below)
        # direct: &out[I0 + 32 rp][J0 + 32 c2] = od + rp * (32 ld 8) + c2 * 256
        e("s_mul_i32 %s, s58, %s" % (S_TMP, LOOPC))
        e("s_lshl_b32 s56, s59, 8")
        e("s_add_u32 %s, %s, s56" % (S_TMP, S_TMP))
        e("s_add_u32 s56, %%[odl], %s" % S_TMP)
        e("s_addc_u32 s57, %[odh], 0")
        e("global_store_dwordx4 v119, v[160:163], s[56:57] nt")
        e("s_add_u32 s56, s56, %[l8]")
        e("s_addc_u32 s57, s57, 0")
        e("global_store_dwordx4 v119, v[164:167], s[56:57] nt")
        # mirrored: &out[J0 + 32 c2][I0 + 32 rp] = om + c2 * (32 ld 8) + rp * 256
        e("s_mul_i32 %s, s59, %s" % (S_TMP, LOOPC))
        e("s_lshl_b32 s56, s58, 8")
        e("s_add_u32 %s, %s, s56" % (S_TMP, S_TMP))
        e("s_add_u32 s56, %%[oml], %s" % S_TMP)
        e("s_addc_u32 s57, %[omh], 0")
        e("s_nop 1")                                               # the two stores above have read their data registers
        e("v_swap_b32 v162, v164")
        e("v_swap_b32 v163, v165")
        e("global_store_dwordx4 v127, v[160:163], s[56:57] nt")
        e("s_add_u32 s56, s56, %[l8]")
        e("s_addc_u32 s57, s57, 0")
        e("global_store_dwordx4 v127, v[164:167], s[56:57] nt")
        e("s_add_u32 %s, %s, 1" % (P, P))
        e("s_add_u32 %s, %s, 4" % (CNT, CNT))
        e("s_nop 1")

    e("2:")
    if QTIMING:
        e("s_memtime s[64:65]")
    # ---- wait for this stage's DMA: allowed in flight = stores of the two previous stages + the next stage's DMA (if issued)
    e("s_add_u32 %s, %s, %s" % (S_TMP, W1, W2))
    e("s_add_u32 s58, %s, 1" % S_STAGE)
    e("s_cmp_lt_u32 s58, %[ns]")
    e("s_cbranch_scc1 20f")
    e("s_bitcmp1_b32 %[fl], 1")
    e("s_cbranch_scc0 21f")
    e("20:")
    e("s_add_u32 %s, %s, 3" % (S_TMP, S_TMP))
    e("21:")
    vals = [11, 3, 7, 8, 0, 4, 19, 16, 35, 32]
    for v in vals:
        e("s_cmp_eq_u32 %s, %d" % (S_TMP, v))
        e("s_cbranch_scc1 %df" % (100 + v))
    e("s_waitcnt vmcnt(0)")
    e("s_branch 4f")
    for v in vals:
        e("%d:" % (100 + v))
        e("s_waitcnt vmcnt(%d)" % v)
        e("s_branch 4f")
    e("4:")
    if QTIMING:
        e("s_memtime s[66:67]")
    e("s_barrier")
    if QTIMING:
        e("s_memtime s[70:71]")
        e("s_waitcnt lgkmcnt(0)")
        e("s_sub_u32 s64, s66, s64")          # cycles in the vmcnt wait (+ the compare chain)
        e("s_add_u32 s68, s68, s64")
        e("s_sub_u32 s66, s70, s66")          # cycles at the barrier
        e("s_add_u32 s69, s69, s66")
    e("v_add_u32 v116, %s, v120" % S_SLOT)
    e("v_add_u32 v117, %s, v121" % S_SLOT)
    b_read(0, 0, 0)
    b_read(1, 0, 1)
    for r in range(8):
        e("ds_read_b64 %s, v116 offset:%d" % (a(r), r * ROW))
    # issue stage + 2 (of this tile, or stage 0 / 1 of the next one)
    e("s_add_u32 %s, %s, 2" % (S_TMP, S_STAGE))
    e("s_cmp_lt_u32 %s, %%[ns]" % S_TMP)
    e("s_cbranch_scc1 30f")
    e("s_bitcmp1_b32 %[fl], 1")
    e("s_cbranch_scc0 5f")
    e("s_cmp_eq_u32 %s, %%[ns]" % S_TMP)
    e("s_cbranch_scc0 30f")
    e("s_mov_b32 s48, %[nl]")
    e("s_mov_b32 s49, %[nh]")
    e("30:")
    e("s_add_u32 %s, s46, %s" % (S_M0, S_ISSUE_SLOT))
    issue_saddr()
    e("5:")
    # ---- the previous tile's pieces of this stage: the first one's table reads fly under the popcounts
    e("s_mov_b32 %s, 0" % CNT)
    e("s_mov_b32 %s, 0" % DO)
    e("s_bitcmp1_b32 %[fl], 2")
    e("s_cbranch_scc0 40f")
    e("s_cmp_lt_u32 %s, 16" % P)
    e("s_cbranch_scc0 40f")
    e("s_mov_b32 %s, 1" % DO)
    piece_a()
    e("40:")
    e("s_cmp_eq_u32 %s, 0" % S_STAGE)
    e("s_cbranch_scc1 6f")
    count_group()
    e("6:")
    e("s_cmp_eq_u32 %s, 0" % DO)
    e("s_cbranch_scc1 43f")
    piece_b()
    e("s_sub_u32 %s, %%[kk], 1" % LOOPC)
    e("41:")
    e("s_cmp_eq_u32 %s, 0" % LOOPC)
    e("s_cbranch_scc1 43f")
    e("s_cmp_lt_u32 %s, 16" % P)
    e("s_cbranch_scc0 43f")
    e("s_mov_b32 s62, %s" % LOOPC)                                  # piece_b uses LOOPC as scratch
    piece_a()
    piece_b()
    e("s_sub_u32 %s, s62, 1" % LOOPC)
    e("s_branch 41b")
    e("43:")
    cur = 0
    for k in range(STEPS):
        cur = step(k, first=(k == 0), last=(k == STEPS - 1), cur=cur)
    assert cur == 0
    e("s_mov_b32 %s, %s" % (W2, W1))
    e("s_mov_b32 %s, %s" % (W1, CNT))
    e("s_add_u32 %s, %s, %d" % (S_SLOT, S_SLOT, STAGE_BYTES))
    e("s_cmp_lt_u32 %s, %d" % (S_SLOT, 3 * STAGE_BYTES))
    e("s_cselect_b32 %s, %s, 0" % (S_SLOT, S_SLOT))
    e("s_add_u32 %s, %s, %d" % (S_ISSUE_SLOT, S_ISSUE_SLOT, STAGE_BYTES))
    e("s_cmp_lt_u32 %s, %d" % (S_ISSUE_SLOT, 3 * STAGE_BYTES))
    e("s_cselect_b32 %s, %s, 0" % (S_ISSUE_SLOT, S_ISSUE_SLOT))
    e("s_add_u32 %s, %s, 1" % (S_STAGE, S_STAGE))
    e("s_cmp_lt_u32 %s, %%[ns]" % S_STAGE)
    e("s_cbranch_scc1 2b")
    count_group()
    if PRIO or EPRIO:
        e("s_setprio %d" % EPRIO)
    if QTIMING:
        e("s_mov_b32 %[tw], s68")
        e("s_mov_b32 %[tbr], s69")
    e("s_mov_b32 m0, s47")

def gen_16():
    e("// generated by tools/gen_k2_asm.py (K2ASM_PLANES=%d) -- do not edit" % PLANES)
    e("s_mov_b32 s47, m0")
    if PRIO:
        e("s_setprio %d" % PRIO)
    e("s_mov_b32 %s, 0" % S_STAGE)
    e("v_and_b32 v118, 0xfc, v124")          # 16 * lane: this lane's offset inside the wave's 1 KiB piece
    e("v_lshlrev_b32 v118, 2, v118")
    for r in range(8):
        for c2 in range(4):
            e("v_mov_b32 %s, 0" % mis(r, c2))
    e("s_mul_i32 %s, %%[wv], %d" % (S_TMP, SEGS * 1024))
    e("s_add_u32 s46, %[lb], " + S_TMP)                 # ring + this wave's part of a stage (5 KiB)
    e("s_mov_b32 s48, %[sl]")
    e("s_mov_b32 s49, %[sh]")
    def issue5():
        for q in range(SEGS):
            e("s_add_u32 m0, %s, %d" % (S_M0, q * 1024))
            e("s_add_u32 s50, s48, %d" % (q * 1024))
            e("s_addc_u32 s51, s49, 0")
            e("global_load_lds_dwordx4 v118, s[50:51]")
        e("s_add_u32 s48, s48, %[st]")
        e("s_addc_u32 s49, s49, 0")
    e("s_mov_b32 %s, s46" % S_M0)
    issue5()                                             # stage 0 -> ring slot 0
    e("s_mov_b32 %s, 0" % S_SLOT)
    e("s_mov_b32 %s, %d" % (S_ISSUE_SLOT, STAGE_BYTES))
    e("2:")
    e("s_waitcnt vmcnt(0)")                              # two-stage ring: only the stage about to be computed is in flight
    e("s_barrier")
    e("v_add_u32 v116, %s, v120" % S_SLOT)
    e("v_add_u32 v117, %s, v121" % S_SLOT)
    b_read(0, 0, 0)
    b_read(1, 0, 1)
    for r in range(8):
        e("ds_read_b64 %s, v116 offset:%d" % (a(r), r * ROW))
    e("s_add_u32 %s, %s, 1" % (S_TMP, S_STAGE))
    e("s_cmp_lt_u32 %s, %%[ns]" % S_TMP)
    e("s_cbranch_scc0 5f")
    e("s_add_u32 %s, s46, %s" % (S_M0, S_ISSUE_SLOT))   # the other slot: everyone left it before this stage's barrier
    issue5()
    e("5:")
    e("s_cmp_eq_u32 %s, 0" % S_STAGE)
    e("s_cbranch_scc1 6f")
    count_group()
    e("6:")
    cur = 0
    for k in range(STEPS):
        cur = step(k, first=(k == 0), last=(k == STEPS - 1), cur=cur)   # every stage starts again with buffer 0 (fresh preload)
    e("s_xor_b32 %s, %s, %d" % (S_SLOT, S_SLOT, STAGE_BYTES))
    e("s_xor_b32 %s, %s, %d" % (S_ISSUE_SLOT, S_ISSUE_SLOT, STAGE_BYTES))
    e("s_add_u32 %s, %s, 1" % (S_STAGE, S_STAGE))
    e("s_cmp_lt_u32 %s, %%[ns]" % S_STAGE)
    e("s_cbranch_scc1 2b")
    count_group()
    if PRIO or EPRIO:
        e("s_setprio %d" % EPRIO)
    e("s_mov_b32 m0, s47")


if INLOOP:
    gen_inloop()
elif PERSIST:
    gen_persistent()
elif PLANES in (14, 15, 16):
    gen_16()
else:
    e("// generated by tools/gen_k2_asm.py -- do not edit")
    e("s_mov_b32 s47, m0")                                # m0 is compiler-reserved: saved here, restored at the end of the block
    if PRIO:
        e("s_setprio %d" % PRIO)
    # ---- setup
    e("s_mov_b32 %s, 0" % S_STAGE)                       # stage being computed
    e("s_sub_u32 %s, %%[st], 2048" % S_LEFT)               # the running DMA source sits 2 KiB into the stage it last issued
    if SADDR:
        e("s_mov_b32 s48, %[sl]")               # the wave's first source address (lane 0) as the scalar base ...
        e("s_mov_b32 s49, %[sh]")
        e("v_and_b32 v118, 0xfc, v124")         # ... and 16 * lane as the per-lane offset
        e("v_lshlrev_b32 v118, 2, v118")
    else:
        e("v_mov_b32 v118, v122")
        e("v_mov_b32 v119, v123")
    for r in range(8):
        for c2 in range(4):
            e("v_mov_b32 %s, 0" % mis(r, c2))
    # LDS DMA offset of this wave inside a stage: wave * 3 KiB (the wave id comes in as an SGPR operand)
    e("s_mul_i32 %s, %%[wv], 3072" % S_TMP)
    e("s_add_u32 s46, %[lb], " + S_TMP)                 # s46 = ring + wave part (slot 0)
    # issue stages 0 and 1
    e("s_mov_b32 %s, s46" % S_M0)
    issue("0")
    e("s_cmp_lt_u32 1, %[ns]")
    e("s_cbranch_scc0 1f")
    if SADDR:
        e("s_add_u32 s48, s48, %[st]")
        e("s_addc_u32 s49, s49, 0")
    else:
        e("v_add_co_u32 v118, vcc, %s, v118" % S_LEFT)
        e("v_addc_co_u32 v119, vcc, 0, v119, vcc")
    e("s_add_u32 %s, s46, %d" % (S_M0, STAGE_BYTES))
    issue("1")
    e("1:")
    e("s_mov_b32 %s, 0" % S_SLOT)                        # ring slot of the stage being computed (byte offset)
    e("s_mov_b32 %s, %d" % (S_ISSUE_SLOT, 2 * STAGE_BYTES))  # ring slot the next issue goes to
    # ---- stage loop
    e("2:")
    e("s_add_u32 %s, %s, 1" % (S_TMP, S_STAGE))
    e("s_cmp_lt_u32 %s, %%[ns]" % S_TMP)                  # is there a younger stage in flight?
    e("s_cbranch_scc0 3f")
    e("s_waitcnt vmcnt(3)")
    e("s_branch 4f")
    e("3:")
    e("s_waitcnt vmcnt(0)")
    e("4:")
    e("s_barrier")
    # LDS read bases of this stage
    e("v_add_u32 v116, %s, v120" % S_SLOT)
    e("v_add_u32 v117, %s, v121" % S_SLOT)
    # preload step 0 operands: b(col 0), b(col 1), then the 8 row operands (the order step() counts on)
    b_read(0, 0, 0)
    b_read(1, 0, 1)
    for r in range(8):
        e("ds_read_b64 %s, v116 offset:%d" % (a(r), r * ROW))
    # issue stage + 2 while those reads fly
    if SPREAD:
        issue_piece(0)
    else:
        e("s_add_u32 %s, %s, 2" % (S_TMP, S_STAGE))
        e("s_cmp_lt_u32 %s, %%[ns]" % S_TMP)
        e("s_cbranch_scc0 5f")
        if SADDR:
            e("s_add_u32 s48, s48, %[st]")
            e("s_addc_u32 s49, s49, 0")
        else:
            e("v_add_co_u32 v118, vcc, %s, v118" % S_LEFT)
            e("v_addc_co_u32 v119, vcc, 0, v119, vcc")
        e("s_add_u32 %s, s46, %s" % (S_M0, S_ISSUE_SLOT))
        issue("st+2")
        e("5:")
    # popcounts of the previous group (skipped for stage 0) while the reads fly
    e("s_cmp_eq_u32 %s, 0" % S_STAGE)
    e("s_cbranch_scc1 6f")
    count_group()
    e("6:")
    # the 6 steps.  NOTE: every step starts with b in buffer 0: 8 columns -> the buffer index is back at 0 after a step
    cur = 0
    for k in range(STEPS):
        if SPREAD and k in (2, 4):
            issue_piece(k // 2)
        cur = step(k, first=(k == 0), last=(k == STEPS - 1), cur=cur)
    # 6 steps x 8 columns = 48 buffer advances = 0 mod 3: the next stage starts with buffer 0 again
    assert cur == 0
    # advance
    e("s_add_u32 %s, %s, %d" % (S_SLOT, S_SLOT, STAGE_BYTES))
    e("s_cmp_lt_u32 %s, %d" % (S_SLOT, 3 * STAGE_BYTES))
    e("s_cselect_b32 %s, %s, 0" % (S_SLOT, S_SLOT))
    e("s_add_u32 %s, %s, %d" % (S_ISSUE_SLOT, S_ISSUE_SLOT, STAGE_BYTES))
    e("s_cmp_lt_u32 %s, %d" % (S_ISSUE_SLOT, 3 * STAGE_BYTES))
    e("s_cselect_b32 %s, %s, 0" % (S_ISSUE_SLOT, S_ISSUE_SLOT))
    e("s_add_u32 %s, %s, 1" % (S_STAGE, S_STAGE))
    e("s_cmp_lt_u32 %s, %%[ns]" % S_STAGE)
    e("s_cbranch_scc1 2b")
    count_group()
    # ---- counters -> LDS (plane k of the write-back area = 1 KiB of lane-consecutive dwords), after everyone left the ring
    if PRIO or EPRIO:
        e("s_setprio %d" % EPRIO)
    if not REGOUT:
        e("s_barrier")
    if not DUMP and not REGOUT:
        e("v_add_u32 v125, %[lb], v124")
        for r in range(8):
            for c2 in range(4):
                e("ds_write_b32 v125, %s offset:%d" % (mis(r, c2), (4 * r + c2) * 1024))
        e("s_waitcnt lgkmcnt(0)")
    e("s_mov_b32 m0, s47")


with open(sys.argv[1], "w") as f:
    for l in out:
        if l.startswith("//"):
            f.write(l + "\n")
        else:
            f.write('"%s\\n\\t"\n' % l)
print("wrote", sys.argv[1], len(out), "lines")
