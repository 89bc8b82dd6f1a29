#!/usr/bin/env python3
"""Generates tools/experiments/nw_rows_p<NMAX>.inc: the hand-scheduled DP rows of k_nw_short's combined-key cell (nw_kernels.hip,
nw_row_ck; reference src/pairwiseSeqAlign.cpp:238-281) -- all rows 1..m of ONE sequence1 against the lane's sequence2, as ONE
inline-asm statement with a fixed register map (+ nw_rows_p<NMAX>_bind.inc: the C++ register binding that goes with it).

Structure:
  * TWO ROWS per sweep, skewed by one column: row A (r) at column c and row B (r + 1) at column c - 1 in the same step.  Row A's
    results never enter the row arrays (row B consumes them one and two steps later from rotating temporaries), row B overwrites
    VM[c-1] / XP[c-1] right after row A's cell c has read VM[c-1] as its diagonal -- no copy of the diagonal neighbour (the compiled
    row spends a v_mov per cell on it), 9 + 1 VALU instructions per cell, two independent dependency chains per wave.
  * every per-cell operand is a VGPR (an SGPR source costs v_add_u32 / v_bitop3 their full rate on gfx950), and no v_bitop3 reads three
    VGPRs of equal index parity (half rate; v_max3 / v_add3 do not care: profiles/r04_b_ubench_bank_parity.txt).
  * the table reads go through a ring R columns ahead (the read returns into the register that held its address), also across block
    boundaries; the row's residue code is read a block ahead, read + wait + use inside this one statement (the compiled
    kernel's row-ahead ds_read_u8 relied on register allocation between two statements).
What round 4 measured with it (profiles/r04_c_nw_row_block_variants.txt, r04_c_nw_row_forms_in_kernel.txt; tools/ubench/nw_rows.hip):
  * it is NOT faster than the compiled row (below).
  * the row is bound by its three max-class instructions: with v_max_i32 / v_max3_i32 replaced by v_add_u32 (wrong results) the block
    runs 333 instead of 452 ms per 2*10^12 cells -- 2.2 add-units each, half of the row's time -- while removing ALL table reads and
    address adds changes nothing (461 ms) and neither does the order of the instructions (interleaved or not: 452 / 450).
  * reads in flight cost time: ring 10 with counted waits 452 ms, ring 5: 440, ring 2: 443, and draining the queue (lgkmcnt(0)) before
    every use 430-432 whatever the ring -- latency is hidden by the other three waves of the SIMD, a deep LDS queue is not free.
    Hence the defaults: ring 2, drained waits (NWASM_WAITS=counted keeps the counted form; the CPU model checks both).
  * v_max_i16 IS full rate but only on the low halves (SDWA / op_sel forms are not: profiles/r04_b_ubench_max16_hi.txt), and the key's
    score must sit in the high bits for the three-way choice -- no 16-bit shortcut for the gap-state maxes.
  * in the kernel (same box, same process): ordered DP 102.1 vs 97.5 ms for the compiled row, direct sweep 424 vs 417 ms.
The product library therefore keeps the compiled row; the block is built into the experiment twin only (tools/experiments/build.sh,
DYNAALIGN_NW_ASM=1) and stays verified: CPU model (tests/test_nw_asm_model.py) and the GPU suite run on that library.

Register map (VGPR), N = NMAX, R = ring columns (divides N); NWASM_PARITY=1 (default) shifts XP / BO / rings by one register:
  VM[c] = v(c)            best'(r-1, c+1), priority cleared  (outputs: the last row's values)
  XP[c] = v(N+1+c)        Ix' of the row above with the payload of the cell it leaves
  BO[c] = v(2N+1+c)       inputs: 4 * (sequence2's residue code at column c+1)   (preserved)
  TA[k], TB[k]            table-read ring of rows A / B
  then: kx ky pm pc vi xf iy0 rc cA cB offA offB offAn offBn | a_vmn[3] a_xp[2] a_ypl a_t a_vd | b_ypl b_t b_vd  (parities chosen, see PARITY)
operands: %[m] rows (>= 1), %[rc] LDS address of the row's residue codes (bytes, zero-padded), %[tb] LDS address of the 24 x 24 key
  table, %[kx] %[ky] %[pm] %[pc] the four per-cell constants, %[vi] (ge - go) << S2, %[l0] NEG << S2, %[xf] Ix' of row 1
scalar scratch: s40 (rows left), vcc / scc
"""
import os
import sys

N = int(os.environ.get("NWASM_NMAX", "20"))
R = int(os.environ.get("NWASM_RING", "2"))
assert N % R == 0 and R >= 2
SAFE = os.environ.get("NWASM_WAITS", "drain") == "drain"   # drain the LDS queue before every use (measured faster than counted waits)
ROWSTRIDE = 24 * 4                                     # bytes per table row (24 int32 keys)

PARITY = int(os.environ.get("NWASM_PARITY", "1"))      # 1: register numbers chosen so that no three-source instruction (v_max3, v_bitop3) reads three
# VGPRs of equal index parity (half rate on gfx950: profiles/r04_b_ubench_bank_parity.txt); 0: the plain consecutive map (experiments)
ABLATE = os.environ.get("NWASM_ABLATE", "")            # timing experiments only (WRONG results): noread / nowait / nomax / nobitop
ILV = int(os.environ.get("NWASM_ILV", "1"))             # 1: rows A and B alternate instruction by instruction inside a phase; 0: A's phase, then B's
_names = ["kx", "ky", "pm", "pc", "vi", "xf", "iy0", "rc", "cA", "cB", "offA", "offB", "offAn", "offBn",
          "a_vmn0", "a_vmn1", "a_vmn2", "a_xp0", "a_xp1", "a_ypl", "a_t", "a_vd", "b_ypl", "b_t", "b_vd"]
VM = ["v%d" % c for c in range(N)]
if PARITY:
    assert N % 2 == 0
    # XP[c] opposite to VM[c]; the payload mask odd; the diagonal sums odd; both rows' Iy' and row A's Ix' temporaries even
    XP = ["v%d" % (N + 1 + c) for c in range(N)]
    BO = ["v%d" % (2 * N + 1 + c) for c in range(N)]
    TA = ["v%d" % (3 * N + 1 + k) for k in range(R)]
    TB = ["v%d" % (3 * N + 1 + R + k) for k in range(R)]
    want = {"pm": 1, "a_vd": 1, "b_vd": 1, "a_ypl": 0, "b_ypl": 0, "a_xp0": 0, "a_xp1": 0}
    free = [N] + list(range(3 * N + 1 + 2 * R, 3 * N + 1 + 2 * R + 64))
    REG = {}
    for nm in sorted(_names, key=lambda x: (x not in want, _names.index(x))):
        par = want.get(nm)
        r = next(x for x in free if par is None or x % 2 == par)
        free.remove(r)
        REG[nm] = "v%d" % r
    NREGS = max(int(v[1:]) for v in REG.values()) + 1
else:
    XP = ["v%d" % (N + c) for c in range(N)]
    BO = ["v%d" % (2 * N + c) for c in range(N)]
    TA = ["v%d" % (3 * N + k) for k in range(R)]
    TB = ["v%d" % (3 * N + R + k) for k in range(R)]
    _base = 3 * N + 2 * R
    REG = {nm: "v%d" % (_base + i) for i, nm in enumerate(_names)}
    NREGS = _base + len(_names)
S_LEFT = "s40"

out = []
queue = []          # LDS operations in issue order: (tag)
done_upto = 0       # queue[:done_upto] are known to have returned


def e(x):
    if ABLATE == "noread" and (x.startswith("ds_read_b32") or (x.startswith("v_add_u32") and any(x.startswith("v_add_u32 %s," % t) for t in TA + TB))):
        return
    if ABLATE in ("nowait", "noread") and x.startswith("s_waitcnt") and not x.endswith("lgkmcnt(0)"):
        return
    if ABLATE == "nomax":
        if x.startswith("v_max_i32"):
            x = x.replace("v_max_i32", "v_add_u32")
        elif x.startswith("v_max3_i32"):
            a = x.split(",")
            x = "v_add_u32" + a[0][len("v_max3_i32"):] + "," + a[1] + "," + a[2]
    if ABLATE == "nobitop" and x.startswith("v_bitop3_b32"):
        a = x.split(",")
        x = "v_and_b32" + a[0][len("v_bitop3_b32"):] + "," + a[1] + "," + a[2]
    out.append(x)


def lds_issue(text, tag):
    e(text)
    queue.append(tag)


def lds_need(tag):
    """the value of the LDS read `tag` is about to be used: counted wait (in-order return)"""
    global done_upto
    idx = max(i for i, t in enumerate(queue) if t == tag)
    if idx < done_upto:
        return
    if SAFE:
        e("s_waitcnt lgkmcnt(0)")
        done_upto = len(queue)
        return
    younger = len(queue) - 1 - idx
    k = min(younger, 15)
    e("s_waitcnt lgkmcnt(%d)" % k)
    done_upto = len(queue) - k


def queue_state():
    """what the next body may assume: tags of the reads still outstanding, oldest first"""
    return tuple(queue[done_upto:])


def prefetch(row, col, nxt):
    """issue the table read of (row A / B, column col) of this block (nxt = False) or the next one (True): address add into the ring
    register, read back into it"""
    ring = (TA if row == "A" else TB)[col % R]
    off = REG[("offA" if row == "A" else "offB") + ("n" if nxt else "")]
    e("v_add_u32 %s, %s, %s" % (ring, BO[col], off))
    lds_issue("ds_read_b32 %s, %s" % (ring, ring), ("T", row, col, 1 if nxt else 0))


def codes_to_offsets(ca, cb, oa, ob):
    """table-row byte offsets of two rows from their residue codes (& 31: a code read past the sequence's padding stays inside LDS)"""
    for c, o in ((ca, oa), (cb, ob)):
        e("v_and_b32 %s, 31, %s" % (REG[c], REG[c]))
        e("v_mul_u32_u24 %s, %d, %s" % (REG[o], ROWSTRIDE, REG[c]))
        e("v_add_u32 %s, %%[tb], %s" % (REG[o], REG[o]))


def cell_A(c, first):
    """row A at column c, as four phases of instructions (gap states; table value + diagonal; choice; payload merge)"""
    vmn, vmn_left = REG["a_vmn%d" % (c % 3)], REG["a_vmn%d" % ((c - 1) % 3)]
    xp = REG["a_xp%d" % (c % 2)]
    ypl, t, vd = REG["a_ypl"], REG["a_t"], REG["a_vd"]
    p1, p2, p3, p4 = [], [], [], []
    if not first:
        p1.append("v_add_u32 %s, %s, %s" % (t, VM[c], REG["kx"]))                        # open a gap from M(r-1, c), priority 1
        p1.append("v_max_i32 %s, %s, %s" % (xp, t, XP[c]))                                # Ix'(r, c)                      (:255-257)
    if c > 0:
        p2.append("v_add_u32 %s, %s, %s" % (t, vmn_left, REG["ky"]))                     # open a gap from M(r, c-1), priority 0
        p2.append("v_max_i32 %s, %s, %s" % (ypl, t, ypl))                                 # Iy'(r, c)                      (:260-262)
    p3.append(("need", ("T", "A", c, 0)))
    diag = (REG["vi"] if c > 0 else None) if first else (VM[c - 1] if c > 0 else REG["vi"])
    if diag is None:                                                                      # row 1, column 1: max(M, Ix, Iy)(0, 0) = 0
        p3.append("v_mov_b32 %s, %s" % (vd, TA[c % R]))
    else:
        p3.append("v_add_u32 %s, %s, %s" % (vd, diag, TA[c % R]))                         # diagonal, priority 2, D + 1 (+ match)  (:265-271)
    ixf = REG["xf"] if first else xp
    iyf = ypl if c > 0 else REG["iy0"]
    p4.append("v_max3_i32 %s, %s, %s, %s" % (vmn, vd, ixf, iyf))
    p4.append("v_and_b32 %s, %s, %s" % (vmn, vmn, REG["pc"]))
    p4.append("v_bitop3_b32 %s, %s, %s, %s bitop3:0xd8" % (xp, ixf, vmn, REG["pm"]))      # gap state's score + the payload of the cell it leaves
    p4.append("v_bitop3_b32 %s, %s, %s, %s bitop3:0xd8" % (ypl, iyf, vmn, REG["pm"]))
    return [p1, p2, p3, p4]


def cell_B(c):
    """row B (r + 1) at column c, one step behind row A: its `up` inputs are row A's temporaries of the previous step, its diagonal
    row A's result of two steps ago; results go into the row arrays.  Only phase 4 writes VM[c]: row A's cell c + 1 reads it as
    its diagonal in phase 3 of the same step."""
    up_vmn, diag_vmn = REG["a_vmn%d" % (c % 3)], REG["a_vmn%d" % ((c - 1) % 3)]
    up_xp = REG["a_xp%d" % (c % 2)]
    ypl, t, vd = REG["b_ypl"], REG["b_t"], REG["b_vd"]
    p1, p2, p3, p4 = [], [], [], []
    p1.append("v_add_u32 %s, %s, %s" % (t, up_vmn, REG["kx"]))
    p1.append("v_max_i32 %s, %s, %s" % (XP[c], t, up_xp))
    if c > 0:
        p2.append("v_add_u32 %s, %s, %s" % (t, VM[c - 1], REG["ky"]))
        p2.append("v_max_i32 %s, %s, %s" % (ypl, t, ypl))
    p3.append(("need", ("T", "B", c, 0)))
    p3.append("v_add_u32 %s, %s, %s" % (vd, diag_vmn if c > 0 else REG["vi"], TB[c % R]))
    iyf = ypl if c > 0 else REG["iy0"]
    p4.append("v_max3_i32 %s, %s, %s, %s" % (VM[c], vd, XP[c], iyf))
    p4.append("v_and_b32 %s, %s, %s" % (VM[c], VM[c], REG["pc"]))
    p4.append("v_bitop3_b32 %s, %s, %s, %s bitop3:0xd8" % (ypl, iyf, VM[c], REG["pm"]))
    p4.append("v_bitop3_b32 %s, %s, %s, %s bitop3:0xd8" % (XP[c], XP[c], VM[c], REG["pm"]))
    return [p1, p2, p3, p4]


def emit_op(op):
    if isinstance(op, tuple):
        lds_need(op[1])
    else:
        e(op)


def emit_interleaved(a_ph, b_ph):
    """phase by phase, the two rows' instructions alternating inside a phase (two independent dependency chains)"""
    for ph in range(4):
        a_ops = a_ph[ph] if a_ph else []
        b_ops = b_ph[ph] if b_ph else []
        if ph == 2:                                    # row B's table read was issued after row A's: one wait covers both
            for op in [o for o in b_ops if isinstance(o, tuple)] + [o for o in a_ops if isinstance(o, tuple)]:
                emit_op(op)
            a_ops = [o for o in a_ops if not isinstance(o, tuple)]
            b_ops = [o for o in b_ops if not isinstance(o, tuple)]
        if not ILV:
            for op in a_ops + b_ops:
                emit_op(op)
            continue
        for k in range(max(len(a_ops), len(b_ops))):
            if k < len(a_ops):
                emit_op(a_ops[k])
            if k < len(b_ops):
                emit_op(b_ops[k])


def block(first):
    """two rows (r, r + 1); `first`: r = 1.  On entry: offA / offB = table offsets of these rows, cA / cB = residue codes of the NEXT
    block's rows (read a block ago), ring = this block's columns 0 .. R-1 of both rows in flight."""
    lds_need(("C", "A"))
    lds_need(("C", "B"))
    codes_to_offsets("cA", "cB", "offAn", "offBn")
    e("v_add_u32 %s, 2, %s" % (REG["rc"], REG["rc"]))
    lds_issue("ds_read_u8 %s, %s" % (REG["cA"], REG["rc"]), ("C", "A"))                  # codes of the block after next
    lds_issue("ds_read_u8 %s, %s offset:1" % (REG["cB"], REG["rc"]), ("C", "B"))
    for s in range(N + 1):
        emit_interleaved(cell_A(s, first) if s < N else None, cell_B(s - 1) if s >= 1 else None)
        # the ring slots just consumed take the reads R columns further on (the next block's when past the row's end)
        if s < N:
            prefetch("A", (s + R) % N, s + R >= N)
        if s >= 1:
            prefetch("B", (s - 1 + R) % N, s - 1 + R >= N)
    e("v_mov_b32 %s, %s" % (REG["offA"], REG["offAn"]))
    e("v_mov_b32 %s, %s" % (REG["offB"], REG["offBn"]))
    # relabel: the next block's reads become this block's
    for i, t in enumerate(queue):
        if t[0] == "T" and t[3] == 1:
            queue[i] = ("T", t[1], t[2], 0)
        elif t[0] == "T":
            queue[i] = ("T", t[1], t[2], -1)


def single(first):
    """one row alone (the last of an odd count; or m = 1).  A first row touches neither array and writes VM directly; a later row reads
    VM[c-1] as its diagonal from the array, so its results go into XP[c] (read just before, dead afterwards) and are copied at the end."""
    for c in range(N):
        emit_interleaved(cell_A(c, first), None)
        e("v_mov_b32 %s, %s" % (VM[c] if first else XP[c], REG["a_vmn%d" % (c % 3)]))
        if c + R < N:
            prefetch("A", c + R, False)
    if not first:
        for c in range(N):
            e("v_mov_b32 %s, %s" % (VM[c], XP[c]))


def main():
    global done_upto
    path = sys.argv[1]
    e("v_mov_b32 %s, %%[kx]" % REG["kx"])
    e("v_mov_b32 %s, %%[ky]" % REG["ky"])
    e("v_mov_b32 %s, %%[pm]" % REG["pm"])
    e("v_mov_b32 %s, %%[pc]" % REG["pc"])
    e("v_mov_b32 %s, %%[vi]" % REG["vi"])
    e("v_mov_b32 %s, %%[xf]" % REG["xf"])
    e("v_mov_b32 %s, %%[l0]" % REG["a_ypl"])
    e("v_add_u32 %s, %s, %s" % (REG["iy0"], REG["a_ypl"], REG["ky"]))
    e("v_max_i32 %s, %s, %s" % (REG["iy0"], REG["iy0"], REG["a_ypl"]))                   # Iy' at column 1: max(M(r,0) - goe, Iy(r,0)) with both = NEG
    e("v_mov_b32 %s, %%[rc]" % REG["rc"])
    e("s_mov_b32 %s, %%[m]" % S_LEFT)
    lds_issue("ds_read_u8 %s, %s" % (REG["offA"], REG["rc"]), ("C0", "A"))               # rows 1, 2 (their codes land in the offset registers)
    lds_issue("ds_read_u8 %s, %s offset:1" % (REG["offB"], REG["rc"]), ("C0", "B"))
    lds_issue("ds_read_u8 %s, %s offset:2" % (REG["cA"], REG["rc"]), ("C", "A"))         # rows 3, 4
    lds_issue("ds_read_u8 %s, %s offset:3" % (REG["cB"], REG["rc"]), ("C", "B"))
    e("v_add_u32 %s, 2, %s" % (REG["rc"], REG["rc"]))
    lds_need(("C0", "B"))
    for o in ("offA", "offB"):
        e("v_and_b32 %s, 31, %s" % (REG[o], REG[o]))
        e("v_mul_u32_u24 %s, %d, %s" % (REG[o], ROWSTRIDE, REG[o]))
        e("v_add_u32 %s, %%[tb], %s" % (REG[o], REG[o]))
    for k in range(R):                                     # the ring's first fill, in the order the steady state issues it
        prefetch("A", k, False)
        if k >= 1:
            prefetch("B", k - 1, False)
    prefetch("B", R - 1, False)
    entry_state = queue_state()
    saved = (list(queue), done_upto)
    e("s_cmp_lt_u32 %s, 2" % S_LEFT)
    e("s_cbranch_scc1 5f")
    # ---- rows 1, 2
    block(True)
    loop_state = queue_state()
    e("s_sub_u32 %s, %s, 2" % (S_LEFT, S_LEFT))
    e("2:")
    e("s_cmp_lt_u32 %s, 2" % S_LEFT)
    e("s_cbranch_scc1 3f")
    # ---- rows r, r + 1 (r >= 3)
    block(False)
    # the loop body was generated from what is KNOWN after rows 1, 2; after itself it knows at least as much (the same reads in the same
    # order, possibly fewer of them still counted as outstanding): its waits hold on every iteration
    st = queue_state()
    assert len(st) <= len(loop_state) and loop_state[len(loop_state) - len(st):] == st, "the two-row block must leave the LDS queue as it found it"
    done_upto = len(queue) - len(loop_state)               # what follows the loop may be entered after rows 1, 2 directly: the weaker knowledge
    e("s_sub_u32 %s, %s, 2" % (S_LEFT, S_LEFT))
    e("s_branch 2b")
    e("3:")
    e("s_cmp_eq_u32 %s, 0" % S_LEFT)
    e("s_cbranch_scc1 9f")
    single(False)                                          # the last row of an odd count
    e("s_branch 9f")
    e("5:")                                                # m == 1
    queue[:] = saved[0]
    done_upto = saved[1]
    assert queue_state() == entry_state
    single(True)
    e("9:")
    e("s_waitcnt lgkmcnt(0)")                              # reads issued ahead for rows that do not exist
    with open(path, "w") as f:
        f.write("// generated by tools/gen_nw_asm.py (NMAX = %d, ring %d, %d VGPRs) -- do not edit\n" % (N, R, NREGS))
        for x in out:
            f.write('"%s\\n\\t"\n' % x)
    # the C++ binding that goes with it: register declarations, operand lists, clobbers
    hdr = os.path.splitext(path)[0] + "_bind.inc"
    with open(hdr, "w") as f:
        f.write("// generated by tools/gen_nw_asm.py (NMAX = %d) -- do not edit: the register binding of nw_rows_p%d.inc\n" % (N, N))
        f.write("#define NW_ASM_DECL_%d \\\n" % N)
        for c in range(N):
            f.write("  register int32_t nwvm%d asm(\"%s\"); \\\n" % (c, VM[c]))
        for c in range(N):
            f.write("  register uint32_t nwbo%d asm(\"%s\") = boff[%d]; \\\n" % (c, BO[c], c))
        f.write("\n")
        f.write("#define NW_ASM_OUTS_%d %s\n" % (N, ", ".join('"=v"(nwvm%d)' % c for c in range(N))))
        f.write("#define NW_ASM_INS_%d %s\n" % (N, ", ".join('"v"(nwbo%d)' % c for c in range(N))))
        f.write("#define NW_ASM_COPY_%d %s\n" % (N, " ".join("VM[%d] = nwvm%d;" % (c, c) for c in range(N))))
        clob = [r for r in ("v%d" % i for i in range(NREGS)) if r not in VM and r not in BO]
        f.write("#define NW_ASM_CLOBBERS_%d \"memory\", \"vcc\", \"scc\", \"%s\", %s\n" % (N, S_LEFT, ", ".join('"%s"' % r for r in clob)))
        f.write("#define NW_ASM_VGPRS_%d %d\n" % (N, NREGS))
    print("wrote %s (%d instructions, %d VGPRs) and %s" % (path, len(out), NREGS, hdr))


if __name__ == "__main__":
    main()
