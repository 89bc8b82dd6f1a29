#!/usr/bin/env python3
"""CPU model of tools/experiments/nw_rows_p<NMAX>.inc (the hand-scheduled DP rows of k_nw_short, tools/gen_nw_asm.py; experiment library only).

Interprets the generated instruction stream for ONE lane: scalar control flow, the VALU subset the block uses, LDS reads.  LDS reads
return IN ORDER and -- adversarially -- only when a counted `s_waitcnt lgkmcnt(k)` forces them to: a read's destination register is
poisoned from issue until then, and any instruction that touches a poisoned register fails the run.  So the model checks what the GPU
cannot show reliably: every counted wait of the table-read ring and of the row-ahead residue reads is sufficient on every path
(m = 1, even and odd row counts), besides the register map and the recurrence itself.

Reference: `rows_reference` restates nw_row_ck (nw_kernels.hip) cell by cell; `decode` turns the last row's combined key into
(matches, length, score) the way the kernel does, and tests compare that with tests/nw_model.py (independent, traceback-free) and
through it with the oracle.  Used by tests/test_nw_asm_model.py (no GPU needed)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S, S2, LB, NEG = 13, 15, 7, -24000          # CKBits<NMAX <= 32> (nw_kernels.hip)
MASK = 0xFFFFFFFF


def s32(x):
    x &= MASK
    return x - (1 << 32) if x & 0x80000000 else x


def constants(go, ge):
    goe = go + ge
    return {"kx": (((ge - goe) << S2) + (1 << S)) & MASK, "ky": ((ge - goe) << S2) & MASK, "pm": (1 << S) - 1, "pc": (~(3 << S)) & MASK,
            "vi": ((ge - go) << S2) & MASK, "l0": (NEG << S2) & MASK, "xf": (((NEG - min(goe, ge)) << S2) + (1 << S)) & MASK}


def key_table(table, ge):
    """tabk of k_nw_short<CK>: (score + 2 ge) << S2 | priority 2 | D + 1 | match << LB"""
    return [((((table[e] + 2 * ge) << S2) + (2 << S) + 1 + ((1 << LB) if e // 24 == e % 24 else 0)) & MASK) for e in range(576)]


def rows_reference(a_codes, b_codes, nmax, tabk, go, ge):
    """nw_row_ck over rows 1..m for one lane: returns VM (list of nmax uint32)"""
    k = constants(go, ge)
    boff = [(b_codes[c] if c < len(b_codes) else 0) for c in range(nmax)]
    VM = [k["vi"]] * nmax
    XP = [0] * nmax
    for r in range(1, len(a_codes) + 1):
        row = a_codes[r - 1] * 24
        vmd = 0 if r == 1 else k["vi"]
        vml = ypl = k["l0"]
        for c in range(nmax):
            e = tabk[row + boff[c]]
            ixf = k["xf"] if r == 1 else max(s32(VM[c] + k["kx"]), s32(XP[c])) & MASK
            iyf = max(s32(vml + k["ky"]), s32(ypl)) & MASK
            vd = (vmd + e) & MASK
            w = max(s32(vd), s32(ixf), s32(iyf)) & MASK
            vmn = w & k["pc"]
            vmd = VM[c]
            VM[c] = vmn
            XP[c] = (vmn & k["pm"]) | (ixf & ~k["pm"] & MASK)
            ypl = (vmn & k["pm"]) | (iyf & ~k["pm"] & MASK)
            vml = vmn
    return VM


def decode(vm, m, nj, ge):
    """(matches, length, score) of cell (m, nj) from the combined key, as k_nw_short does"""
    mt = (vm >> LB) & ((1 << (S - LB)) - 1)
    ln = m + nj - (vm & ((1 << LB) - 1))
    sc = (s32(vm) >> S2) - (m + nj) * ge
    return mt, ln, sc


class Hazard(Exception):
    pass


def load(inc):
    lines = []
    for l in open(inc):
        m = re.match(r'^"(.*)\\n\\t"$', l.rstrip("\n"))
        if m:
            lines.append(m.group(1))
    return lines


def run(inc, nmax, a_codes, b_codes, tabk, go, ge, rc_pad=None, max_steps=2_000_000):
    """executes the block for one lane; returns (VM list, stats).  LDS image: key table at TB, the row's residue codes at RC
    (zero-padded to nmax, followed by `rc_pad` bytes standing for whatever lies behind the array)"""
    prog = load(inc)
    labels = {}
    for i, l in enumerate(prog):
        m = re.match(r"^(\d+):$", l)
        if m:
            labels.setdefault(m.group(1), []).append(i)
    TB, RC = 0x4000, 0x100
    lds = bytearray(0x6000)
    for e, v in enumerate(tabk):
        lds[TB + 4 * e:TB + 4 * e + 4] = int(v).to_bytes(4, "little")
    codes = list(a_codes) + [0] * (nmax - len(a_codes)) + list(rc_pad if rc_pad is not None else [0xEE, 0x17, 0xFF, 0x05, 0x99, 0x1F])
    lds[RC:RC + len(codes)] = bytes(codes)
    k = constants(go, ge)
    ops = dict(k, m=len(a_codes), rc=RC, tb=TB)
    v, s = {}, {}
    nv = int(re.search(r"(\d+) VGPRs", open(inc).readline()).group(1))
    bind = open(os.path.splitext(inc)[0] + "_bind.inc").read()           # the register binding generated with the block
    bo = {int(c): int(r) for c, r in re.findall(r'nwbo(\d+) asm\("v(\d+)"\)', bind)}
    vm = {int(c): int(r) for c, r in re.findall(r'nwvm(\d+) asm\("v(\d+)"\)', bind)}
    assert sorted(bo) == sorted(vm) == list(range(nmax))
    for c in range(nmax):
        v[bo[c]] = 4 * (b_codes[c] if c < len(b_codes) else 0)
    pending = []            # (dst vgpr, value) in issue order
    poisoned = {}
    scc = 0
    stats = {"inst": 0, "valu": 0, "lds": 0, "waits": 0}

    def src(tok, reading=True):
        tok = tok.strip()
        if tok.startswith("%["):
            return ops[tok[2:-1]] & MASK
        if tok[0] == "v":
            r = int(tok[1:])
            if r in poisoned:
                raise Hazard("v%d read while its LDS read may still be in flight (pc %d: %s)" % (r, pc, prog[pc]))
            if r not in v:
                raise Hazard("v%d read before written (pc %d: %s)" % (r, pc, prog[pc]))
            return v[r]
        if tok[0] == "s":
            return s[int(tok[1:])]
        return int(tok, 0) & MASK

    def dst(tok, val):
        r = int(tok.strip()[1:])
        if tok.strip()[0] == "v":
            if r in poisoned:
                raise Hazard("v%d written while its LDS read may still be in flight (pc %d: %s)" % (r, pc, prog[pc]))
            assert r < nv, "register v%d beyond the block's map" % r
            v[r] = val & MASK
        else:
            s[r] = val & MASK

    pc = 0
    steps = 0
    while pc < len(prog):
        steps += 1
        if steps > max_steps:
            raise Hazard("runaway")
        l = prog[pc]
        if re.match(r"^\d+:$", l):
            pc += 1
            continue
        stats["inst"] += 1
        op, _, rest = l.partition(" ")
        a = [x.strip() for x in rest.split(",")] if rest else []
        if op == "s_waitcnt":
            kk = int(re.search(r"lgkmcnt\((\d+)\)", l).group(1))
            stats["waits"] += 1
            while len(pending) > kk:
                d, val = pending.pop(0)
                poisoned[d] -= 1
                if poisoned[d] == 0:
                    del poisoned[d]
                v[d] = val
        elif op in ("ds_read_b32", "ds_read_u8"):
            m = re.match(r"(\S+),\s*(\S+)(?:\s+offset:(\d+))?$", rest)
            addr = (src(m.group(2)) + int(m.group(3) or 0)) & MASK
            assert addr + 4 <= len(lds), "LDS read at %#x outside the image (pc %d: %s)" % (addr, pc, l)
            val = int.from_bytes(lds[addr:addr + 4], "little") if op == "ds_read_b32" else lds[addr]
            if op == "ds_read_b32":
                assert addr % 4 == 0
            d = int(m.group(1)[1:])
            if d in poisoned and op == "ds_read_b32":
                raise Hazard("v%d is the target of two reads in flight (pc %d)" % (d, pc))
            pending.append((d, val))
            poisoned[d] = poisoned.get(d, 0) + 1
            v.pop(d, None)
            stats["lds"] += 1
        elif op == "v_mov_b32":
            dst(a[0], src(a[1])); stats["valu"] += 1
        elif op == "v_add_u32":
            dst(a[0], src(a[1]) + src(a[2])); stats["valu"] += 1
        elif op == "v_max_i32":
            dst(a[0], max(s32(src(a[1])), s32(src(a[2])))); stats["valu"] += 1
        elif op == "v_max3_i32":
            dst(a[0], max(s32(src(a[1])), s32(src(a[2])), s32(src(a[3])))); stats["valu"] += 1
        elif op == "v_and_b32":
            dst(a[0], src(a[1]) & src(a[2])); stats["valu"] += 1
        elif op == "v_mul_u32_u24":
            dst(a[0], (src(a[1]) & 0xFFFFFF) * (src(a[2]) & 0xFFFFFF)); stats["valu"] += 1
        elif op == "v_bitop3_b32":
            m = re.match(r"(\S+),\s*(\S+),\s*(\S+),\s*(\S+)\s+bitop3:(\S+)$", rest)
            tt = int(m.group(5), 0)
            x, y, z = src(m.group(2)), src(m.group(3)), src(m.group(4))
            r = 0
            for idx in range(8):
                if (tt >> idx) & 1:
                    r |= (x if idx & 4 else ~x) & (y if idx & 2 else ~y) & (z if idx & 1 else ~z)
            dst(m.group(1), r); stats["valu"] += 1
        elif op == "s_mov_b32":
            dst(a[0], src(a[1]))
        elif op == "s_sub_u32":
            dst(a[0], src(a[1]) - src(a[2]))
        elif op == "s_cmp_lt_u32":
            scc = 1 if src(a[0]) < src(a[1]) else 0
        elif op == "s_cmp_eq_u32":
            scc = 1 if src(a[0]) == src(a[1]) else 0
        elif op in ("s_cbranch_scc1", "s_cbranch_scc0", "s_branch"):
            take = op == "s_branch" or (scc == 1) == (op == "s_cbranch_scc1")
            if take:
                lab, way = a[0][:-1], a[0][-1]
                cands = labels[lab]
                pc = min(i for i in cands if i > pc) if way == "f" else max(i for i in cands if i < pc)
                continue
        else:
            raise Hazard("instruction not modelled: " + l)
        pc += 1
    if pending:
        raise Hazard("%d LDS reads still in flight at the end of the block" % len(pending))
    for c in range(nmax):                                         # the inputs must survive the block
        assert v[bo[c]] == 4 * (b_codes[c] if c < len(b_codes) else 0)
    clob = set(int(r) for r in re.findall(r'"v(\d+)"', re.search(r"#define NW_ASM_CLOBBERS_\d+ (.*)", bind).group(1)))
    touched = set(v) | set(poisoned)
    assert touched <= clob | set(bo.values()) | set(vm.values()), "the block writes registers its clobber list does not name: %s" % sorted(touched - clob - set(bo.values()) - set(vm.values()))
    return [v[vm[c]] for c in range(nmax)], stats


def inc_path(nmax):
    return os.path.join(ROOT, "tools", "experiments", "nw_rows_p%d.inc" % nmax)
