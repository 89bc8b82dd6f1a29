"""The hand-scheduled DP rows of k_nw_short (tools/gen_nw_asm.py -> tools/experiments/nw_rows_p<NMAX>.inc; measured no faster than the
compiled row and therefore built into the experiment twin of the library only) on the CPU: the generated instruction stream
is interpreted for one lane by tools/sim_nw_asm.py -- register map, two-row skew, LDS addressing, and every wait (LDS reads return in
order and, in the model, only when a wait forces them) -- against nw_row_ck restated cell by cell, and the decoded (matches, length,
score) against the independent traceback-free model (tests/nw_model.py) and the C oracle.  No GPU needed.

Also here: the build-time checks ADVICE r3 asked for on the compiled direct sweep's row-ahead residue read (the value in flight lives in
v127 and nothing but its own asm statements names that register)."""
import json
import os
import random
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import nw_model  # noqa: E402
import sim_nw_asm as S  # noqa: E402

CSRC = os.path.join(ROOT, "dynaalign_amd", "csrc")
EXP = os.path.join(ROOT, "tools", "experiments")
TABLES = json.load(open(os.path.join(ROOT, "tests", "golden", "blosum_tables.json")))["tables"]
PENALTIES = [(10, 4), (0, 0), (3, 1), (11, 1), (5, 5), (40, 20)]


def gen(tmp_path, nmax, **env):
    out = tmp_path / ("nw_rows_p%d.inc" % nmax)
    e = {k: v for k, v in os.environ.items() if not k.startswith("NWASM_")}
    e.update({"NWASM_NMAX": str(nmax)}, **{k: str(v) for k, v in env.items()})
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_nw_asm.py"), str(out)], env=e, stdout=subprocess.DEVNULL)
    return str(out)


@pytest.mark.parametrize("nmax", [12, 20])
def test_committed_includes_are_the_generators_output(tmp_path, nmax):
    out = gen(tmp_path, nmax)
    assert open(out).read() == open(os.path.join(EXP, "nw_rows_p%d.inc" % nmax)).read()
    assert open(out.replace(".inc", "_bind.inc")).read() == open(os.path.join(EXP, "nw_rows_p%d_bind.inc" % nmax)).read()


def cases(rnd, nmax, count):
    for _ in range(count):
        go, ge = rnd.choice(PENALTIES)
        m = rnd.choice([1, 2, 3, 4, nmax - 1, nmax, rnd.randint(1, nmax)])
        nj = rnd.choice([0, 1, nmax, rnd.randint(0, nmax)])
        alpha = rnd.choice([20, 3, 24])                      # standard residues / low complexity (forces gaps) / B Z X * as well
        yield go, ge, [rnd.randrange(alpha) for _ in range(m)], [rnd.randrange(alpha) for _ in range(nj)]


@pytest.mark.parametrize("nmax", [12, 20])
def test_committed_block_equals_the_compiled_row_and_the_models(nmax):
    """every path of the block (m = 1; even and odd row counts; the two-row loop entered 0 .. N/2 - 1 times), ragged sequence2 lengths
    incl. empty, six penalty pairs, three alphabets, two matrices: all N combined keys of the last row, and the decoded result of the
    lane's own column against the traceback-free model and the oracle's DP + traceback"""
    import oracle_lib as O
    rnd = random.Random(nmax)
    inc = S.inc_path(nmax)
    for name in ("BLOSUM62", "BLOSUM100"):
        tab = TABLES[name]["values"]
        for go, ge, a, b in cases(rnd, nmax, 70):
            tabk = S.key_table(tab, ge)
            got, stats = S.run(inc, nmax, a, b, tabk, go, ge)
            assert got == S.rows_reference(a, b, nmax, tabk, go, ge), (nmax, len(a), len(b), go, ge)
            if b:
                sa, sb = "".join(nw_model.ORDER[x] for x in a), "".join(nw_model.ORDER[x] for x in b)
                res = S.decode(got[len(b) - 1], len(a), len(b), ge)
                assert res == nw_model.nw_identity(sa, sb, tab, go, ge)
                rc, mt, ln, sc, _ = O.nw_pair(sa, sb, name, go, ge)
                assert rc == 0 and res == (mt, ln, sc)


@pytest.mark.parametrize("nmax,env", [(20, {"NWASM_WAITS": "counted", "NWASM_RING": 10}), (20, {"NWASM_WAITS": "counted", "NWASM_RING": 5}),
                                      (20, {"NWASM_WAITS": "counted"}), (20, {"NWASM_PARITY": 0, "NWASM_ILV": 0}), (12, {"NWASM_WAITS": "counted", "NWASM_RING": 6}),
                                      (8, {}), (16, {"NWASM_WAITS": "counted", "NWASM_RING": 8}), (24, {}), (32, {"NWASM_RING": 4})])
def test_generator_variants(tmp_path, nmax, env):
    """the forms round 4 timed (profiles/r04_c_*): deep rings with COUNTED waits -- the model returns a read only when a wait forces it,
    so an insufficient count fails here, not on the GPU -- the plain register map, other NMAX"""
    inc = gen(tmp_path, nmax, **env)
    rnd = random.Random(7 * nmax)
    tab = TABLES["BLOSUM62"]["values"]
    for go, ge, a, b in cases(rnd, nmax, 40):
        tabk = S.key_table(tab, ge)
        got, _ = S.run(inc, nmax, a, b, tabk, go, ge)
        assert got == S.rows_reference(a, b, nmax, tabk, go, ge)


def test_the_model_catches_a_missing_wait(tmp_path):
    """the checker checks: with the waits of the counted form weakened by one, some read is used while it may still be in flight"""
    inc = gen(tmp_path, 20, NWASM_WAITS="counted", NWASM_RING=10)
    txt = open(inc).read()
    weak = re.sub(r"lgkmcnt\((\d+)\)", lambda m: "lgkmcnt(%d)" % min(15, int(m.group(1)) + 1) if int(m.group(1)) not in (0, 15) else m.group(0), txt)
    assert weak != txt
    open(inc, "w").write(weak)
    tab = TABLES["BLOSUM62"]["values"]
    with pytest.raises(S.Hazard):
        for go, ge, a, b in cases(random.Random(3), 20, 20):
            S.run(inc, 20, a, b, S.key_table(tab, ge), go, ge)


def test_block_reads_stay_inside_the_staged_arrays():
    """residue codes are read a block ahead, up to three bytes past the row's m codes: past a 20-residue row that is whatever lies behind
    rowcodes[lr][] in LDS -- the block masks the code (& 31), so the table address stays within 31 * 96 + 23 * 4 bytes of the table"""
    tab = TABLES["BLOSUM62"]["values"]
    a, b = [19] * 20, [3] * 20
    for pad in ([0xFF] * 6, [0x00] * 6, [0x7F, 0x80, 0x1F, 0x20, 0xE0, 0x55]):
        got, _ = S.run(S.inc_path(20), 20, a, b, S.key_table(tab, 4), 10, 4, rc_pad=pad)
        assert got == S.rows_reference(a, b, 20, S.key_table(tab, 4), 10, 4)


@pytest.fixture(scope="module")
def nw_disassembly(tmp_path_factory):
    out = tmp_path_factory.mktemp("nw") / "nw_kernels.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-fwrapv", "--offload-arch=gfx950", "-Wno-unused-function", "-DDA_K2_EXPERIMENTS", "-I", CSRC, "-I", EXP, "-x", "hip",
                           "--cuda-device-only", "-S", os.path.join(CSRC, "nw_kernels.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    text = open(out).read()
    kernels = {}
    for m in re.finditer(r"^(_ZN2da12_GLOBAL__N_110k_nw_shortILi(\d+)ELb([01])ELb([01])ELb([01])ELb([01])E\w+):.*?^\.Lfunc_end\d+:", text, re.S | re.M):
        if m.group(6) == "1":
            continue                                                          # (the prefix-sharing instances of the ordered mode: see the test below)
        kernels[(int(m.group(2)), m.group(3) == "1", m.group(4) == "1", m.group(5) == "1")] = (m.group(1), m.group(0))
    pfx = {}
    for m in re.finditer(r"^(_ZN2da12_GLOBAL__N_110k_nw_shortILi(\d+)ELb1ELb1ELb0ELb1E\w+):.*?^\.Lfunc_end\d+:", text, re.S | re.M):
        pfx[int(m.group(2))] = (m.group(1), m.group(0))
    kernels["pfx"] = pfx
    meta = {}
    for (key, (name, _)) in [kv for kv in kernels.items() if kv[0] != "pfx"] + [(("pfx", n_), v_) for n_, v_ in pfx.items()]:
        meta[key] = {k: int(re.search(r"\.set %s\.%s, (\d+)" % (re.escape(name), k), text).group(1)) for k in ("num_vgpr", "private_seg_size")}
    return kernels, meta


def test_direct_sweeps_row_ahead_read_owns_its_register(nw_disassembly):
    """the residue code of the next row is in flight between two asm statements of the compiled direct sweep.  It
    lives in v127, and no instruction outside those statements may name v127 (a copy, a spill or a reuse would read or clobber it
    before the wait)"""
    kernels, meta = nw_disassembly
    checked = 0
    for key, val in kernels.items():
        if key == "pfx":
            continue
        (nmax, ck, ordered, asm), (name, body) = key, val
        if not ck or ordered or asm or nmax > 24:
            continue
        lines = body.split("\n")
        inside = False
        for l in lines:
            if "#ASMSTART" in l:
                inside = True
            elif "#ASMEND" in l:
                inside = False
            elif re.search(r"\bv127\b|v\[\d+:127\]|v\[12[0-7]:1[23]\d\]", l) and not l.strip().startswith(";"):
                assert inside, "%s: v127 named outside the row-ahead read's statements: %s" % (name, l.strip())
        assert sum("ds_read_u8 v127" in l for l in lines) == 2, name       # the first row's read + the read inside the row loop
        assert any("v_readfirstlane_b32" in l and "v127" in l for l in lines)
        checked += 1
    assert checked >= 5                                                       # NMAX 8, 12, 16, 20, 24


def test_generated_row_kernels_use_no_scratch_and_fit_four_waves(nw_disassembly):
    kernels, meta = nw_disassembly
    asm_kernels = [k for k in kernels if k != "pfx" and k[3]]
    assert sorted(set(k[0] for k in asm_kernels)) == [12, 20] and all(k[1] for k in asm_kernels)       # combined key; ordered mode and direct sweep
    for k in asm_kernels:
        assert meta[k]["num_vgpr"] <= 128 and (meta[k]["private_seg_size"] == 0 or not k[2]), (k, meta[k])   # (the direct sweep spills 8 bytes around the block)
        assert sum("v_max3_i32" in l for l in kernels[k][1].split("\n")) >= 4 * k[0]


def _row_loop(body):
    """the innermost loop of a k_nw_short instance that holds one DP row (NMAX v_max3_i32): (first line, last line) of the disassembly"""
    lines = body.split("\n")
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    best = None
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a, b = labels[m.group(1)], i
            if sum("v_max3_i32" in x for x in lines[a:b]) >= 8 and (best is None or b - a < best[1] - best[0]):
                best = (a, b)
    return lines, best


def test_ordered_kernels_fit_five_waves_without_spills_in_the_row_loop(nw_disassembly):
    """round 4b: the ordered mode (plain and prefix-sharing) runs five waves per SIMD -- 96 VGPRs: sequence2's table offsets four to a register (SDWA byte
    select in the address add), wave-uniform loop state in SGPRs, the checkpoint's Ix' part as byte deltas (25 words: 30 KB of LDS per workgroup).  What may
    not happen again: a scratch access inside the DP row loop (the first five-wave build reloaded the residue address every row: 92.5 ms instead of 82.5)"""
    kernels, meta = nw_disassembly
    checked = 0
    for nmax, (name, body) in sorted(kernels["pfx"].items()):
        vg = meta[("pfx", nmax)]["num_vgpr"]
        assert vg <= 96, (name, vg)
        lines, loop = _row_loop(body)
        assert loop is not None, name
        assert not any("scratch_" in l for l in lines[loop[0]:loop[1] + 1]), name
        assert sum("v_add_u32_sdwa" in l for l in lines[loop[0]:loop[1] + 1]) >= nmax, name        # the packed offsets are unpacked inside the loop, not hoisted
        checked += 1
    assert checked >= 4                                                                            # NMAX 8, 12, 16, 20
    for (key, val) in kernels.items():
        if key == "pfx":
            continue
        (nmax, ck, ordered, asm), (name, body) = key, val
        if ck and ordered and not asm and nmax <= 24:
            assert meta[key]["num_vgpr"] <= 96, (name, meta[key])
            lines, loop = _row_loop(body)
            assert loop is not None and not any("scratch_" in l for l in lines[loop[0]:loop[1] + 1]), name
