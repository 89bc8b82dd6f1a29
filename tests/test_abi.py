"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol
include/dynaalign.h declares, validates like the reference, and refuses to compute
without a GPU (no CPU fallback).  No compute calls are made here."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(built):
    from dynaalign_amd import _capi
    return _capi.load()


def test_library_exports_every_declared_symbol(lib):
    from dynaalign_amd import _capi
    declared = _capi.header_symbols()
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(lib, name), name
    # the Python binding covers the header exactly
    assert sorted(_capi.SIGNATURES) == declared
    assert lib.da_abi_version() == 2


def test_no_torch_types_or_cxx_in_header():
    from dynaalign_amd import _capi
    import re
    txt = open(_capi.HEADER_PATH).read()
    assert 'extern "C"' in txt
    code = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)  # declarations only, comments stripped
    for banned in ("torch", "at::", "std::", "template", "class "):
        assert banned not in code, banned
    # the header compiles as plain C
    subprocess.check_call(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", _capi.HEADER_PATH])


def test_library_does_not_link_the_oracle(lib):
    from dynaalign_amd import _capi
    out = subprocess.check_output(["readelf", "-d", _capi.LIB_PATH]).decode()
    assert "liborc" not in out
    syms = subprocess.check_output(["nm", "-D", "--defined-only", _capi.LIB_PATH]).decode()
    assert "orc_" not in syms


def test_hash_family_seeds_match_mt19937(lib, kats):
    import dynaalign_amd as da
    for kat in kats["mt19937"]:
        if "first" in kat:
            assert da.hash_family_seeds(kat["seed"], len(kat["first"])).tolist() == kat["first"]
        else:
            assert int(da.hash_family_seeds(kat["seed"], kat["index"] + 1)[kat["index"]]) == kat["value"]
    assert np.array_equal(da.hash_family_seeds(12345, 500), O.seeds(12345, 500))
    assert len(da.hash_family_seeds(7, 0)) == 0


def test_synth_stream_is_mt19937():
    from dynaalign_amd import synth
    assert np.array_equal(synth.mt19937_raw(1, 700), O.seeds(1, 700))
    res, off = synth.uniform_peptides(10, 20, seed=7)
    draws = O.seeds(7, 200)
    assert bytes(res) == bytes(synth.AA20[draws % 20])
    assert off.tolist() == list(range(0, 201, 20))
    r2, o2 = synth.h3n2_like(50, 20)
    assert len(r2) == 1000 and set(bytes(r2)) <= set(b"ACDEFGHIKLMNPQRSTVWY")
    assert np.array_equal(r2, synth.h3n2_like(50, 20)[0])


def test_reference_error_messages_and_order(lib, kats):
    """Same validation order and texts as reference src/minHash.cpp:121-131 and
    src/pairwiseSeqAlign.cpp:204,242,249 -- raised before any device is needed."""
    import dynaalign_amd as da
    e = kats["mh_errors"]
    with pytest.raises(da.DynaAlignError, match="^" + e["empty"] + "$") as ei:
        da.similarityMH([], 0, 0)
    assert ei.value.code == 1
    with pytest.raises(da.DynaAlignError) as ei:
        da.similarityMH(["ACDE"], 0, 0)
    assert (ei.value.code, str(ei.value)) == (2, e["k"])
    with pytest.raises(da.DynaAlignError) as ei:
        da.similarityMH(["ACDE"], 4, -3)
    assert (ei.value.code, str(ei.value)) == (3, e["n_hash"])
    with pytest.raises(da.DynaAlignError) as ei:
        da.similarityNW(["AA"], "PAM250")
    assert (ei.value.code, str(ei.value)) == (4, kats["nw_bad_matrix"]["error"])
    # matrix name is checked before the (possibly empty) input, like the reference (:338)
    with pytest.raises(da.DynaAlignError):
        da.similarityNW([], "nope")
    assert da.similarityNW([]).shape == (0, 0)


@pytest.mark.parametrize("seqs", [
    ["AJ", "AA"], ["AA", "AJ"], ["JA", "AA"], ["", "AJ", "AA"], ["", "", "J"], ["AA", "CC", "AUA", "JJ"],
    ["AA", "a"], ["A A"], ["AC", "", "C1"], ["", "J", "AA"],
])
def test_nw_residue_errors_match_the_lazy_reference_order(lib, seqs):
    """The library validates up front but must raise what the reference's lazy row-major
    fill would raise first; the oracle implements that lazily."""
    import dynaalign_amd as da
    rc, _, msg = O.similarity_nw(seqs)
    assert rc in (O.ERR_BAD_RES1, O.ERR_BAD_RES2)
    with pytest.raises(da.DynaAlignError) as ei:
        da.similarityNW(seqs)
    assert (ei.value.code, str(ei.value)) == (rc, msg)


def test_nw_errors_fuzz(lib):
    import dynaalign_amd as da
    rng = np.random.RandomState(5)
    alpha = "ARNDCQEGHILKMFPSTWYVBZX*" * 3 + "JUO"
    for _ in range(200):
        seqs = ["".join(alpha[i] for i in rng.randint(0, len(alpha), rng.randint(0, 6))) for _ in range(rng.randint(1, 6))]
        rc, _, msg = O.similarity_nw(seqs)
        if rc == 0:
            continue
        with pytest.raises(da.DynaAlignError) as ei:
            da.similarityNW(seqs)
        assert (ei.value.code, str(ei.value)) == (rc, msg), seqs


def test_compute_fails_loudly_without_gpu(lib):
    import dynaalign_amd as da
    if lib.da_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(da.DynaAlignError) as ei:
        da.similarityMH(["ACDEFG", "ACDEFH"], 4, 8, seed=1)
    assert ei.value.code == 8 and "no CPU fallback" in str(ei.value)
    with pytest.raises(da.DynaAlignError) as ei:
        da.similarityNW(["ACDEFG", "ACDEFH"])
    assert ei.value.code == 8


def test_pack_sequences_layout():
    import dynaalign_amd as da
    res, off = da.pack_sequences(["AC", "", "DEF"])
    assert off.tolist() == [0, 2, 2, 5] and bytes(res) == b"ACDEF"
    res, off = da.pack_sequences([])
    assert off.tolist() == [0]


def test_product_package_never_imports_the_oracle():
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "dynaalign_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"oracle_lib|liborc|dynaalign_oracle|orc_", txt), os.path.join(dirpath, f)


def test_opts_struct_layout_and_rccl_binding(lib):
    """struct da_opts of the header == the ctypes mirror; RCCL can be bound in this image (dlopen of librccl.so.1);
    validation happens before any device work and keeps the reference's order"""
    from dynaalign_amd import _capi
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "dynaalign.h"
    int main(void) { printf("%zu %zu %zu %zu %zu %d %d %d %d\n", sizeof(da_opts), offsetof(da_opts, n_devices), offsetof(da_opts, devices),
                            offsetof(da_opts, exchange), offsetof(da_opts, phase_ms), DA_EXCHANGE_ROWS, DA_EXCHANGE_ALLGATHER,
                            DA_EXCHANGE_PEERCOPY, DA_PHASE_COUNT); return 0; }'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.dirname(_capi.HEADER_PATH), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        got = [int(v) for v in subprocess.check_output([os.path.join(d, "t")]).split()]
    D = _capi.DaOpts
    assert got == [C.sizeof(D), D.n_devices.offset, D.devices.offset, D.exchange.offset, D.phase_ms.offset,
                   _capi.DA_EXCHANGE["rows"], _capi.DA_EXCHANGE["allgather"], _capi.DA_EXCHANGE["peercopy"], len(_capi.DA_PHASES)]
    assert lib.da_rccl_available() == 1
    import dynaalign_amd as da
    for kw in ({}, {"devices": [0]}, {"devices": [0, 0], "exchange": "peercopy"}):
        with pytest.raises(da.DynaAlignError) as e:                      # reference order: empty -> k -> n_hash (src/minHash.cpp:121-131)
            da.similarityMH([], 0, 0, **kw)
        assert e.value.code == 1
        with pytest.raises(da.DynaAlignError) as e:
            da.similarityMH(["ACDEF"], 0, 0, **kw)
        assert e.value.code == 2
        with pytest.raises(da.DynaAlignError) as e:
            da.similarityNW(["ACDEF"], "PAM250", **kw)
        assert str(e.value) == "Invalid substitution matrix name: PAM250"
        with pytest.raises(da.DynaAlignError) as e:
            da.similarityNW(["AJ", "AA"], **kw)
        assert str(e.value) == "Invalid amino acid in sequence2: J"
    with pytest.raises(ValueError):
        da.similarityMH(["ACDEF"], 4, 50, devices=[0], exchange="ring")


def test_duplicate_route_entry_points_reject_bad_arguments(lib):
    """argument checks of the round-2 device entry points run before anything touches a GPU: NULL pointers, a da_unique_plan whose
    struct_size is too small, bad ranks / worlds / leading dimensions all come back as DA_ERR_BAD_ARG with a message"""
    import ctypes
    from dynaalign_amd import _capi
    BAD = _capi.DA_ERR_BAD_ARG
    plan = _capi.DaUniquePlan()
    plan.struct_size = ctypes.sizeof(_capi.DaUniquePlan)
    pp = ctypes.addressof(plan)
    assert lib.da_dev_unique_plan(None, None, 10, 100, None, 0, pp, None) == BAD and b"NULL" in lib.da_last_error()
    small = _capi.DaUniquePlan()
    small.struct_size = 8
    assert lib.da_dev_unique_plan(1, 1, 10, 100, 256, 1 << 30, ctypes.addressof(small), None) == BAD
    assert lib.da_dev_unique_plan(1, 1, 0, 0, 256, 1 << 30, pp, None) != 0          # empty input: the reference's message
    assert lib.da_dev_unique_plan_bytes(100000, 2000000) > 4 * 100000 * 10
    assert lib.da_dev_shards_to_table(None, 0, 100, 2, 9, None, 104, None) == BAD
    assert lib.da_dev_shards_to_table(1, 0, 100, 0, 9, 1, 104, None) == BAD          # world < 1
    assert lib.da_dev_shards_to_table(1, 0, 100, 2, 17, 1, 104, None) == BAD         # value_bits > 16
    assert lib.da_dev_expand_unique(1, 8, 1, pp, 0, 500, 0, None, 0, 1, 8, None) == BAD    # plan without device pointers
    plan.n, plan.unique, plan.d_uidx, plan.d_ufirst = 100, 40, 1, 1
    assert lib.da_dev_expand_unique(None, 40, 1, pp, 0, 500, 0, None, 0, 1, 100, None) == BAD
    assert lib.da_dev_expand_unique(1, 8, 1, pp, 0, 500, 0, None, 0, 1, 100, None) == BAD  # ld_table < unique
    assert lib.da_dev_expand_unique(1, 40, 0, pp, 0, 500, 0, None, 0, 1, 100, None) == BAD # table_world < 1
    assert lib.da_dev_nw_unique_rows(pp, 20, 0, 10, 4, 0, 1, 1, 40, None) == BAD           # plan lacks strings / block bounds
    assert lib.da_dev_unique_rows(None, 40, pp, None, None) == BAD
    assert lib.da_dev_upper_histogram_rows(1, 8, 1, 100, 501, 1, None) == BAD              # ld < n
    assert lib.da_dev_extract_edges_rows(1, 100, None, 100, 1, 501, 1, 1, 1, 1, 10, 1, None) == BAD
    assert lib.da_dev_expand_workspace_bytes(100000, 45000, 0, 500, 0) == 45000 * 100000 * 2
    assert lib.da_dev_expand_workspace_bytes(100000, 70000, 0, 500, 0) == 256              # > 65536 unique strings: no fast passes
    assert lib.da_dev_unique_rows_bytes(100001, 10) == 10 * 100008 * 2


def test_band_ranges_of_the_symmetric_tile_numbering():
    """host logic of the pipelined duplicate route (minhash_kernels.hip mh_sym_band_prefix / decode_tile, run on the host): the symmetric
    tiles are numbered band by band (8 tile rows), so that [prefix(b), prefix(b + 1)) is exactly band b's tiles right of or on the diagonal --
    every tile once, every id valid -- for full bands and a partial last band"""
    import ctypes
    from dynaalign_amd import _capi
    lib = _capi.load()
    lib.da_debug_sym_band_prefix.argtypes = [ctypes.c_int64, ctypes.c_int64]
    lib.da_debug_sym_band_prefix.restype = ctypes.c_int64
    lib.da_debug_decode_sym_tile.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    lib.da_debug_decode_sym_tile.restype = ctypes.c_int
    for n in (1, 127, 128, 129, 1024, 1025, 3000, 5 * 1024 + 7, 16 * 1024, 44931):
        T = -(-n // 128)
        bands = -(-T // 8)
        assert lib.da_debug_sym_band_prefix(n, 0) == 0 and lib.da_debug_sym_band_prefix(n, bands) == T * (T + 1) // 2
        assert lib.da_debug_sym_band_prefix(n, bands + 3) == T * (T + 1) // 2
        seen = set()
        ti, tj = ctypes.c_int(0), ctypes.c_int(0)
        for b in range(bands):
            lo, hi = lib.da_debug_sym_band_prefix(n, b), lib.da_debug_sym_band_prefix(n, b + 1)
            rows = min(8, T - 8 * b)
            assert hi - lo == sum(T - r for r in range(8 * b, 8 * b + rows))
            step = 1 if T <= 64 else 37                      # every id of the small cases, a stride through the large ones
            for L in sorted(set(range(lo, hi, step)) | {hi - 1}):
                assert lib.da_debug_decode_sym_tile(L, T, ctypes.byref(ti), ctypes.byref(tj)) == 1
                assert 8 * b <= ti.value < 8 * b + rows and ti.value <= tj.value < T
                if step == 1:
                    assert (ti.value, tj.value) not in seen
                    seen.add((ti.value, tj.value))
        if T <= 64:
            assert len(seen) == T * (T + 1) // 2


def test_switches_are_parsed_in_one_place_and_all_documented():
    """the library's DYNAALIGN_* switches are read once into one struct (api.cpp parse_config); no other product source
    calls getenv, and INTEGRATION.md section 9 lists exactly the parsed set"""
    import re
    csrc = os.path.join(ROOT, "dynaalign_amd", "csrc")
    api = open(os.path.join(csrc, "api.cpp")).read()
    body = api[api.index("Config parse_config() {"):api.index("std::atomic<const Config *> g_config")]
    parsed = set(re.findall(r'"(DYNAALIGN_[A-Z0-9_]+)"', body))
    assert len(parsed) >= 25
    for f in os.listdir(csrc):
        if not f.endswith((".cpp", ".hip", ".hpp")):
            continue
        src = open(os.path.join(csrc, f)).read()
        src = re.sub(r"#ifdef DA_K2_EXPERIMENTS.*?#endif", "", src, flags=re.S)          # (the experiment twin of the library, tools/experiments/)
        if f == "api.cpp":
            src = src.replace(api[api.index("bool env_flag("):api.index("std::atomic<const Config *> g_config")], "")
        assert "getenv(" not in src, f
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 9. Environment switches"):]
    table = sec[:sec.index("Read by the host layers")]
    documented = set()
    for m in re.finditer(r"`(DYNAALIGN_[A-Z0-9_]+)[=`]|`(_[A-Z_]+)`", table):
        documented.add(m.group(1) or ("DYNAALIGN_MH_PIPE" + m.group(2)))
    assert documented == parsed, (sorted(parsed - documented), sorted(documented - parsed))


def test_config_reload_hook_is_exported_and_harmless_without_a_gpu():
    from dynaalign_amd import _capi
    lib = _capi.load()
    lib.da_config_reload()
    os.environ["DYNAALIGN_MH_NO_DEDUP"] = "1"
    try:
        assert lib.da_device_count() >= 0                    # (the front end notices the changed environment and reloads by itself)
    finally:
        del os.environ["DYNAALIGN_MH_NO_DEDUP"]
    lib.da_config_reload()
