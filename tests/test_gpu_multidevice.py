"""GPU tests of the single-process multi-device entry points (da_similarity_mh_opts / _nw_opts, SURVEY 8(b) da_opts,
8(e)).  The box has ONE GPU, so:
  * DA_EXCHANGE_ROWS and DA_EXCHANGE_PEERCOPY are driven with the same device listed 1, 2, 3 and 5 times -- the host
    threads, the row split, the shard geometry, the exchange and the per-rank device-to-host copies all run for real,
    only the peer copies stay on one device;
  * DA_EXCHANGE_ALLGATHER (RCCL, ncclCommInitAll + ncclAllGather) runs with a one-device list -- RCCL refuses two
    ranks on one GPU.  A world > 1 all-gather has never executed on this pool.
Checker: the CPU oracle, float64 compared as uint64."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def da(built):
    import dynaalign_amd
    from dynaalign_amd import _capi
    assert _capi.load().da_device_count() > 0
    return dynaalign_amd


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


CASES = [("rows", [0]), ("rows", [0, 0]), ("rows", [0, 0, 0]), ("rows", [0] * 5), ("peercopy", [0]), ("peercopy", [0, 0]),
         ("peercopy", [0, 0, 0]), ("peercopy", [0] * 5), ("allgather", [0])]


@pytest.mark.parametrize("exchange,devices", CASES)
@pytest.mark.parametrize("n", [1, 130, 1000, 2600])
def test_mh_opts_matches_oracle(da, exchange, devices, n):
    from dynaalign_amd import synth, similarity
    seqs = synth.to_strings(*synth.h3n2_like(n, 20))
    rc, want = O.similarity_mh(seqs, 4, 200, O.seeds(777, 200))
    assert rc == 0
    got = da.similarityMH(seqs, 4, 200, seed=777, devices=devices, exchange=exchange)
    assert same(got, want)
    ph = dict(similarity.last_phase_ms)
    assert set(ph) == {"setup", "compute", "exchange", "finalize", "d2h", "total"} and ph["total"] > 0
    assert (ph["exchange"] > 0) == (exchange != "rows")


@pytest.mark.parametrize("exchange,devices", CASES)
@pytest.mark.parametrize("n,lens", [(1, (20, 20)), (200, (0, 30)), (700, (20, 20)), (1100, (5, 64))])
def test_nw_opts_matches_oracle(da, exchange, devices, n, lens):
    rng = np.random.RandomState(n)
    alpha = np.frombuffer(b"ARNDCQEGHILKMFPSTWYVBZX*", np.uint8)
    seqs = ["".join(map(chr, alpha[rng.randint(0, 24, rng.randint(lens[0], lens[1] + 1))])) for _ in range(n)]
    rc, want, _ = O.similarity_nw(seqs, "BLOSUM80", 7, 2)
    assert rc == 0
    got = da.similarityNW(seqs, "BLOSUM80", 7, 2, devices=devices, exchange=exchange)
    assert same(got, want)                                      # includes NaN bit patterns of empty-vs-empty pairs


def test_opts_null_is_the_plain_entry_point(da):
    import ctypes
    from dynaalign_amd import _capi, synth
    lib = _capi.load()
    res, off = synth.h3n2_like(300, 20)
    seeds = da.hash_family_seeds(5, 64)
    a, b = np.empty((300, 300)), np.empty((300, 300))
    _capi.check(lib.da_similarity_mh(res.ctypes.data, off.ctypes.data, 300, 4, 64, seeds.ctypes.data, a.ctypes.data))
    _capi.check(lib.da_similarity_mh_opts(res.ctypes.data, off.ctypes.data, 300, 4, 64, seeds.ctypes.data, b.ctypes.data, None))
    assert same(a, b)


def test_opts_errors(da):
    from dynaalign_amd import synth
    seqs = synth.to_strings(*synth.h3n2_like(50, 20))
    with pytest.raises(da.DynaAlignError, match="not present"):
        da.similarityMH(seqs, 4, 50, seed=1, devices=[99])
    with pytest.raises(da.DynaAlignError, match="listed twice"):
        da.similarityMH(seqs, 4, 50, seed=1, devices=[0, 0], exchange="allgather")
    with pytest.raises(da.DynaAlignError, match="up to 64 residues"):
        da.similarityNW(["A" * 70, "C" * 3], devices=[0, 0], exchange="peercopy")
    got = da.similarityNW(["A" * 70, "C" * 3], devices=[0, 0], exchange="rows")       # ROWS has no such limit
    rc, want, _ = O.similarity_nw(["A" * 70, "C" * 3])
    assert same(got, want)


def test_forced_row_block_streaming_in_rows_mode(da, monkeypatch):
    """a device whose share does not fit its memory budget streams it in row blocks"""
    from dynaalign_amd import synth
    monkeypatch.setenv("DYNAALIGN_BLOCK_BYTES", str(2 * 1024 * 1024))
    seqs = synth.to_strings(*synth.h3n2_like(1500, 20))
    rc, want = O.similarity_mh(seqs, 4, 100, O.seeds(3, 100))
    got = da.similarityMH(seqs, 4, 100, seed=3, devices=[0, 0, 0])
    assert same(got, want)


def test_failed_allgather_call_drops_its_cached_communicators(da):
    """the RCCL communicators of a device list stay cached between calls; a call in which a rank failed must not leave a
    communicator that saw an abandoned collective behind for the next call.  The test hook makes rank 0 fail before the collective;
    every rank then skips the exchange, the list's cache entry is destroyed, and the next call builds new communicators and works."""
    import ctypes
    from dynaalign_amd import _capi, synth
    lib = _capi.load()
    st = (ctypes.c_size_t * 2)()
    seqs = synth.to_strings(*synth.h3n2_like(400, 20))
    rc, want = O.similarity_mh(seqs, 4, 100, O.seeds(3, 100))
    assert rc == 0
    assert same(da.similarityMH(seqs, 4, 100, seed=3, devices=[0], exchange="allgather"), want)
    _capi.check(lib.da_debug_comm_cache_state(0, ctypes.addressof(st)))
    cached, evicted = st[0], st[1]
    assert cached >= 1
    _capi.check(lib.da_debug_comm_cache_state(1, ctypes.addressof(st)))
    with pytest.raises(da.DynaAlignError, match="forced exchange failure"):
        da.similarityMH(seqs, 4, 100, seed=3, devices=[0], exchange="allgather")
    _capi.check(lib.da_debug_comm_cache_state(0, ctypes.addressof(st)))
    assert (st[0], st[1]) == (cached - 1, evicted + 1)
    assert same(da.similarityMH(seqs, 4, 100, seed=3, devices=[0], exchange="allgather"), want)
    _capi.check(lib.da_debug_comm_cache_state(0, ctypes.addressof(st)))
    assert (st[0], st[1]) == (cached, evicted + 1)
