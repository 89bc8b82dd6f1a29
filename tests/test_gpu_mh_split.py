"""GPU tests of the heavy / rare split of similarityMH's compare (api.cpp build_planes / mh_full_symmetric, dict_kernels.hip k_hy_split,
minhash_kernels.hip k_mh_compare_p12<.., 8, ..> + k_sp_* lists + k_hy_fixup*): per hash function the 254 most frequent signature values get dense
codes on EIGHT bit planes -- every other value reads as "never equal" there -- and the matching incidences of the remaining repeated values are
enumerated from lists and added to the dense result.  matches(i, j) = sum over h of [sig equal] (reference src/minHash.cpp:168-173) is a sum over
hash functions, so splitting the values of each column into two classes and counting the classes separately is exact; these tests hold the result
bit for bit against the oracle and against the full-width compare, whole matrices, on every route the split serves."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)


@pytest.fixture(scope="module")
def da(built):
    import dynaalign_amd
    from dynaalign_amd import _capi
    assert _capi.load().da_device_count() > 0
    return dynaalign_amd


class env:
    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def oracle(seqs, k, n_hash, seed=12345):
    rc, m = O.similarity_mh(seqs, k, n_hash, O.seeds(seed, n_hash))
    assert rc == 0
    return np.asarray(m)


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def run(seqs, k, n_hash, seed=12345, out=None, **switches):
    import torch
    from dynaalign_amd import device
    import dynaalign_amd as da_
    res, off = O.pack(seqs)
    ds = device.DeviceSequences(np.asarray(res, np.uint8), np.asarray(off, np.int64))
    seeds = da_.hash_family_seeds(seed, n_hash)
    with env(**switches):
        got = device.similarity_mh(ds, k, n_hash, seeds, out=out)
        torch.cuda.synchronize()
        route = device.mh_last_route()
    return got, route


def clustered(rng, n, parents=6, parent_len=400, length=20, mut=0.03, ragged=False):
    """windows of a few random parents with point mutations (what synth.h3n2_like draws), optionally of ragged length"""
    par = AA[rng.randint(0, 20, (parents, parent_len))]
    out = []
    for _ in range(n):
        L = length if not ragged else int(rng.randint(0, length + 8))
        p = rng.randint(0, parents)
        s0 = rng.randint(0, parent_len - L + 1)
        w = par[p, s0:s0 + L].copy()
        hit = rng.rand(L) < mut
        w[hit] = AA[rng.randint(0, 20, int(hit.sum()))]
        out.append(w.tobytes().decode("latin-1"))
    return out


def low_complexity(rng, n, letters, length=20):
    return ["".join(chr(c) for c in AA[rng.randint(0, letters, length)]) for _ in range(n)]


SPLIT = dict(DYNAALIGN_MH_HYBRID_MIN_N=256, DYNAALIGN_MH_NO_DEDUP=1, DYNAALIGN_MH_NO_SPARSE=1)


@pytest.mark.parametrize("name", ["clustered", "clustered_ragged", "mixed_with_low_complexity", "uniform", "border_n"])
def test_split_direct_route_whole_matrix_against_the_oracle(da, name):
    """the direct route (hand-scheduled 8-plane kernel on interior tiles, compiled 8-plane kernel on diagonal / border tiles, lists + both
    fix-up kernels): every element of the float64 matrix against the oracle's; the full-width compare gives the same bits"""
    rng = np.random.RandomState(len(name))
    if name == "clustered":
        seqs = clustered(rng, 6000)
    elif name == "clustered_ragged":
        seqs = clustered(rng, 5000, ragged=True)
    elif name == "mixed_with_low_complexity":     # a 7-letter alphabet has 2401 4-mers: large classes just outside the heavy set + ties at its threshold
        seqs = clustered(rng, 5000) + low_complexity(rng, 60, 7) + low_complexity(rng, 30, 5)
        rng.shuffle(seqs)
    elif name == "uniform":
        seqs = low_complexity(rng, 6000, 20)
    else:
        seqs = clustered(rng, 4000 + 77)           # n not a multiple of 128: the last tile column goes through the compiled kernel
    got, route = run(seqs, 4, 500, **SPLIT)
    assert not route["dedup"] and not route["sparse"]
    assert route["split"] and route["plane_bits"] == 8 and route["plane_bits_without"] >= 12, route
    want = oracle(seqs, 4, 500)
    assert same(got.cpu().numpy(), want)
    full, froute = run(seqs, 4, 500, **dict(SPLIT, DYNAALIGN_MH_NO_HYBRID=1))
    assert not froute["split"] and froute["plane_bits"] >= 12
    assert same(full.cpu().numpy(), want)
    if name in ("clustered", "mixed_with_low_complexity"):
        assert route["rare_pairs"] > 0


def test_split_really_moves_incidences_to_the_lists(da):
    """the fix-up is not a no-op: without it the dense half alone is short by exactly the rare incidences (counted from the oracle's signatures)"""
    rng = np.random.RandomState(5)
    seqs = clustered(rng, 5000)
    n_hash = 500
    sig = np.asarray(O.signatures(seqs, 4, n_hash, O.seeds(12345, n_hash))).reshape(len(seqs), n_hash)
    rare = 0
    for h in range(n_hash):
        _, cnt = np.unique(sig[:, h], return_counts=True)
        cnt = np.sort(cnt[cnt >= 2])[::-1].astype(np.int64)
        rare += int((cnt[254:] * (cnt[254:] - 1) // 2).sum())
    _, route = run(seqs, 4, n_hash, **SPLIT)
    assert route["split"] and route["rare_pairs"] == rare and rare > 0


@pytest.mark.parametrize("case", ["odd_ld", "even_padded_ld", "n_hash_600", "n_hash_2047", "n_hash_40", "n_hash_33"])
def test_split_shapes_at_the_edges_of_the_hand_scheduled_kernel(da, case):
    """odd leading dimension: every tile through the compiled 8-plane kernel, then the same fix-up; a padded even one, the largest n_hash the 8-plane kernel's
    float64 table holds (2047), the smallest with two stages (33): the hand-scheduled kernel"""
    import torch
    rng = np.random.RandomState(11)
    seqs = clustered(rng, 3000)
    n = len(seqs)
    n_hash = {"odd_ld": 500, "even_padded_ld": 500, "n_hash_600": 600, "n_hash_2047": 2047, "n_hash_40": 40, "n_hash_33": 33}[case]
    out = None
    if case in ("odd_ld", "even_padded_ld"):
        buf = torch.full((n, n + (1 if case == "odd_ld" else 6)), -1.0, dtype=torch.float64, device="cuda")
        out = buf[:, :n]
    got, route = run(seqs, 4, n_hash, out=out, **SPLIT)
    assert route["split"], route
    assert same(got.cpu().numpy(), oracle(seqs, 4, n_hash))
    if case in ("odd_ld", "even_padded_ld"):
        assert bool((buf[:, n:] == -1.0).all())


def test_split_is_not_asked_for_where_it_cannot_pay(da):
    rng = np.random.RandomState(3)
    seqs = clustered(rng, 3000)
    _, route = run(seqs, 4, 500, DYNAALIGN_MH_NO_DEDUP=1, DYNAALIGN_MH_NO_SPARSE=1)          # below the built-in size bound
    assert not route["split"]
    _, route = run(seqs, 4, 32, **SPLIT)                                                      # a single stage: no 8-plane block
    assert not route["split"]
    few = low_complexity(rng, 3000, 3)                                                        # 81 4-mers: 8 planes are all the dictionaries need
    got, route = run(few, 4, 100, **SPLIT)
    assert not route["split"] and route["plane_bits"] == 8
    assert same(got.cpu().numpy(), oracle(few, 4, 100))


@pytest.mark.parametrize("form", ["rowspipe", "rows", "pipe", "tiles"])
def test_split_in_the_duplicate_route(da, form):
    """DYNAALIGN_MH_HYBRID_DEDUP=1: the unique table's compare on 8 planes (band kernel / one launch), the lists added to the uint16 table band by band
    before its rows are expanded -- all four expansion forms against the oracle"""
    rng = np.random.RandomState(17)
    pool = clustered(rng, 3500)
    seqs = [pool[q] for q in rng.randint(0, len(pool), 7000)]
    got, route = run(seqs, 4, 500, DYNAALIGN_MH_HYBRID_MIN_N=256, DYNAALIGN_MH_HYBRID_DEDUP=1, DYNAALIGN_MH_DEDUP_MIN_N=1, DYNAALIGN_MH_DEDUP_MAX_PCT=100,
                     DYNAALIGN_MH_EXPAND=form)
    assert route["dedup"] and route["split"] and route["plane_bits"] == 8, route
    assert same(got.cpu().numpy(), oracle(seqs, 4, 500))
