"""GPU tests of the duplicate-collapsing route of similarityNW (nw_kernels.hip "duplicate sequences", api.cpp
nw_full_symmetric): byte-identical sequences are collapsed, the DP runs on the table of unique strings as an ORDERED square
(calculate_similarity is not symmetric and the reference evaluates calc(seq[i], seq[j]) for i < j, reference
src/pairwiseSeqAlign.cpp:340-346) and the n x n result is an index expansion.  Must be bit-identical to the
direct kernel and to the oracle -- in particular where both orders of one pair of strings occur in the input."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

ASYM_A, ASYM_B = "YDYIHIYADKQDRIGWLGNT", "MYCEMNVEIQYMATKNMWNT"      # calc(A, B) = 3/21, calc(B, A) = 4/21


@pytest.fixture(scope="module")
def da(built):
    import dynaalign_amd
    from dynaalign_amd import _capi
    assert _capi.load().da_device_count() > 0
    return dynaalign_amd


@pytest.fixture()
def small_n_route(monkeypatch):
    monkeypatch.setenv("DYNAALIGN_NW_DEDUP_MIN_N", "1")


def oracle_matrix(seqs, matrix="BLOSUM62", go=10, ge=4):
    rc, mt, ln, _, msg = O.nw_rows(seqs, 0, len(seqs), matrix, go, ge)
    assert rc == 0, msg
    with np.errstate(invalid="ignore", divide="ignore"):
        up = mt.astype(np.float64) / ln.astype(np.float64)
    return up


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def both_routes(da, seqs, *args):
    got = np.asarray(da.similarityNW(seqs, *args))
    os.environ["DYNAALIGN_NW_NO_DEDUP"] = "1"
    try:
        direct = np.asarray(da.similarityNW(seqs, *args))
    finally:
        del os.environ["DYNAALIGN_NW_NO_DEDUP"]
    return got, direct


def test_both_orders_of_an_asymmetric_pair(da, small_n_route):
    """[A, B, A, B, A]: (0,1) needs calc(A,B), (1,2) needs calc(B,A) -- the unique table must hold both"""
    rc, mt, ln, _, _ = O.nw_pair(ASYM_A, ASYM_B)
    assert (mt, ln) == (3, 21)
    rc, mt, ln, _, _ = O.nw_pair(ASYM_B, ASYM_A)
    assert (mt, ln) == (4, 21)
    for seqs in ([ASYM_A, ASYM_B, ASYM_A, ASYM_B, ASYM_A], [ASYM_B, ASYM_B, ASYM_A, ASYM_A], [ASYM_A, ASYM_A, ASYM_B, ASYM_B],
                 [ASYM_A, ASYM_B], [ASYM_B, ASYM_A, ASYM_B]):
        got, direct = both_routes(da, seqs)
        want = oracle_matrix(seqs)
        assert same(got, want) and same(direct, want)
    got, _ = both_routes(da, [ASYM_A, ASYM_B, ASYM_A])
    assert got[0, 1] == 3 / 21 and got[1, 2] == 4 / 21 and got[0, 2] == 1.0


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_duplicates_ragged_lengths(da, small_n_route, seed):
    """few distinct strings drawn many times in random order (every pair of strings occurs in both orders), lengths 0..40,
    all 24 residue symbols, several matrices / penalties"""
    rng = np.random.RandomState(seed)
    alpha = np.frombuffer(b"ARNDCQEGHILKMFPSTWYVBZX*", np.uint8)
    pool = ["".join(map(chr, alpha[rng.randint(0, 24, rng.randint(0, 41))])) for _ in range(60)] + ["", ASYM_A, ASYM_B]
    seqs = [pool[k] for k in rng.randint(0, len(pool), 700)] + ["".join(map(chr, alpha[rng.randint(0, 24, 20)])) for _ in range(150)]
    rng.shuffle(seqs)
    args = [("BLOSUM62", 10, 4), ("BLOSUM45", 3, 1), ("BLOSUM100", 0, 0)][seed - 1]
    got, direct = both_routes(da, seqs, *args)
    want = oracle_matrix(seqs, *args)
    assert same(direct, want)
    assert same(got, want)


def test_headline_like_set_takes_the_route_and_matches(da):
    """h3n2-like windows: ~half of the sequences are exact duplicates -> the route is taken by itself (n >= 2048)"""
    from dynaalign_amd import synth
    n = 4000
    seqs = synth.to_strings(*synth.h3n2_like(n, 20))
    assert len(set(seqs)) < 0.85 * n
    got, direct = both_routes(da, seqs)
    assert same(got, direct)
    rows = [0, 1, 63, 64, 1999, 3999]
    for r in rows:
        rc, mt, ln, _, _ = O.nw_rows(seqs, r, r + 1)
        with np.errstate(invalid="ignore", divide="ignore"):
            want = mt[0].astype(np.float64) / ln[0].astype(np.float64)
        # the oracle's row r holds calc(seq[min], seq[max]) for every column, like the matrix row
        assert same(got[r], want)


def test_all_identical_and_all_distinct(da, small_n_route):
    from dynaalign_amd import synth
    got, direct = both_routes(da, ["MKTIIALSYIFCLVFA"] * 300)
    assert same(got, direct) and np.all(got == 1.0)
    seqs = synth.to_strings(*synth.uniform_peptides(500, 20))          # nothing to collapse: the plan hands over to the direct kernel
    got, direct = both_routes(da, seqs)
    assert same(got, direct)


def test_edge_list_and_device_entry_use_the_same_route(da):
    from dynaalign_amd import synth, device, _capi
    import torch
    seqs = synth.to_strings(*synth.h3n2_like(3000, 20))
    thr, ei, ej, ew = da.similarityNW_edges(seqs, "BLOSUM62", 10, 4, 0.8)
    os.environ["DYNAALIGN_NW_NO_DEDUP"] = "1"
    try:
        thr2, ei2, ej2, ew2 = da.similarityNW_edges(seqs, "BLOSUM62", 10, 4, 0.8)
    finally:
        del os.environ["DYNAALIGN_NW_NO_DEDUP"]
    assert thr == thr2 and np.array_equal(ei, ei2) and np.array_equal(ej, ej2) and np.array_equal(ew, ew2)
    res, off = synth.h3n2_like(3000, 20)
    ds = device.DeviceSequences(res, off)
    assert int(device.nw_encode(ds).item()) == 0
    a = device.nw(ds, kind=_capi.DA_OUT_COMPACT)
    b = device.nw(ds)
    os.environ["DYNAALIGN_NW_NO_DEDUP"] = "1"
    try:
        a2 = device.nw(ds, kind=_capi.DA_OUT_COMPACT)
        b2 = device.nw(ds)
    finally:
        del os.environ["DYNAALIGN_NW_NO_DEDUP"]
    assert torch.equal(a, a2) and torch.equal(b.view(torch.int64), b2.view(torch.int64))


@pytest.mark.parametrize("seed,n_pool,n,alpha", [(1, 120, 400, "ACDEFGHIKLMNPQRSTVWY"), (2, 300, 900, "ARN"), (3, 60, 700, "ACDEFGHIKLMNPQRSTVWYBZX*")])
def test_prefix_sharing_of_the_ordered_dp(da, small_n_route, monkeypatch, seed, n_pool, n, alpha):
    """round 4: the ordered DP processes the unique strings in lexicographic order and resumes a row's DP at the depth it shares with the
    previous needed row (k_nw_short<.., PFX>).  Inputs built to stress it: families of strings that share prefixes of every length, strings
    that are proper prefixes of others (incl. the empty string), ragged lengths 0..20, a three-letter alphabet (long common prefixes AND many
    gaps), every string several times at scattered positions (both orders of most pairs needed) -- against the oracle's DP + traceback, the
    direct kernel, and the same route with the sharing switched off (DYNAALIGN_NW_NO_PREFIX_SHARE=1), bit for bit."""
    rng = np.random.RandomState(seed)
    pool = set()
    while len(pool) < n_pool:
        base = "".join(rng.choice(list(alpha), rng.randint(1, 21)))
        pool.add(base)
        for _ in range(rng.randint(0, 4)):                          # relatives: a prefix of `base` + a different tail
            cut = rng.randint(0, len(base) + 1)
            tail = "".join(rng.choice(list(alpha), rng.randint(0, 21 - cut)))
            pool.add((base[:cut] + tail)[:20])
        pool.add(base[:rng.randint(0, len(base) + 1)])              # a proper prefix (possibly empty)
    pool = sorted(pool)
    seqs = [pool[k] for k in rng.randint(0, len(pool), n)]
    assert "" in seqs or True
    got, direct = both_routes(da, seqs, "BLOSUM62", 10, 4)
    want = oracle_matrix(seqs)
    assert same(got, want) and same(direct, want)
    monkeypatch.setenv("DYNAALIGN_NW_NO_PREFIX_SHARE", "1")
    plain = np.asarray(da.similarityNW(seqs, "BLOSUM62", 10, 4))
    monkeypatch.delenv("DYNAALIGN_NW_NO_PREFIX_SHARE")
    assert same(plain, want)
    got2, _ = both_routes(da, seqs, "BLOSUM45", 3, 1)               # other penalties: more gaps, other tie-breaks
    assert same(got2, oracle_matrix(seqs, "BLOSUM45", 3, 1))
    if seed == 2:
        # the checkpoint keeps Ix' as byte deltas min(score(VM) - score(Ix'), gapOpen): no gap-open cost at all (every delta clamps to 0), the largest
        # gapOpen the bytes hold, one past it (that call takes the plain ordered kernel), extension-only and open-only penalties
        for go, ge in ((0, 0), (1, 0), (0, 3), (255, 2), (256, 2), (40, 20)):
            got3 = np.asarray(da.similarityNW(seqs, "BLOSUM62", go, ge))
            assert same(got3, oracle_matrix(seqs, "BLOSUM62", go, ge)), (go, ge)
