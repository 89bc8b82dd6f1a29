"""the threshold + sparsify step that follows the hot path in clusterbreak
(reference R/clusterbreak.R:219-221, netcluster :122-124), fused on the device.
Checker: the dense oracle matrix + a line-by-line restatement of R's quantile type 7."""
import math

import numpy as np
import pytest

import oracle_lib as O


def r_quantile_type7(x, p):
    """stats::quantile.default(x, p, type = 7), restated"""
    x = np.sort(np.asarray(x, np.float64))
    n = len(x)
    index = 1 + max(n - 1, 0) * p
    lo, hi = math.floor(index), math.ceil(index)
    qs = x[lo - 1]
    if index > lo and x[hi - 1] != qs:
        h = index - lo
        qs = (1 - h) * qs + h * x[hi - 1]
    return qs


@pytest.fixture(scope="module")
def da(built):
    import dynaalign_amd
    return dynaalign_amd


def test_quantile_type7_from_histogram(da):
    rng = np.random.RandomState(1)
    for it in range(300):
        nb = int(rng.randint(1, 40))
        hist = rng.randint(0, 6, nb).astype(np.uint64)
        if it % 5 == 0:
            hist[rng.randint(0, nb)] += int(rng.randint(1, 10 ** 6))
        if hist.sum() == 0:
            hist[0] = 1
        values = np.arange(nb) / max(nb - 1, 1)
        x = np.repeat(values, hist.astype(np.int64))
        for p in (0.0, 0.5, 0.8, 0.95, 1.0, float(rng.rand())):
            got = da.quantile_type7(hist, values, p)
            assert got == r_quantile_type7(x, p), (hist, p)
            assert abs(got - np.quantile(x, p)) <= 1e-12      # numpy's "linear" method is the same definition
    with pytest.raises(da.DynaAlignError):
        da.quantile_type7([0, 0], [0.0, 1.0], 0.5)
    with pytest.raises(da.DynaAlignError):
        da.quantile_type7([1], [0.0], 1.5)


def reference_edges(M, p):
    """what clusterbreak/netcluster keep: threshold, then non-zero upper-triangle entries incl. diagonal"""
    n = M.shape[0]
    thr = r_quantile_type7(M[np.triu_indices(n, 1)], p)
    S = M.copy()
    S[S < thr] = 0
    iu = np.triu_indices(n)
    keep = S[iu] != 0
    return thr, iu[0][keep], iu[1][keep], S[iu][keep]


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,n_hash,p", [(2, 4, 50, 0.8), (130, 4, 500, 0.8), (700, 4, 500, 0.8), (700, 4, 500, 0.99),
                                          (641, 2, 50, 0.8), (300, 3, 33, 0.5), (257, 4, 64, 0.0), (257, 4, 64, 1.0)])
def test_mh_edges_match_dense_threshold(da, evp, n, k, n_hash, p):
    from dynaalign_amd import synth
    seqs = evp if n == 641 else synth.to_strings(*synth.h3n2_like(n, 20))
    seeds = da.hash_family_seeds(12345, n_hash)
    rc, M = O.similarity_mh(seqs, k, n_hash, seeds)
    assert rc == 0
    thr_w, iw, jw, ww = reference_edges(M, p)
    thr, i, j, w = da.similarityMH_edges(seqs, k, n_hash, p, seed=12345)
    assert thr == thr_w
    assert np.array_equal(i, iw) and np.array_equal(j, jw)
    assert np.array_equal(w.view(np.uint64), ww.view(np.uint64))
    assert np.all(i <= j) and len(i) >= len(seqs)              # the diagonal (1.0) always survives


@pytest.mark.gpu
def test_mh_edges_10k_counts(da):
    """config-2 size: the edge list reproduces the dense threshold step without the dense matrix on the host"""
    from dynaalign_amd import synth
    seqs = synth.to_strings(*synth.h3n2_like(10000, 20))
    M = np.asarray(da.similarityMH(seqs, 4, 500, seed=12345))
    thr_w, iw, jw, ww = reference_edges(M, 0.8)
    thr, i, j, w = da.similarityMH_edges(seqs, 4, 500, 0.8, seed=12345)
    assert thr == thr_w and np.array_equal(i, iw) and np.array_equal(j, jw) and np.array_equal(w, ww)


@pytest.mark.gpu
def test_mh_edges_argument_errors(da):
    with pytest.raises(da.DynaAlignError):
        da.similarityMH_edges(["ACDEF"], 4, 50, 0.8, seed=1)          # no pairs: quantile of an empty set
    with pytest.raises(da.DynaAlignError):
        da.similarityMH_edges(["ACDEF", "ACDEG"], 4, 50, 1.5, seed=1)
    with pytest.raises(da.DynaAlignError, match="cannot be empty"):
        da.similarityMH_edges([], 4, 50, 0.8, seed=1)


@pytest.mark.gpu
@pytest.mark.parametrize("n,p,args", [(2, 0.8, ("BLOSUM62", 10, 4)), (130, 0.8, ("BLOSUM62", 10, 4)),
                                      (641, 0.8, ("BLOSUM62", 10, 4)), (500, 0.5, ("BLOSUM45", 5, 1)),
                                      (257, 0.0, ("BLOSUM80", 12, 2)), (257, 1.0, ("BLOSUM62", 10, 4)),
                                      (400, 0.97, ("BLOSUM62", 0, 0))])
def test_nw_edges_match_dense_threshold(da, evp, n, p, args):
    """NW ratios take few distinct values too: histogram of (matches, length) codes -> exact type-7 quantile.
    Different codes with equal ratio (1/2, 2/4) must behave as one value."""
    from dynaalign_amd import synth
    seqs = evp if n == 641 else synth.to_strings(*synth.h3n2_like(n, 20))
    rc, M, _ = O.similarity_nw(seqs, *args)
    assert rc == 0
    thr_w, iw, jw, ww = reference_edges(M, p)
    thr, i, j, w = da.similarityNW_edges(seqs, *args, thresh_p=p)
    assert thr == thr_w
    assert np.array_equal(i, iw) and np.array_equal(j, jw)
    assert np.array_equal(w.view(np.uint64), ww.view(np.uint64))


@pytest.mark.gpu
def test_nw_edges_ragged_lengths_and_errors(da):
    rng = np.random.RandomState(11)
    alpha = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", np.uint8)
    seqs = ["".join(map(chr, alpha[rng.randint(0, 20, rng.randint(1, 60))])) for _ in range(220)]
    rc, M, _ = O.similarity_nw(seqs, "BLOSUM62", 10, 4)
    thr_w, iw, jw, ww = reference_edges(M, 0.9)
    thr, i, j, w = da.similarityNW_edges(seqs, thresh_p=0.9)
    assert thr == thr_w and np.array_equal(i, iw) and np.array_equal(j, jw) and np.array_equal(w, ww)
    with pytest.raises(da.DynaAlignError, match="empty"):
        da.similarityNW_edges(["ACD", "", "ACE"])
    with pytest.raises(da.DynaAlignError, match="Invalid substitution matrix name"):
        da.similarityNW_edges(["ACD", "ACE"], "PAM250")
    with pytest.raises(da.DynaAlignError, match="Invalid amino acid"):
        da.similarityNW_edges(["ACD", "AJE"])
    with pytest.raises(da.DynaAlignError):
        da.similarityNW_edges(["ACD"])


@pytest.mark.gpu
def test_session_subsets_equal_fresh_calls(da):
    """signatures stay in HBM; a recursion level on an index subset gives exactly what a
    fresh similarityMH / similarityMH_edges call on those sequences gives under the same seed"""
    from dynaalign_amd import synth
    from dynaalign_amd.session import MinHashSession
    seqs = synth.to_strings(*synth.h3n2_like(1500, 20))
    s = MinHashSession(seqs, 4, 200, seed=777)
    rng = np.random.RandomState(0)
    full = np.asarray(da.similarityMH(seqs, 4, 200, seed=777))
    assert np.array_equal(np.asarray(s.similarity()).view(np.uint64), full.view(np.uint64))
    for m in (2, 129, 640):
        idx = rng.permutation(1500)[:m]
        sub = [seqs[i] for i in idx]
        want = np.asarray(da.similarityMH(sub, 4, 200, seed=777))
        assert np.array_equal(np.asarray(s.similarity(idx)).view(np.uint64), want.view(np.uint64))
        assert np.array_equal(want, full[np.ix_(idx, idx)])                  # and it is the sub-block of the full matrix
        thr_w, iw, jw, ww = da.similarityMH_edges(sub, 4, 200, 0.8, seed=777)
        thr, i, j, w = s.edges(idx, 0.8)
        assert thr == thr_w and np.array_equal(i, iw) and np.array_equal(j, jw) and np.array_equal(w, ww)
    with pytest.raises(da.DynaAlignError):
        s.similarity([])
