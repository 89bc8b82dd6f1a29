"""BASELINE config 5 -- clusterbreak(size_max=800, thresh_p=.8) with the GPU similarityMH backend,
cluster-membership parity (reference R/clusterbreak.R:180-275).

The same restated driver (dynaalign_amd/clusterbreak.py) runs twice:
  * device path : MinHashSession -- signatures resident in HBM; every recursion level is K1b + K2 + histogram +
                  exact type-7 quantile + edge extraction on the index subset; only edges leave the GPU;
  * oracle path : sim_fn = the CPU oracle's dense similarityMH matrix of the level's sequences, put through the
                  literal restatement of R's three statements (quantile type 7, S[S < thr] <- 0, upper triangle
                  incl. diagonal).
Same hash seed, same clustering function and seeds => the memberships ("<itr>.<cluster>" labels), the filtered
sequences and every level's threshold / edge count must be IDENTICAL.  (igraph's own Louvain cannot run here --
no R, no igraph -- so "parity vs reference" is parity of everything up to and including the graph handed to the
clustering function, plus identical output of one deterministic clustering function on both graphs.)"""
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


def oracle_sim_fn(k, n_hash, seed):
    seeds = O.seeds(seed, n_hash)

    def sim(x):
        rc, M = O.similarity_mh(x, k, n_hash, seeds)
        assert rc == 0
        return M
    return sim


def assert_same_result(a, b):
    assert a["clustered_seq"].tolist() == b["clustered_seq"].tolist()
    assert a["filtered_seq"] == b["filtered_seq"]
    assert (a.calls, a.convergence) == (b.calls, b.convergence)
    assert len(a.levels) == len(b.levels)
    for la, lb in zip(a.levels, b.levels):
        for key in ("itr", "n", "edges", "clusters", "oversize"):
            assert la[key] == lb[key], (key, la, lb)
        assert la["threshold"] == lb["threshold"] or (np.isnan(la["threshold"]) and np.isnan(lb["threshold"]))


@pytest.mark.parametrize("gen,n,size_max,min_calls", [("h3n2_like", 12000, 800, 3), ("h3n2_like", 3000, 100, 5),
                                                      ("uniform_peptides", 2500, 12, 5)])
def test_membership_parity_device_edge_path_vs_oracle_dense_path(da, gen, n, size_max, min_calls):
    """config 5's parameters (k=4, n_hash=500, thresh_p=.8, size_max=800) at an N whose dense matrix the oracle can build"""
    from dynaalign_amd import synth
    from dynaalign_amd.session import MinHashSession
    seqs = synth.to_strings(*getattr(synth, gen)(n, 20))
    k, n_hash, seed = 4, 500, 12345
    sess = MinHashSession(seqs, k, n_hash, seed=seed)
    dev = da.clusterbreak(seqs, thresh_p=0.8, size_max=size_max, size_min=3, session=sess, cluster_seed=42)
    ref = da.clusterbreak(seqs, thresh_p=0.8, size_max=size_max, size_min=3, sim_fn=oracle_sim_fn(k, n_hash, seed),
                          cluster_seed=42)
    assert_same_result(dev, ref)
    assert dev.calls >= min_calls                                   # the recursion really recursed
    assert sorted(dev["clustered_seq"][:, 0].tolist() + dev["filtered_seq"]) == sorted(seqs)
    sizes = np.unique(dev["clustered_seq"][:, 1], return_counts=True)[1]
    assert sizes.min() >= 3 and sizes.max() <= size_max


def test_sim_fn_contract_with_the_gpu_dense_matrix(da):
    """clusterbreak(pep, sim_fn = function(x) similarityMH(x, k = 4, n_hash = 500)) -- the reference's own call form
    (R/clusterbreak.R:174-179) -- through the dense host-pointer entry point, against the session path"""
    from dynaalign_amd import synth
    from dynaalign_amd.session import MinHashSession
    seqs = synth.to_strings(*synth.h3n2_like(2000, 20))
    dense = da.clusterbreak(seqs, thresh_p=0.8, size_max=60, size_min=3, cluster_seed=7,
                            sim_fn=lambda x: da.similarityMH(x, k=4, n_hash=500, seed=99))
    sess = da.clusterbreak(seqs, thresh_p=0.8, size_max=60, size_min=3, cluster_seed=7,
                           session=MinHashSession(seqs, 4, 500, seed=99))
    assert_same_result(dense, sess)
    assert dense.calls > 3


def test_nw_similarity_as_sim_fn(da):
    """sim_fn = similarityNW works through the same driver (dense contract); checked against the oracle's NW matrix"""
    from dynaalign_amd import synth
    seqs = synth.to_strings(*synth.h3n2_like(600, 20))

    def sim_oracle(x):
        rc, M, _ = O.similarity_nw(x)
        assert rc == 0
        return M
    a = da.clusterbreak(seqs, thresh_p=0.8, size_max=30, size_min=3, sim_fn=lambda x: da.similarityNW(x), cluster_seed=3)
    b = da.clusterbreak(seqs, thresh_p=0.8, size_max=30, size_min=3, sim_fn=sim_oracle, cluster_seed=3)
    assert_same_result(a, b)
    assert a.calls > 3


def test_device_csr_equals_the_host_built_graph(da):
    """da_dev_edges_to_csr (radix sort on the device) against a numpy construction from the same edge list, and the whole
    recursion with the CSR path against the edge-list path (DYNAALIGN_CLUSTERBREAK_NO_CSR=1): identical labels"""
    import os
    import torch
    from dynaalign_amd import device, synth, _capi
    from dynaalign_amd.session import MinHashSession
    n = 3000
    seqs = synth.to_strings(*synth.h3n2_like(n, 20))
    sess = MinHashSession(seqs, 4, 200, seed=12345)
    thr, ei, ej, ew = sess.edges(None, 0.8, sort=False)
    thr2, n_edges, ptr, adj, codes, loops, values = sess.edges_csr(None, 0.8)
    assert thr2 == thr and n_edges == len(ei)
    cd = np.rint(ew * 200).astype(np.uint16)
    assert np.array_equal(values[cd], ew)
    off = ei != ej
    r = np.r_[ei[off], ej[off]].astype(np.int64)
    c = np.r_[ej[off], ei[off]].astype(np.int64)
    v = np.r_[cd[off], cd[off]]
    order = np.lexsort((c, r))
    want_ptr = np.zeros(n + 1, np.int64)
    np.add.at(want_ptr, r + 1, 1)
    assert np.array_equal(ptr, np.cumsum(want_ptr)) and np.array_equal(adj, c[order].astype(np.int32)) and np.array_equal(codes, v[order])
    want_loops = np.full(n, 0xFFFF, np.uint16)
    want_loops[ei[~off]] = cd[~off]
    assert np.array_equal(loops, want_loops)
    a = da.clusterbreak(seqs, thresh_p=0.8, size_max=200, size_min=3, session=sess, cluster_seed=1)
    os.environ["DYNAALIGN_CLUSTERBREAK_NO_CSR"] = "1"
    try:
        b = da.clusterbreak(seqs, thresh_p=0.8, size_max=200, size_min=3, session=sess, cluster_seed=1)
    finally:
        del os.environ["DYNAALIGN_CLUSTERBREAK_NO_CSR"]
    assert a.calls == b.calls and a.calls > 1
    assert np.array_equal(a["clustered_seq"], b["clustered_seq"]) and a["filtered_seq"] == b["filtered_seq"]


def _dense_level_on_device(D, n, n_hash, thresh_p):
    """R/clusterbreak.R:219-221 applied ON THE DEVICE to a dense float64 similarity matrix D (n x n torch tensor): the type-7 quantile of
    S[upper.tri(S)] by R's own arithmetic (index = 1 + (M - 1) p, lo / hi order statistics from the exact value histogram -- the
    similarities take n_hash + 1 values -- qs = (1 - h) x[lo] + h x[hi]), S[S < thr] <- 0, and the graph of the upper triangle incl. the
    diagonal as a symmetric CSR (both directions of every off-diagonal entry, row-major) + the self-loops.  torch is the checker here,
    nothing of the library runs in this function.  Returns (thr, n_edges, ptr, adj, codes, loops) -- the last four as device tensors."""
    import torch
    dev = D.device
    hist = torch.zeros(n_hash + 1, dtype=torch.int64, device=dev)
    nh = torch.tensor([[float(n_hash)]], dtype=torch.float64, device=dev)
    step = max(1, min(n, (1 << 28) // max(n, 1)))
    cols = torch.arange(n, device=dev)
    for r0 in range(0, n, step):
        r1 = min(n, r0 + step)
        blk = D[r0:r1, :n]
        cnt = torch.round(blk * n_hash).to(torch.int64)
        # count / n_hash IS the matrix (src/minHash.cpp:174).  (An element-wise tensor / tensor division: torch turns a division by a
        # Python scalar into a multiplication by its reciprocal, which is not the reference's divide.)
        assert torch.equal(torch.div(cnt.to(torch.float64), nh.expand_as(blk)), blk)
        upper = cols[None, :] > torch.arange(r0, r1, device=dev)[:, None]            # upper.tri(): strictly above the diagonal
        hist += torch.bincount(cnt[upper], minlength=n_hash + 1)
    h = hist.cpu().numpy()
    M = int(h.sum())
    assert M == n * (n - 1) // 2
    cum = np.cumsum(h)
    values = np.arange(n_hash + 1, dtype=np.float64) / n_hash
    index = 1.0 + (M - 1) * float(thresh_p)
    lo, hi = int(np.floor(index)), int(np.ceil(index))
    x_lo, x_hi = values[np.searchsorted(cum, lo)], values[np.searchsorted(cum, hi)]   # the lo-th / hi-th smallest (1-based)
    thr = float(x_lo)
    if index > lo and x_hi != x_lo:
        thr = (1.0 - (index - lo)) * float(x_lo) + (index - lo) * float(x_hi)
    ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    adj_parts, code_parts = [], []
    loops = torch.full((n,), -1, dtype=torch.int16, device=dev)                      # 0xFFFF = no self-loop
    n_edges = 0
    for r0 in range(0, n, step):
        r1 = min(n, r0 + step)
        blk = D[r0:r1, :n]
        keep = (~(blk < thr)) & (blk != 0)                                           # S[S < thr] <- 0, then the non-zero entries
        rows = torch.arange(r0, r1, device=dev)
        diag = keep[rows - r0, rows]
        cnt = torch.round(blk * n_hash).to(torch.int16)
        loops[r0:r1] = torch.where(diag, cnt[rows - r0, rows], loops[r0:r1])
        n_edges += int(diag.sum().item()) + int((keep & (cols[None, :] > rows[:, None])).sum().item())
        keep[rows - r0, rows] = False
        ptr[r0 + 1:r1 + 1] = keep.sum(dim=1)
        rr, cc = torch.nonzero(keep, as_tuple=True)                                  # row-major = CSR order
        adj_parts.append(cc.to(torch.int32))
        code_parts.append(cnt[rr, cc])
    return thr, n_edges, torch.cumsum(ptr, 0), torch.cat(adj_parts), torch.cat(code_parts), loops


def test_config5_first_levels_at_100k_device_edge_path_vs_dense_matrix(da):
    """BASELINE config 5 at its FULL size, graph level: clusterbreak(size_max = 800, thresh_p = .8) on the 100 000
    h3n2-like peptides.  The device edge path's first level -- threshold, edge count, the CSR handed to the clustering function
    (MinHashSession.edges_csr) -- against R's three statements (R/clusterbreak.R:219-221) applied to the dense float64 matrix of
    da_dev_similarity_mh (itself compared with the oracle by whole rows in test_gpu_fullsize.py), and the same for one deeper level: the
    subset of the largest first-level cluster (what :246-254 recurses on)."""
    import torch
    from dynaalign_amd import device, synth
    from dynaalign_amd.clusterbreak import louvain_csr
    from dynaalign_amd.session import MinHashSession
    n, k, n_hash, seed = 100000, 4, 500, 12345
    res, off = synth.h3n2_like(n, 20)
    seqs = synth.to_strings(res, off)
    sess = MinHashSession(seqs, k, n_hash, seed=seed)
    thr, n_edges, ptr, adj, codes, loops, values = sess.edges_csr(None, 0.8)
    assert np.array_equal(values, np.arange(n_hash + 1) / n_hash)
    member = louvain_csr(n, ptr, adj, codes, loops, values, seed=1)                   # (only to pick the subset of the deeper level)
    big = np.flatnonzero(member == np.bincount(member).argmax())
    assert 800 < len(big) < n                                                        # oversize: clusterbreak would recurse on it
    sub = sess.edges_csr(big, 0.8)
    torch.cuda.empty_cache()
    ds = device.DeviceSequences(res, off)
    D = device.similarity_mh(ds, k, n_hash, da.hash_family_seeds(seed, n_hash))       # dense float64, 80 GB, stays on the device
    torch.cuda.synchronize()

    def compare(got, Dm, m):
        g_thr, g_edges, g_ptr, g_adj, g_codes, g_loops = got[:6]
        w_thr, w_edges, w_ptr, w_adj, w_codes, w_loops = _dense_level_on_device(Dm, m, n_hash, 0.8)
        assert g_thr == w_thr and g_edges == w_edges
        assert np.array_equal(g_ptr, w_ptr.cpu().numpy())
        assert torch.equal(torch.from_numpy(g_adj).to(w_adj.device), w_adj)
        assert torch.equal(torch.from_numpy(g_codes.view(np.int16)).to(w_codes.device), w_codes)
        assert np.array_equal(g_loops.view(np.int16), w_loops.cpu().numpy())
        return w_thr, w_edges
    t1, e1 = compare((thr, n_edges, ptr, adj, codes, loops), D, n)
    assert e1 > n and t1 >= 0.0
    idx = torch.from_numpy(big).to(D.device)
    Dsub = D.index_select(0, idx).index_select(1, idx).contiguous()                   # similarityMH(sequences[idx]) under the same hash family
    del D
    compare(sub, Dsub, len(big))
