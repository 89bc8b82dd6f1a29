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
