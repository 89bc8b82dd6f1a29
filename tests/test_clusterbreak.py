"""CPU tests of the caller-side driver: the restated clusterbreak recursion
(reference R/clusterbreak.R:180-275), netcluster (:112-136) and the host-side Louvain stand-in for
igraph::cluster_louvain.  The similarity matrices come from the CPU oracle here (sim_fn contract); the GPU
twin -- device edge path vs this dense path, identical memberships -- is tests/test_gpu_clusterbreak.py."""
import numpy as np
import pytest

import oracle_lib as O
from test_threshold_edges import reference_edges


@pytest.fixture(scope="module")
def da(built):
    import dynaalign_amd
    return dynaalign_amd


def planted(n, groups, p_in, p_out, seed, loops=True):
    rng = np.random.RandomState(seed)
    ei, ej, ew = [], [], []
    for a in range(n):
        if loops:
            ei.append(a); ej.append(a); ew.append(1.0)
        for b in range(a + 1, n):
            if rng.rand() < (p_in if a * groups // n == b * groups // n else p_out):
                ei.append(a); ej.append(b); ew.append(float(rng.randint(1, 500)) / 500)
    return np.array(ei, np.int32), np.array(ej, np.int32), np.array(ew)


def nx_graph(n, ei, ej, ew):
    import networkx as nx
    G = nx.Graph()
    G.add_nodes_from(range(n))
    for a, b, w in zip(ei.tolist(), ej.tolist(), ew.tolist()):
        G.add_edge(a, b, weight=w)
    return G


# ------------------------------------------------------------------ Louvain

def test_louvain_finds_planted_groups_and_numbers_by_first_appearance(da):
    n = 90
    ei, ej, ew = planted(n, 3, 0.7, 0.01, 0)
    m, q = da.louvain(n, ei, ej, ew, return_modularity=True)
    assert m.tolist() == [1] * 30 + [2] * 30 + [3] * 30          # ids start at 1 (R), in order of first appearance
    assert 0.5 < q < 1.0


def test_louvain_is_independent_of_edge_order_and_orientation(da):
    n = 400
    ei, ej, ew = planted(n, 8, 0.3, 0.02, 1)
    base = da.louvain(n, ei, ej, ew, seed=7)
    rng = np.random.RandomState(2)
    for _ in range(3):
        perm = rng.permutation(len(ei))
        flip = rng.rand(len(ei)) < 0.5
        a = np.where(flip, ej, ei)[perm]
        b = np.where(flip, ei, ej)[perm]
        assert np.array_equal(da.louvain(n, a, b, ew[perm], seed=7), base)
    # duplicate entries are summed: splitting every weight in two halves is the same graph
    assert np.array_equal(da.louvain(n, np.r_[ei, ei], np.r_[ej, ej], np.r_[ew / 2, ew / 2], seed=7), base)


def test_louvain_modularity_matches_networkx_definition_and_quality(da):
    import networkx as nx
    n = 300
    ei, ej, ew = planted(n, 6, 0.25, 0.03, 3)
    G = nx_graph(n, ei, ej, ew)
    for res in (1.0, 1.05, 0.7):
        m, q = da.louvain(n, ei, ej, ew, resolution=res, seed=5, return_modularity=True)
        comms = [set(np.nonzero(m == c)[0].tolist()) for c in np.unique(m)]
        # same modularity convention (self-loops count twice in the strength, once as internal weight)
        assert abs(nx.community.modularity(G, comms, resolution=res) - q) < 1e-12
        best_nx = max(nx.community.modularity(G, nx.community.louvain_communities(G, resolution=res, seed=s), resolution=res)
                      for s in range(3))
        assert q >= 0.97 * best_nx


def test_louvain_edge_cases(da):
    assert da.louvain(0, [], [], []).tolist() == []
    assert da.louvain(4, [], [], []).tolist() == [1, 2, 3, 4]                 # no edges: everyone alone
    assert da.louvain(3, [0, 1, 2], [0, 1, 2], [1.0, 1.0, 1.0]).tolist() == [1, 2, 3]   # loops only
    assert da.louvain(4, [2, 0], [3, 1], [1.0, 1.0]).tolist() == [1, 1, 2, 2]
    # weights = FALSE (an unweighted graph, igraph_weight = FALSE): the heavy bridge no longer decides
    ei, ej = np.array([0, 0, 1, 3, 3, 4, 2], np.int32), np.array([1, 2, 2, 4, 5, 5, 3], np.int32)
    ew = np.array([1, 1, 1, 1, 1, 1, 50.0])
    assert da.louvain(6, ei, ej, ew, weights=False).tolist() == [1, 1, 1, 2, 2, 2]
    assert da.louvain(6, ei, ej, ew).tolist() != [1, 1, 1, 2, 2, 2]
    with pytest.raises(da.DynaAlignError):
        da.louvain(3, [0], [5], [1.0])
    with pytest.raises(da.DynaAlignError):
        da.louvain(3, [0], [1], [float("nan")])


# ------------------------------------------------------------------ threshold step, netcluster

def test_dense_threshold_step_is_the_r_statement(da):
    from dynaalign_amd.clusterbreak import threshold_edges_dense
    from dynaalign_amd import synth
    seqs = synth.to_strings(*synth.h3n2_like(500, 20))
    rc, M = O.similarity_mh(seqs, 4, 100, O.seeds(3, 100))
    for p in (0.0, 0.5, 0.8, 0.97, 1.0):
        thr, i, j, w = threshold_edges_dense(M, p)
        thr_w, iw, jw, ww = reference_edges(M, p)
        assert thr == thr_w and np.array_equal(i, iw) and np.array_equal(j, jw) and np.array_equal(w, ww)
    with pytest.raises(ValueError, match="NaN"):
        threshold_edges_dense(np.array([[1.0, np.nan], [np.nan, 1.0]]), 0.8)


def test_netcluster_example_of_the_reference(da):
    """R/clusterbreak.R:89-98: two blocks of ones -> two clusters"""
    A = np.array([[1, 1, 0, 0], [1, 1, 0, 0], [0, 0, 1, 1], [0, 0, 1, 1]], float)
    assert da.netcluster(A).tolist() == [1, 1, 2, 2]
    with pytest.raises(ValueError, match="square"):
        da.netcluster(np.zeros((2, 3)))
    with pytest.raises(ValueError, match="Wrong clustering output format"):
        da.netcluster(A, cluster_func=lambda n, i, j, w, seed, weights: "x")


# ------------------------------------------------------------------ the recursion

def test_argument_checks(da):
    with pytest.raises(ValueError, match="size_max must be greater than size_min"):
        da.clusterbreak(["AAAA"], size_max=3, size_min=3)
    with pytest.raises(ValueError, match="empty input sequence vector"):
        da.clusterbreak([])


def scripted(script):
    """cluster_fn that returns pre-written memberships call by call and records what it was given"""
    calls = []

    def fn(n, i, j, w, seed, weights):
        calls.append((n, seed))
        out = script[len(calls) - 1]
        assert len(out) == n
        return np.array(out)
    fn.calls = calls
    return fn


def test_bookkeeping_follows_the_r_code(da):
    """Hand-derived from R/clusterbreak.R:217-254 with size_max = 3, size_min = 2:
    call 1 on p0..p9, memberships 2 2 2 2 1 1 3 2 2 4:
        sizes c(2, 6, 1, 1); id.itr = 2, id.rm = 3, 4 -> filtered p6, p9; kept now: p4, p5 as "1.1";
        cluster 2 (p0 p1 p2 p3 p7 p8) recurses as call 2
    call 2, memberships 1 1 2 2 2 2: sizes c(2, 4); id.itr = 2; kept "2.1": p0 p1; cluster 2 (p2 p3 p7 p8) -> call 3
    call 3, memberships 1 1 1 2: sizes c(3, 1); nothing oversize; id.rm = 2 -> filtered p8; kept "3.1": p2 p3 p7"""
    pep = ["p%d" % t for t in range(10)]
    fn = scripted([[2, 2, 2, 2, 1, 1, 3, 2, 2, 4], [1, 1, 2, 2, 2, 2], [1, 1, 1, 2]])
    sim = lambda x: np.ones((len(x), len(x)))
    r = da.clusterbreak(pep, size_max=3, size_min=2, sim_fn=sim, cluster_fn=fn, cluster_seed=100)
    assert r["clustered_seq"].tolist() == [["p4", "1.1"], ["p5", "1.1"], ["p0", "2.1"], ["p1", "2.1"],
                                            ["p2", "3.1"], ["p3", "3.1"], ["p7", "3.1"]]
    assert r["filtered_seq"] == ["p6", "p9", "p8"]
    assert (r.calls, r.convergence) == (3, 1)
    assert fn.calls == [(10, 101), (6, 102), (4, 103)]


def test_oversize_clusters_recurse_in_order_of_first_appearance(da):
    """unique(pep.new[,2]) (R/clusterbreak.R:247): cluster 3 shows up before cluster 1 -> it is call 2"""
    pep = ["p%d" % t for t in range(8)]
    fn = scripted([[3, 3, 3, 1, 1, 1, 2, 2], [1, 1, 2], [1, 2, 2]])
    r = da.clusterbreak(pep, size_max=2, size_min=1, sim_fn=lambda x: np.ones((len(x), len(x))), cluster_fn=fn)
    assert [n for n, _ in fn.calls] == [8, 3, 3]
    assert r["clustered_seq"].tolist() == [["p6", "1.2"], ["p7", "1.2"], ["p0", "2.1"], ["p1", "2.1"], ["p2", "2.2"],
                                            ["p3", "3.1"], ["p4", "3.2"], ["p5", "3.2"]]
    assert r["filtered_seq"] == [] and r.calls == 3


def test_single_surviving_row_is_the_r_error(da):
    """pep.ref[mask, ] with ONE row drops to a vector; nrow() is NULL; `if (NULL > 0)` stops with this message"""
    fn = scripted([[1, 2, 2, 2, 2]])      # cluster 2 is oversize, cluster 1 (one member) is the only row of pep.out
    with pytest.raises(RuntimeError, match="argument is of length zero"):
        da.clusterbreak(list("abcde"), size_max=3, size_min=0, sim_fn=lambda x: np.ones((5, 5)), cluster_fn=fn)


def test_max_itr_stops_the_recursion(da):
    pep = ["p%d" % t for t in range(6)]
    fn = scripted([[1, 1, 1, 2, 2, 2], [1, 1, 1], [1, 1, 1]])
    r = da.clusterbreak(pep, size_max=2, size_min=1, max_itr=2, sim_fn=lambda x: np.ones((len(x), len(x))), cluster_fn=fn)
    # call 2 runs (itr = 2 is not > max_itr), finds its cluster oversize again and recurses as call 3 -> refused;
    # then call 1's second cluster becomes call 4 -> refused as well
    assert r.convergence == 0 and len(fn.calls) == 2 and r.calls == 4
    assert r["clustered_seq"].tolist() == []


def test_clusterbreak_on_oracle_similarities(da):
    """end to end with the reference's example parameters scaled down: similarityMH(k=4, n_hash=500) via the CPU oracle"""
    from dynaalign_amd import synth
    n = 1500
    seqs = synth.to_strings(*synth.h3n2_like(n, 20))
    seeds = O.seeds(12345, 500)

    def sim(x):
        rc, M = O.similarity_mh(x, 4, 500, seeds)
        assert rc == 0
        return M
    r = da.clusterbreak(seqs, thresh_p=0.8, size_max=100, size_min=3, sim_fn=sim, cluster_seed=1)
    again = da.clusterbreak(seqs, thresh_p=0.8, size_max=100, size_min=3, sim_fn=sim, cluster_seed=1)
    assert r["clustered_seq"].tolist() == again["clustered_seq"].tolist() and r["filtered_seq"] == again["filtered_seq"]
    assert r.convergence == 1 and r.calls > 1                      # the recursion was exercised
    # every sequence ends up exactly once, clustered or filtered
    assert sorted(r["clustered_seq"][:, 0].tolist() + r["filtered_seq"]) == sorted(seqs)
    labels, sizes = np.unique(r["clustered_seq"][:, 1], return_counts=True)
    assert sizes.min() >= 3 and sizes.max() <= 100
    assert all(int(l.split(".")[0]) <= r.calls for l in labels)
    assert [lv["itr"] for lv in r.levels] == list(range(1, r.calls + 1))
    assert r.levels[0]["n"] == n


def _planted_graph(rng, n, groups, p_in, p_out, levels=50):
    """dense-ish planted-partition graph with DISCRETE weights k/levels (MinHash-like: exact ties are common)"""
    lab = rng.randint(0, groups, n)
    iu, ju = np.triu_indices(n, 1)
    keep = rng.random_sample(iu.size) < np.where(lab[iu] == lab[ju], p_in, p_out)
    ei, ej = iu[keep].astype(np.int32), ju[keep].astype(np.int32)
    ew = (rng.randint(1, levels + 1, ei.size) / float(levels)).astype(np.float64)
    return ei, ej, ew


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_louvain_result_does_not_depend_on_the_thread_count(da, monkeypatch, seed):
    """csrc/louvain.cpp: helper threads scan neighbourhoods ahead of the deciding thread and the CSR build / aggregation are
    threaded; memberships AND modularity must be bit-identical to the one-thread run whatever the count or the timing"""
    rng = np.random.RandomState(seed)
    n = 1500
    ei, ej, ew = _planted_graph(rng, n, 12, 0.5, 0.04)
    perm = rng.permutation(ei.size)
    monkeypatch.setenv("DYNAALIGN_LOUVAIN_THREADS", "1")
    base, q0 = da.louvain(n, ei, ej, ew, resolution=1.05, seed=seed, return_modularity=True)
    assert len(np.unique(base)) > 1
    for threads in ("2", "4", "7"):
        monkeypatch.setenv("DYNAALIGN_LOUVAIN_THREADS", threads)
        for rep in range(3):                                    # the helpers' timing differs from run to run
            m, q = da.louvain(n, ei[perm], ej[perm], ew[perm], resolution=1.05, seed=seed, return_modularity=True)
            assert np.array_equal(m, base) and q == q0


def _csr_of(n, ei, ej, codes):
    """numpy construction of what da_dev_edges_to_csr builds: both directions of every off-diagonal edge sorted by (row, col),
    diagonal entries as loop codes"""
    off = ei != ej
    r = np.r_[ei[off], ej[off]].astype(np.int64)
    c = np.r_[ej[off], ei[off]].astype(np.int64)
    v = np.r_[codes[off], codes[off]]
    order = np.lexsort((c, r))
    r, c, v = r[order], c[order], v[order]
    ptr = np.zeros(n + 1, np.int64)
    np.add.at(ptr, r + 1, 1)
    ptr = np.cumsum(ptr)
    loops = np.full(n, 0xFFFF, np.uint16)
    loops[ei[~off]] = codes[~off]
    return ptr, c.astype(np.int32), v.astype(np.uint16), loops


@pytest.mark.parametrize("seed", [1, 2])
@pytest.mark.parametrize("weights", [True, False])
def test_louvain_csr_equals_louvain_on_the_edge_list(da, seed, weights):
    """da_louvain_csr (canonical CSR + weight codes, what the device hands over) clusters exactly like da_louvain on the edge list,
    loops included, weighted and unweighted"""
    rng = np.random.RandomState(seed)
    n, levels = 900, 50
    ei, ej, _ = _planted_graph(rng, n, 9, 0.4, 0.03, levels)
    codes = rng.randint(1, levels + 1, ei.size).astype(np.uint16)
    ei = np.r_[ei, np.arange(n, dtype=np.int32)]                 # + every vertex's self-loop (the similarity matrix's diagonal)
    ej = np.r_[ej, np.arange(n, dtype=np.int32)]
    codes = np.r_[codes, np.full(n, levels, np.uint16)]
    values = np.arange(levels + 1, dtype=np.float64) / levels
    want, qw = da.louvain(n, ei, ej, values[codes], resolution=1.05, seed=seed, weights=weights, return_modularity=True)
    ptr, adj, cds, loops = _csr_of(n, ei, ej, codes)
    got, qg = da.louvain_csr(n, ptr, adj, cds, loops, values, resolution=1.05, seed=seed, weights=weights, return_modularity=True)
    assert np.array_equal(got, want) and qg == qw and len(np.unique(got)) > 1


def test_louvain_csr_rejects_a_graph_that_is_not_canonical(da):
    values = np.array([0.0, 0.5, 1.0])
    none = np.full(3, 0xFFFF, np.uint16)
    ok = da.louvain_csr(3, [0, 1, 2, 2], [1, 0], [2, 2], none, values)
    assert ok.tolist() == [1, 1, 2]
    for ptr, adj, codes in (([0, 2, 3, 3], [1, 1, 0], [2, 2, 2]),      # duplicate neighbour
                            ([0, 1, 2, 2], [0, 0], [2, 2]),            # diagonal entry
                            ([0, 1, 2, 2], [1, 0], [2, 7]),            # code out of range
                            ([0, 1, 2, 2], [5, 0], [2, 2])):           # neighbour out of range
        with pytest.raises(da.DynaAlignError):
            da.louvain_csr(3, ptr, adj, codes, none, values)


def test_netcluster_modes_and_what_cluster_weight_false_means(da):
    """igraph_mode beyond "upper" (igraph's documented undirected modes; unpinned -- igraph is not importable) and the reference's
    cluster_weight = FALSE (R/clusterbreak.R:127-129): with the default cluster_func igraph::cluster_louvain(weights = NULL)
    still reads the graph's `weight` attribute, so the built-in Louvain keeps the weights; only igraph_weight = FALSE
    (an unweighted graph) drops them.  A caller's own cluster_func sees the flag."""
    from dynaalign_amd.clusterbreak import netcluster
    n = 6
    A = np.zeros((n, n))
    for a, b, w in ((0, 1, 1), (0, 2, 1), (1, 2, 1), (3, 4, 1), (3, 5, 1), (4, 5, 1), (2, 3, 50.0)):
        A[a, b] = w                                     # upper triangle only
    weighted = netcluster(A).tolist()
    assert weighted != [1, 1, 1, 2, 2, 2]              # the heavy bridge decides
    assert netcluster(A, cluster_weight=False).tolist() == weighted          # default cluster_func: the weight attribute is still used
    assert netcluster(A, igraph_weight=False).tolist() == [1, 1, 1, 2, 2, 2]   # an unweighted graph
    seen = {}

    def mine(n_, i, j, w, seed=0, weights=True):
        seen["weights"] = weights
        return np.ones(n_, np.int64)
    netcluster(A, cluster_func=mine, cluster_weight=False)
    assert seen["weights"] is False
    # modes: the same graph from the lower triangle, from both (max / min / plus / undirected)
    assert netcluster(A.T, igraph_mode="lower").tolist() == weighted
    assert netcluster(A, igraph_mode="lower").tolist() == [1, 2, 3, 4, 5, 6]   # nothing below the diagonal: no edges
    assert netcluster(A, igraph_mode="max").tolist() == weighted
    assert netcluster(A + A.T, igraph_mode="undirected").tolist() == weighted
    assert netcluster(A + A.T, igraph_mode="min").tolist() == weighted
    assert netcluster(A, igraph_mode="min").tolist() == [1, 2, 3, 4, 5, 6]
    assert netcluster(A, igraph_mode="plus").tolist() == weighted
    with pytest.raises(ValueError):
        netcluster(A, igraph_mode="directed")
    with pytest.raises(ValueError):
        netcluster(A, igraph_mode="sideways")
