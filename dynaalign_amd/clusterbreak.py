"""clusterbreak -- the caller of the hot path, restated (BASELINE config 5).

    clusterbreak(pep, thresh_p=0.8, size_max=10, size_min=3, max_itr=10000, sim_fn=..., cluster_fn=..., cluster_wt=True)
                                                                              reference R/clusterbreak.R:180-275
    netcluster(pepmat, ...)                                                   reference R/clusterbreak.R:112-136

Same argument names, defaults, messages and bookkeeping as the R functions.  Per recursion level
(``cluster_recursive``, R/clusterbreak.R:203-259):

    pep.sim   <- sim_fn(pep)                                          :217   the hot-path call
    threshold <- quantile(pep.sim[upper.tri(pep.sim)], thresh_p)      :219   R type 7
    pep.sim[pep.sim < threshold] <- 0                                 :221
    c.index   <- netcluster(pep.sim, ...)                             :222   upper triangle incl. the 1.0 diagonal
    sizes, id.itr (> size_max), id.rm (< size_min), labels "<itr>.<cluster>", recursion   :224-254

Two ways to produce a level's thresholded graph, with identical results:

  * ``sim_fn`` (the reference's contract): any function sequences -> dense symmetric matrix.  The three R
    statements above are restated literally in `threshold_edges_dense`.  Works with ``similarityMH`` /
    ``similarityNW`` of this package, or -- in the tests -- with the CPU oracle.
  * ``session`` (device fast path): a `MinHashSession` keeps the signatures of all sequences in HBM; a level is
    K1b + K2 + histogram + exact type-7 quantile + edge extraction on the index subset (`MinHashSession.edges`),
    and only the surviving edges leave the GPU.  The dense matrix never exists (80 GB at N = 100k).

The default ``cluster_fn`` is `louvain` (resolution 1.05, the reference's default, :115-116/:186-187) on the
library's host-side multilevel implementation (`da_louvain`); igraph is not importable here and its result
depends on R's RNG stream, so memberships are reproducible given ``cluster_seed`` but are not igraph's.
"""
import sys
import time

import numpy as np

from . import _capi

__all__ = ["clusterbreak", "netcluster", "louvain", "threshold_edges_dense", "ClusterbreakResult"]


def louvain(n, ei, ej, ew, resolution=1.05, seed=0, weights=True, return_modularity=False):
    """igraph::cluster_louvain(g, weights = E(g)$weight, resolution)$membership on an edge list (0-based,
    i == j = self-loop).  weights=False clusters the unweighted graph (a graph built with igraph_weight = FALSE; NOT what
    netcluster's cluster_weight = FALSE does with the default cluster_func -- see _run_cluster_fn).  Returns int32 ids
    starting at 1 (R's numbering)."""
    lib = _capi.load()
    ei = np.ascontiguousarray(ei, np.int32)
    ej = np.ascontiguousarray(ej, np.int32)
    ew = np.ascontiguousarray(ew, np.float64) if weights else np.ones(len(ei), np.float64)
    if not (len(ei) == len(ej) == len(ew)):
        raise ValueError("edge arrays differ in length")
    member = np.zeros(max(int(n), 1), np.int32)
    q = np.zeros(1, np.float64)
    lv = np.zeros(1, np.int32)
    _capi.check(lib.da_louvain(int(n), len(ei), ei.ctypes.data, ej.ctypes.data, ew.ctypes.data, float(resolution),
                               int(seed) & 0xFFFFFFFF, member.ctypes.data, q.ctypes.data if return_modularity else None,   # (the modularity is
                               lv.ctypes.data))                                                                           # a sweep over all edges)
    member = member[:int(n)]
    return (member, float(q[0])) if return_modularity else member


def louvain_csr(n, ptr, adj, codes, loop_codes, values, resolution=1.05, seed=0, weights=True, return_modularity=False):
    """`louvain` on a graph that is already canonical (da_louvain_csr): symmetric CSR, weights as uint16 codes into `values`,
    self-loops per vertex as codes (0xFFFF = none).  What MinHashSession.edges_csr hands over -- sorted on the device, no host-side
    sort of the edge list.  weights=False clusters the unweighted graph (every code stands for 1.0).  Same result as `louvain`
    on the corresponding edge list."""
    lib = _capi.load()
    ptr = np.ascontiguousarray(ptr, np.int64)
    adj = np.ascontiguousarray(adj, np.int32)
    codes = np.ascontiguousarray(codes, np.uint16)
    loop_codes = np.ascontiguousarray(loop_codes, np.uint16)
    values = np.ascontiguousarray(values, np.float64) if weights else np.ones(len(values), np.float64)
    member = np.zeros(max(int(n), 1), np.int32)
    q = np.zeros(1, np.float64)
    lv = np.zeros(1, np.int32)
    _capi.check(lib.da_louvain_csr(int(n), ptr.ctypes.data, adj.ctypes.data, codes.ctypes.data, loop_codes.ctypes.data, values.ctypes.data,
                                   len(values), float(resolution), int(seed) & 0xFFFFFFFF, member.ctypes.data,
                                   q.ctypes.data if return_modularity else None, lv.ctypes.data))
    member = member[:int(n)]
    return (member, float(q[0])) if return_modularity else member


def _quantile_type7_sorted_parts(x, p):
    """stats::quantile(x, p, type = 7) for one probability, R's own arithmetic (quantile.default):
    index = 1 + (n - 1) p; lo = floor, hi = ceiling; qs = x[lo]; if index > lo and x[hi] != qs:
    qs = (1 - h) qs + h x[hi], h = index - lo."""
    n = x.size
    if n == 0:
        return float("nan")                    # quantile(numeric(0), p) is NA
    if np.isnan(x).any():
        raise ValueError("missing values and NaN's not allowed if 'na.rm' is FALSE")
    index = 1.0 + (n - 1) * float(p)
    lo, hi = int(np.floor(index)), int(np.ceil(index))
    part = np.partition(x, sorted({lo - 1, hi - 1}))
    qs, xh = float(part[lo - 1]), float(part[hi - 1])
    if index > lo and xh != qs:
        h = index - lo
        qs = (1.0 - h) * qs + h * xh
    return qs


def threshold_edges_dense(sim, thresh_p):
    """R/clusterbreak.R:219-221 + the graph netcluster builds (:122-124), from a dense matrix:
    threshold = quantile(S[upper.tri(S)], thresh_p); S[S < threshold] <- 0; edges = non-zero entries of the upper
    triangle INCLUDING the diagonal (graph_from_adjacency_matrix(mode = "upper", weighted = TRUE): a zero weight is
    no edge).  Returns (threshold, i, j, weight), i <= j, 0-based, sorted by (i, j)."""
    S = np.array(sim, np.float64)              # a copy: the reference modifies its local pep.sim
    if S.ndim != 2 or S.shape[0] != S.shape[1]:
        raise ValueError("Input must be a square pairwise similarity matrix")          # netcluster, :118-120
    n = S.shape[0]
    upper = np.triu(np.ones((n, n), bool), 1)                                          # upper.tri(pep.sim)
    thr = _quantile_type7_sorted_parts(S[upper], thresh_p)
    del upper
    if not np.isnan(thr):
        S[S < thr] = 0.0
    U = np.triu(S)                                                                     # mode = "upper": diagonal included
    i, j = np.nonzero(U)                                                               # row-major = sorted by (i, j)
    return thr, i.astype(np.int32), j.astype(np.int32), U[i, j]


def netcluster(pepmat, igraph_mode="upper", igraph_weight=True, cluster_func=None, cluster_weight=True, seed=0):
    """reference netcluster (R/clusterbreak.R:112-136) on a dense matrix: graph from the upper triangle incl. the
    diagonal, then cluster_func(n, i, j, w, seed=..., weights=...) -> numeric vector of cluster ids."""
    M = np.asarray(pepmat, np.float64)
    if M.ndim != 2 or M.shape[0] != M.shape[1]:
        raise ValueError("Input must be a square pairwise similarity matrix")
    n = M.shape[0]
    i, j = np.triu_indices(n, 0)
    # igraph::graph_from_adjacency_matrix(mode = ...) for the undirected modes (igraph's documentation; igraph itself is not
    # importable here, so everything but "upper" -- the reference's default and the only mode clusterbreak uses -- is unpinned)
    if igraph_mode == "upper":
        w = M[i, j]
    elif igraph_mode == "lower":
        w = M[j, i]
    elif igraph_mode in ("max", "undirected"):
        w = np.maximum(M[i, j], M[j, i])
    elif igraph_mode == "min":
        w = np.minimum(M[i, j], M[j, i])
    elif igraph_mode == "plus":
        w = np.where(i == j, M[i, j], M[i, j] + M[j, i])
    elif igraph_mode == "directed":
        raise ValueError("igraph_mode = 'directed' builds a directed graph; igraph::cluster_louvain only works with undirected graphs")
    else:
        raise ValueError("igraph_mode must be one of 'upper', 'lower', 'max', 'undirected', 'min', 'plus'")
    nz = w != 0.0
    i, j, w = i[nz].astype(np.int32), j[nz].astype(np.int32), w[nz]
    if not igraph_weight:
        w = np.ones_like(w)
    return _run_cluster_fn(cluster_func or louvain, n, i, j, w, cluster_weight, seed)


def _run_cluster_fn(cluster_fn, n, i, j, w, cluster_wt, seed):
    # netcluster (R/clusterbreak.R:125-129) calls cluster_func(network, weights = E(network)$weight) or cluster_func(network).
    # For the default cluster_func the second form is NOT unweighted: igraph::cluster_louvain(weights = NULL) takes the graph's
    # `weight` edge attribute when there is one, and graph_from_adjacency_matrix(weighted = TRUE) always creates it -- so the
    # built-in Louvain keeps the weights whatever cluster_wt says (an unweighted graph arrives here with w = 1).  A caller's own
    # cluster_fn sees the flag, as the reference's sees the presence of the `weights` argument.
    builtin = cluster_fn is louvain
    out = cluster_fn(n, i, j, w, seed=seed, weights=True if builtin else bool(cluster_wt))
    out = np.asarray(out)
    if out.ndim != 1 or out.size != n or not np.issubdtype(out.dtype, np.number):
        raise ValueError("Wrong clustering output format. Output should be a numeric vector of cluster assignment.")  # :134
    return out.astype(np.int64)


class ClusterbreakResult(dict):
    """list(clustered_seq = <n x 2: sequence, "<itr>.<cluster>">, filtered_seq = <sequences>) of the reference
    (R/clusterbreak.R:257-258), plus bookkeeping the reference only prints: .convergence (1 / 0, :200,:213),
    .calls (state$itr, :270), .levels (per call: itr, n, threshold, edges, clusters, seconds by phase)."""


def clusterbreak(pep, thresh_p=0.8, size_max=10, size_min=3, max_itr=10000, sim_fn=None, cluster_fn=None,
                 cluster_wt=True, *, session=None, cluster_seed=0, verbose=False, log=None):
    """Recursive quantile-threshold + Louvain splitting, reference clusterbreak (R/clusterbreak.R:180-275).

    pep        sequences (character vector)
    sim_fn     sequences -> similarity matrix; default = the reference's default, similarityMH(x, k=2, n_hash=50)
    session    a MinHashSession over `pep`: levels then run on the device edge path and sim_fn is not used
    cluster_fn (n, i, j, w, seed=, weights=) -> ids; default `louvain` with resolution 1.05
    cluster_seed  call number c of the recursion clusters with seed cluster_seed + c (the reference draws from
               R's global RNG instead)
    """
    if size_max <= size_min:
        raise ValueError("size_max must be greater than size_min")                      # :189-191
    pep = list(pep)
    if len(pep) == 0:
        raise ValueError("empty input sequence vector")                                  # :192-194
    if session is not None and session.n != len(pep):
        raise ValueError("session holds %d sequences, pep has %d" % (session.n, len(pep)))
    if sim_fn is None and session is None:
        from .similarity import similarityMH
        sim_fn = lambda x: similarityMH(x, k=2, n_hash=50)                               # noqa: E731  (:185)
    import os
    csr_path = session is not None and cluster_fn is None and not os.environ.get("DYNAALIGN_CLUSTERBREAK_NO_CSR")
    cluster_fn = cluster_fn or louvain
    out_seq, out_lab, filtered = [], [], []                                              # state$out.df, state$filter.df
    state = {"itr": 1, "convergence": 1}                                                 # :199-200
    levels = []

    def log_message(msg, level="INFO"):                                                  # :206-209
        if verbose or level != "INFO":
            print("[%s] %s: %s" % (time.strftime("%H:%M:%S"), level, msg), file=log or sys.stdout)

    def level_edges(idx):
        m = len(idx)
        if session is not None:
            if m < 2:                          # quantile(numeric(0)) is NA: nothing is removed, the 1.0 diagonal stays
                return float("nan"), np.zeros(m, np.int32), np.zeros(m, np.int32), np.ones(m, np.float64)
            return session.edges(idx, thresh_p, sort=False)
        return threshold_edges_dense(sim_fn([pep[t] for t in idx]), thresh_p)

    def cluster_recursive(idx):
        if state["itr"] > max_itr:                                                       # :211-215
            log_message("Maximum function calls reached", "WARNING")
            state["convergence"] = 0
            return
        itr = state["itr"]
        t0 = time.perf_counter()
        m = len(idx)
        seed_c = (int(cluster_seed) + itr) & 0xFFFFFFFF
        if csr_path and m >= 2:
            # device edge path + built-in Louvain: the graph arrives as canonical CSR sorted on the device (same graph, same result)
            thr, n_edges, ptr, adj, codes, loops, values = session.edges_csr(idx, thresh_p)
            t1 = time.perf_counter()
            c_index = louvain_csr(m, ptr, adj, codes, loops, values, seed=seed_c, weights=True).astype(np.int64)   # :222 (built-in: see _run_cluster_fn)
            del ptr, adj, codes, loops
        else:
            thr, ei, ej, ew = level_edges(idx)
            n_edges = len(ei)
            t1 = time.perf_counter()
            c_index = _run_cluster_fn(cluster_fn, m, ei, ej, ew, cluster_wt, seed_c)     # :222
            del ei, ej, ew
        t2 = time.perf_counter()
        c_size = np.bincount(c_index, minlength=1)[1:] if m else np.zeros(0, np.int64)   # tabulate(c.index)   :224
        ids = np.arange(1, len(c_size) + 1)
        id_itr = ids[c_size > size_max]                                                  # :225
        id_rm = ids[c_size < size_min]                                                   # :226
        in_rm = np.isin(c_index, id_rm)
        in_itr = np.isin(c_index, id_itr)
        filtered.extend(pep[idx[t]] for t in np.nonzero(in_rm)[0])                       # :228
        keep = ~in_rm & ~in_itr                                                          # :232 / :238 (in_itr is empty in the first case)
        nkeep = int(keep.sum())
        if nkeep == 1:
            # pep.ref[mask, ] with one TRUE row drops to a character vector, nrow() is NULL and `if (NULL > 0)` stops
            raise RuntimeError("argument is of length zero")
        for t in np.nonzero(keep)[0]:                                                    # :233-236 / :239-243
            out_seq.append(pep[idx[t]])
            out_lab.append("%d.%d" % (itr, c_index[t]))
        levels.append({"itr": itr, "n": m, "threshold": thr, "edges": int(n_edges), "clusters": int(len(c_size)),
                       "oversize": int(len(id_itr)), "similarity_s": t1 - t0, "cluster_s": t2 - t1})
        log_message("call %d: n=%d threshold=%g edges=%d clusters=%d oversize=%d (similarity %.3f s, clustering %.3f s)"
                    % (itr, m, thr, n_edges, len(c_size), len(id_itr), t1 - t0, t2 - t1))
        if len(id_itr) == 0:
            return
        # :246-254 -- oversize clusters in order of first appearance (unique(pep.new[,2])), each a new call
        sub = c_index[in_itr]
        _, first = np.unique(sub, return_index=True)
        for cid in sub[np.sort(first)]:
            state["itr"] += 1                                                            # :252
            cluster_recursive(idx[c_index == cid])                                       # :251,:253

    cluster_recursive(np.arange(len(pep), dtype=np.int64))
    if verbose:                                                                          # :265-270
        print("\nClustering complete:" if state["convergence"] == 1 else "\nClustering incomplete, consider adjusting parameters:",
              file=log or sys.stdout)
        print("Total function calls (clusters broken): %d" % state["itr"], file=log or sys.stdout)
    res = ClusterbreakResult(clustered_seq=np.array(list(zip(out_seq, out_lab)), dtype=object).reshape(-1, 2),
                             filtered_seq=filtered)
    res.convergence, res.calls, res.levels = state["convergence"], state["itr"], levels
    return res
