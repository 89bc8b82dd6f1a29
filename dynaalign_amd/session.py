"""Device-resident MinHash session for the caller's recursion.

clusterbreak re-calls ``sim_fn`` on every subset it splits off (reference R/clusterbreak.R:203-259,
``:246-254``), and each call of the reference re-uploads, re-hashes and re-seeds.  A signature does not
depend on which other sequences are in the call, so with the hash seed held fixed the signatures of the
full set can stay in HBM and a recursion level only needs K1b (codes of the subset's rows) + K2:

    s = MinHashSession(sequences, k=4, n_hash=500, seed=12345)      # upload + K1 once
    S = s.similarity(idx)                                           # == similarityMH(sequences[idx], 4, 500, seed=12345)
    thr, i, j, w = s.edges(idx, thresh_p=0.8)                       # == similarityMH_edges(sequences[idx], ...)

The only difference to calling the reference per level is the random stream (the reference draws fresh
seeds per call, src/minHash.cpp:73,137); the contract -- MinHash estimates under one hash family -- holds.
The recursion itself (and its Louvain step) is restated in dynaalign_amd/clusterbreak.py, which drives this session.
"""
import numpy as np
import torch

from . import _capi, device
from .similarity import SimilarityMatrix, hash_family_seeds, pack_sequences, quantile_type7, _resolve_seed


class MinHashSession:
    def __init__(self, sequences, k=4, n_hash=50, *, seed=None, device_name="cuda", reserve=True):
        res, off = pack_sequences(sequences)
        self.n, self.k, self.n_hash = len(off) - 1, int(k), int(n_hash)
        self.seed = _resolve_seed(seed)
        self.seeds = hash_family_seeds(self.seed, self.n_hash) if self.n_hash > 0 else np.zeros(1, np.uint32)
        self.ds = device.DeviceSequences(res, off, device_name)
        # validation (order and messages of the reference) happens in the library call
        self.sig, _ = device.minhash_signatures(self.ds, self.k, self.n_hash, self.seeds, want_planes=False)
        if reserve and self.n >= 2:
            # the first recursion level needs an n x n uint16 count matrix (20 GB at n = 100k) and a fresh hipMalloc of that size
            # costs 0.3 - 2.3 s depending on the box: take it here once, hand it to the caching allocator, and every level finds it
            try:
                free_b, _ = torch.cuda.mem_get_info(self.sig.device)
                if 2 * self.n * self.n <= free_b // 2:
                    torch.empty(self.n * self.n, dtype=torch.int16, device=self.sig.device)
            except RuntimeError:
                pass

    def _subset(self, idx):
        if idx is None:
            return self.sig, self.n
        idx_t = torch.as_tensor(np.ascontiguousarray(idx, np.int64), device=self.sig.device)
        if idx_t.numel() == 0:
            _capi.check(_capi.DA_ERR_EMPTY_INPUT)
        return self.sig.index_select(0, idx_t).contiguous(), int(idx_t.numel())

    def planes(self, idx=None):
        sig, m = self._subset(idx)
        return device.mh_planes(sig, m, self.n_hash), m

    def similarity(self, idx=None):
        """dense float64 matrix of the subset (host copy), rows/columns in the order of idx"""
        planes, m = self.planes(idx)
        out = device.mh_compare(planes, m, self.n_hash)
        return SimilarityMatrix(out.cpu().numpy())

    def edges_csr(self, idx=None, thresh_p=0.8):
        """The thresholded graph of the subset as the canonical symmetric CSR clusterbreak.louvain_csr takes -- sorted ON THE DEVICE
        (da_dev_edges_to_csr), so neither a host-side sort nor 16 bytes per edge: (threshold, n_edges, ptr, adj, codes, loop_codes,
        values) with weight of entry k = values[codes[k]]; n_edges counts i <= j entries like `edges`."""
        import os, sys, time
        trace = os.environ.get("DYNAALIGN_TRACE") is not None
        marks = [("start", time.perf_counter())]

        def mark(what):
            if trace:
                torch.cuda.synchronize()
                marks.append((what, time.perf_counter()))
        planes, m = self.planes(idx)
        if m < 2:
            raise _capi.DynaAlignError(_capi.DA_ERR_BAD_ARG, "the threshold is a quantile of the strict upper triangle: need >= 2 sequences")
        mark("codes")
        cnt = device.mh_compare(planes, m, self.n_hash, kind=_capi.DA_OUT_COMPACT)
        mark("compare")
        nbins = self.n_hash + 1
        hist = device.upper_histogram(cnt, m, nbins).cpu().numpy().astype(np.uint64)
        values = np.arange(nbins, dtype=np.float64) / self.n_hash           # src/minHash.cpp:174
        thr = quantile_type7(hist, values, thresh_p)
        keep = (~(values < thr)) & (np.arange(nbins) != 0)
        cap = int(hist[keep].sum()) + m
        mark("histogram + quantile")
        ei, ej, ev, c = device.extract_edges(cnt, m, keep, cap)
        got = int(c.item())
        assert got == cap, (got, cap)
        del cnt
        mark("extract")
        ptr, adj, codes, loops = device.edges_to_csr(ei, ej, ev, got, m)
        mark("edges -> CSR")
        out = (thr, got, ptr.cpu().numpy(), adj.cpu().numpy(), codes.cpu().numpy().view(np.uint16), loops.cpu().numpy().view(np.uint16),
               values)
        mark("device -> host")
        if trace and m >= 20000:
            print("[dynaalign] edges_csr m=%d: %s" % (m, ", ".join("%s %.1f ms" % (w, (t - marks[i][1]) * 1e3)
                                                                    for i, (w, t) in enumerate(marks[1:]))), file=sys.stderr)
        return out

    def edges(self, idx=None, thresh_p=0.8, sort=True):
        """(threshold, i, j, weight) of the subset after clusterbreak's quantile threshold, i <= j positions in idx;
        sort=False leaves the edges in the order the device appended them (da_louvain canonicalises anyway)"""
        planes, m = self.planes(idx)
        if m < 2:
            raise _capi.DynaAlignError(_capi.DA_ERR_BAD_ARG, "the threshold is a quantile of the strict upper triangle: need >= 2 sequences")
        cnt = device.mh_compare(planes, m, self.n_hash, kind=_capi.DA_OUT_COMPACT)
        nbins = self.n_hash + 1
        hist = device.upper_histogram(cnt, m, nbins).cpu().numpy().astype(np.uint64)
        values = np.arange(nbins, dtype=np.float64) / self.n_hash           # src/minHash.cpp:174
        thr = quantile_type7(hist, values, thresh_p)
        keep = (~(values < thr)) & (np.arange(nbins) != 0)
        cap = int(hist[keep].sum()) + m
        ei, ej, ev, c = device.extract_edges(cnt, m, keep, cap)
        got = int(c.item())
        assert got == cap, (got, cap)
        ei, ej, ev = ei[:got].cpu().numpy(), ej[:got].cpu().numpy(), ev[:got].cpu().numpy().view(np.uint16)
        if not sort:
            return thr, ei, ej, values[ev]
        order = np.lexsort((ej, ei))
        return thr, ei[order], ej[order], values[ev[order]]
