"""Host-side mirror of the reference's R interface for the hot path.

    similarityMH(sequences, k=4, n_hash=50)                      reference R/RcppExports.R:15-17
    similarityNW(sequences, matrixName="BLOSUM62", gapOpen=10, gapExt=4)   reference R/RcppExports.R:34-36

Same names, argument order, defaults and error texts as the reference; the
bodies marshal to the C ABI (include/dynaalign.h) exactly as the Rcpp glue in
r_glue/ does.  Results are dense symmetric n x n float64 matrices with
``dimnames`` ("1".."n", reference src/minHash.cpp:181-185,
src/pairwiseSeqAlign.cpp:356-362).

Seeds: the reference draws its hash seeds from ``std::random_device``
(src/minHash.cpp:73,137), so it is non-deterministic by construction.  The
default here does the same.  For reproducible runs set ``seed=`` (keyword-only
extension), ``set_option("seed", s)`` or the environment variable
``DYNAALIGN_SEED``; the seed is expanded with the reference's own rule
(``HashFamily(n_hash, seed)``, src/minHash.cpp:73-81).
"""
import os

import numpy as np

from . import _capi

# options(DynaAlign.seed = , DynaAlign.devices = , DynaAlign.exchange = ) of the R glue (r_glue/)
_OPTIONS = {"seed": None, "devices": None, "exchange": "rows"}


def set_option(name, value):
    if name not in _OPTIONS:
        raise KeyError(name)
    _OPTIONS[name] = value


def get_option(name):
    return _OPTIONS[name]


class SimilarityMatrix(np.ndarray):
    """float64 (n, n) ndarray carrying R-style ``dimnames``."""

    def __new__(cls, arr):
        obj = np.asarray(arr).view(cls)
        n = obj.shape[0]
        labels = [str(i + 1) for i in range(n)]
        obj.dimnames = [labels, list(labels)]
        return obj

    def __array_finalize__(self, obj):
        self.dimnames = getattr(obj, "dimnames", None)


def pack_sequences(sequences):
    """Character vector -> (residues uint8[total], offsets int64[n+1]).

    Bytes are taken as they are (no case folding, no re-encoding), like
    ``as<std::string>`` on a CHARSXP (reference src/minHash.cpp:147)."""
    if isinstance(sequences, (str, bytes)):
        sequences = [sequences]
    bs = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in sequences]
    off = np.zeros(len(bs) + 1, np.int64)
    if bs:
        np.cumsum([len(b) for b in bs], out=off[1:])
    total = int(off[-1])
    res = np.frombuffer(b"".join(bs), np.uint8).copy() if total else np.zeros(1, np.uint8)
    return res, off


def _as_int(x, name):
    # R coerces numeric to int at the .Call boundary (Rcpp input_parameter<int>)
    try:
        return int(x)
    except Exception:
        raise TypeError("%s must be an integer" % name)


def hash_family_seeds(seed, n_hash):
    """seeds[h] = h-th raw std::mt19937(seed) output (reference src/minHash.cpp:75-80)."""
    lib = _capi.load()
    out = np.zeros(max(int(n_hash), 1), np.uint32)
    _capi.check(lib.da_hash_family_seeds(int(seed) & 0xFFFFFFFF, int(n_hash), out.ctypes.data))
    return out[:n_hash]


def _resolve_seed(seed):
    if seed is None:
        seed = _OPTIONS["seed"]
    if seed is None and os.environ.get("DYNAALIGN_SEED"):
        seed = int(os.environ["DYNAALIGN_SEED"])
    if seed is None:
        seed = _capi.load().da_random_seed()  # reference default: std::random_device{}()
    return int(seed) & 0xFFFFFFFF


def _mh_prelude(sequences, k, n_hash, seed):
    lib = _capi.load()
    res, off = pack_sequences(sequences)
    n = len(off) - 1
    k, n_hash = _as_int(k, "k"), _as_int(n_hash, "n_hash")
    # validation (and its order) lives in the library; seeds are only needed when it passes
    seeds = hash_family_seeds(_resolve_seed(seed), n_hash) if n_hash > 0 else np.zeros(1, np.uint32)
    if len(seeds) == 0:
        seeds = np.zeros(1, np.uint32)
    return lib, res, off, n, k, n_hash, seeds


def _device_opts(devices, exchange):
    """da_opts for the multi-device entry points, or None for the plain single-device call"""
    devices = _OPTIONS["devices"] if devices is None else devices
    exchange = exchange or _OPTIONS["exchange"] or "rows"
    if devices is None and os.environ.get("DYNAALIGN_DEVICES"):
        devices = [int(d) for d in os.environ["DYNAALIGN_DEVICES"].split(",") if d.strip() != ""]
    if devices is None:
        return None, None
    if isinstance(devices, int):
        devices = [devices]
    return _capi.make_opts(devices, exchange)


last_phase_ms = {}   # phase times of the most recent multi-device call (da_opts.phase_ms), by phase name


def _record_phases(keep):
    last_phase_ms.clear()
    if keep is not None:
        last_phase_ms.update(zip(_capi.DA_PHASES, [float(v) for v in keep[1]]))


def similarityMH(sequences, k=4, n_hash=50, *, seed=None, devices=None, exchange=None):
    """MinHash-estimated Jaccard similarity of k-mer sets, all pairs.

    Mirrors reference ``similarityMH`` (src/minHash.cpp:119-188): errors
    "Input sequences vector cannot be empty" / "'k' must be a positive integer" /
    "Number of hash functions must be positive" in that order; diagonal 1.0.

    devices= / set_option("devices", [...]) / DYNAALIGN_DEVICES=0,1,..: run on several GPUs of the node from this one
    process (da_similarity_mh_opts); exchange = "rows" (default), "allgather" (RCCL) or "peercopy"."""
    lib, res, off, n, k, n_hash, seeds = _mh_prelude(sequences, k, n_hash, seed)
    out = np.empty((max(n, 1), max(n, 1)), np.float64)
    opts, keep = _device_opts(devices, exchange)
    if opts is None:
        _capi.check(lib.da_similarity_mh(res.ctypes.data, off.ctypes.data, n, k, n_hash, seeds.ctypes.data,
                                         out.ctypes.data))
    else:
        import ctypes
        _capi.check(lib.da_similarity_mh_opts(res.ctypes.data, off.ctypes.data, n, k, n_hash, seeds.ctypes.data,
                                              out.ctypes.data, ctypes.addressof(opts)))
    _record_phases(keep)
    return SimilarityMatrix(out[:n, :n])


def minhash_signatures(sequences, k=4, n_hash=50, *, seed=None):
    """The (n, n_hash) uint32 signature matrix (reference src/minHash.cpp:140-157)."""
    lib, res, off, n, k, n_hash, seeds = _mh_prelude(sequences, k, n_hash, seed)
    out = np.empty((max(n, 1), max(n_hash, 1)), np.uint32)
    _capi.check(lib.da_minhash_signatures(res.ctypes.data, off.ctypes.data, n, k, n_hash, seeds.ctypes.data,
                                          out.ctypes.data))
    return out[:n, :n_hash]


def mh_counts(sequences, k=4, n_hash=50, *, seed=None, row_begin=0, row_end=None):
    """uint16 match counts (numerator at reference src/minHash.cpp:168-174) for a row block."""
    lib, res, off, n, k, n_hash, seeds = _mh_prelude(sequences, k, n_hash, seed)
    row_end = n if row_end is None else row_end
    out = np.empty((max(row_end - row_begin, 1), max(n, 1)), np.uint16)
    _capi.check(lib.da_mh_counts(res.ctypes.data, off.ctypes.data, n, k, n_hash, seeds.ctypes.data,
                                 row_begin, row_end, out.ctypes.data))
    return out[:max(row_end - row_begin, 0), :n]


def similarityNW(sequences, matrixName="BLOSUM62", gapOpen=10, gapExt=4, *, devices=None, exchange=None):
    """Fraction identity (matches / alignment length) of the reference's
    affine-gap global alignment, all pairs.

    Mirrors reference ``similarityNW`` (src/pairwiseSeqAlign.cpp:331-365): no input
    validation beyond "Invalid substitution matrix name: %s" and the lazily raised
    "Invalid amino acid in sequence1/2: %c"; n == 0 gives a 0 x 0 matrix.
    devices= / exchange=: as for similarityMH (da_similarity_nw_opts)."""
    lib = _capi.load()
    res, off = pack_sequences(sequences)
    n = len(off) - 1
    out = np.empty((max(n, 1), max(n, 1)), np.float64)
    name = matrixName.encode("latin-1") if isinstance(matrixName, str) else bytes(matrixName)
    opts, keep = _device_opts(devices, exchange)
    if opts is None:
        _capi.check(lib.da_similarity_nw(res.ctypes.data, off.ctypes.data, n, name, _as_int(gapOpen, "gapOpen"),
                                         _as_int(gapExt, "gapExt"), out.ctypes.data))
    else:
        import ctypes
        _capi.check(lib.da_similarity_nw_opts(res.ctypes.data, off.ctypes.data, n, name, _as_int(gapOpen, "gapOpen"),
                                              _as_int(gapExt, "gapExt"), out.ctypes.data, ctypes.addressof(opts)))
    _record_phases(keep)
    return SimilarityMatrix(out[:n, :n])


def nw_pairs(sequences, matrixName="BLOSUM62", gapOpen=10, gapExt=4, *, row_begin=0, row_end=None):
    """(matches, length, score) int32 arrays for a row block: the integers the
    reference divides at src/pairwiseSeqAlign.cpp:311, plus M[m][n]."""
    lib = _capi.load()
    res, off = pack_sequences(sequences)
    n = len(off) - 1
    row_end = n if row_end is None else row_end
    r = max(row_end - row_begin, 0)
    mt = np.zeros((max(r, 1), max(n, 1)), np.int32)
    ln = np.zeros_like(mt)
    sc = np.zeros_like(mt)
    name = matrixName.encode("latin-1") if isinstance(matrixName, str) else bytes(matrixName)
    _capi.check(lib.da_nw_pairs(res.ctypes.data, off.ctypes.data, n, name, _as_int(gapOpen, "gapOpen"),
                                _as_int(gapExt, "gapExt"), row_begin, row_end, mt.ctypes.data, ln.ctypes.data,
                                sc.ctypes.data))
    return mt[:r, :n], ln[:r, :n], sc[:r, :n]


def quantile_type7(hist, values, p):
    """R's quantile(x, p, type = 7) of {values[b] repeated hist[b] times} (values ascending)."""
    lib = _capi.load()
    h = np.ascontiguousarray(hist, np.uint64)
    v = np.ascontiguousarray(values, np.float64)
    q = np.zeros(1, np.float64)
    _capi.check(lib.da_quantile_type7(h.ctypes.data, v.ctypes.data, len(h), float(p), q.ctypes.data))
    return float(q[0])


def similarityMH_edges(sequences, k=4, n_hash=50, thresh_p=0.8, *, seed=None):
    """similarityMH followed by clusterbreak's threshold step, fused on the device.

    Equivalent to (reference R/clusterbreak.R:217-221, netcluster :122-124)

        S <- similarityMH(sequences, k, n_hash)
        threshold <- quantile(S[upper.tri(S)], thresh_p)
        S[S < threshold] <- 0                      # edges = non-zero entries of the upper triangle + diagonal

    but returns only ``(threshold, i, j, weight)`` -- the surviving entries with i <= j (0-based, sorted),
    never the dense matrix."""
    lib, res, off, n, k, n_hash, seeds = _mh_prelude(sequences, k, n_hash, seed)
    return _edges_one_pass(lambda h, thr, cnt: lib.da_similarity_mh_edges_begin(
        res.ctypes.data, off.ctypes.data, n, k, n_hash, seeds.ctypes.data, float(thresh_p), h, thr, cnt))


def _edges_one_pass(begin):
    """*_edges_begin -> da_edges_fetch -> da_edges_free: the pipeline runs once (the size-query form runs it twice)"""
    import ctypes
    lib = _capi.load()
    handle = ctypes.c_void_p()
    thr = np.zeros(1, np.float64)
    cnt = np.zeros(1, np.int64)
    _capi.check(begin(ctypes.addressof(handle), thr.ctypes.data, cnt.ctypes.data))
    try:
        m = int(cnt[0])
        ei, ej, ew = np.empty(max(m, 1), np.int32), np.empty(max(m, 1), np.int32), np.empty(max(m, 1), np.float64)
        _capi.check(lib.da_edges_fetch(handle, m, ei.ctypes.data, ej.ctypes.data, ew.ctypes.data))
    finally:
        lib.da_edges_free(handle)
    return float(thr[0]), ei[:m], ej[:m], ew[:m]


def similarityNW_edges(sequences, matrixName="BLOSUM62", gapOpen=10, gapExt=4, thresh_p=0.8):
    """similarityNW followed by clusterbreak's threshold step (reference R/clusterbreak.R:217-221), fused on
    the device like similarityMH_edges: returns ``(threshold, i, j, weight)`` for the surviving entries with
    i <= j (0-based, sorted).  Sequences up to 127 residues; empty sequences are refused (NaN similarities)."""
    lib = _capi.load()
    res, off = pack_sequences(sequences)
    n = len(off) - 1
    name = matrixName.encode("latin-1") if isinstance(matrixName, str) else bytes(matrixName)
    go, ge = _as_int(gapOpen, "gapOpen"), _as_int(gapExt, "gapExt")
    return _edges_one_pass(lambda h, thr, cnt: lib.da_similarity_nw_edges_begin(
        res.ctypes.data, off.ctypes.data, n, name, go, ge, float(thresh_p), h, thr, cnt))
