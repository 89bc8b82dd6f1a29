"""Device-resident front end: torch tensors in, torch tensors out.

PyTorch is plumbing here (HBM allocation, the current HIP stream,
torch.distributed); every computation is a launch of the library's HIP kernels
through the ``da_dev_*`` entry points of include/dynaalign.h.  Calls are
asynchronous on ``torch.cuda.current_stream()``.
"""
import ctypes

import numpy as np
import torch

from . import _capi
from ._capi import DA_OUT_COMPACT, DA_OUT_F64


def _call(fn, *args):
    """a library call that allocates on the device: when it runs out of memory, hand PyTorch's cached blocks back to the driver and try
    once more (the library's own out-of-memory path can only release its own parked buffers; a tensor PyTorch freed -- e.g. the count
    matrix MinHashSession reserves -- sits in PyTorch's allocator, invisible to hipMalloc)"""
    rc = fn(*args)
    if rc == _capi.DA_ERR_NOMEM:
        torch.cuda.empty_cache()
        rc = fn(*args)
    _capi.check(rc)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(t, name):
    if not t.is_cuda:
        raise _capi.DynaAlignError(_capi.DA_ERR_NO_DEVICE, "%s must live in HBM (dynaalign_amd has no CPU path)" % name)


class DeviceSequences:
    """Packed sequences resident in HBM: residues uint8[total], offsets int64[n+1]."""

    def __init__(self, residues, offsets, device="cuda"):
        offsets = np.ascontiguousarray(offsets, np.int64)
        self.n = len(offsets) - 1
        self.total = int(offsets[-1]) if self.n >= 0 else 0
        lens = np.diff(offsets) if self.n > 0 else np.zeros(0, np.int64)
        self.max_len = int(lens.max()) if self.n > 0 else 0
        self.offsets_host = offsets
        res = np.ascontiguousarray(residues, np.uint8)[:max(self.total, 1)]
        self.residues = torch.from_numpy(res.copy()).to(device)
        self.offsets = torch.from_numpy(offsets.copy()).to(device)
        self.codes = None  # filled by nw_encode


def sig_ld(n_hash):
    return int(_capi.load().da_sig_ld(int(n_hash)))


def planes_words(n, n_hash):
    return int(_capi.load().da_mh_planes_words(int(n), int(n_hash)))


class Planes:
    """Operand of mh_compare: the (opaque, blocked) bit-plane buffer plus the number of planes it holds per
    group of 32 hash functions (8 / 12 / 16: dictionary codes from da_dev_mh_planes, 32: raw signature bits)."""

    def __init__(self, tensor, bits):
        self.tensor, self.bits = tensor, int(bits)

    def data_ptr(self):
        return self.tensor.data_ptr()

    @property
    def device(self):
        return self.tensor.device

    @property
    def is_cuda(self):
        return self.tensor.is_cuda


def planes_workspace_bytes(n, n_hash):
    return int(_capi.load().da_mh_planes_workspace_bytes(int(n), int(n_hash)))


def minhash_signatures(ds, k, n_hash, seeds, out=None, planes=None, want_planes=True, raw_planes=False, work=None,
                       min_plane_bits=0):
    """K1 (+ K1b).  Returns (sig, planes): `sig` is the int32 tensor (n, sig_ld(n_hash)) holding the uint32
    signatures in columns [0, n_hash); `planes` is the Planes operand of mh_compare (None if want_planes is
    False), see mh_planes.  raw_planes=True is min_plane_bits=32."""
    lib = _capi.load()
    if not torch.is_tensor(seeds):
        seeds = torch.from_numpy(np.ascontiguousarray(seeds, np.uint32).view(np.int32).copy()).to(ds.residues.device)
    _require_cuda(ds.residues, "residues")
    ld = sig_ld(n_hash) if n_hash > 0 else 32
    if out is None:
        out = torch.empty((max(ds.n, 1), ld), dtype=torch.int32, device=ds.residues.device)
    _call(lib.da_dev_minhash_signatures, ds.residues.data_ptr(), ds.offsets.data_ptr(), ds.n, ds.total,
                                              ds.max_len, int(k), int(n_hash), seeds.data_ptr(), out.data_ptr(),
                                              out.stride(0), _stream())
    if not want_planes and planes is None:
        return out, None
    return out, mh_planes(out, ds.n, n_hash, planes, work, 32 if raw_planes else min_plane_bits)


def mh_planes(sig, n, n_hash, planes=None, work=None, min_plane_bits=0):
    """K1b.  Signature tensor -> Planes operand: the signatures' exact dictionary codes, bit-transposed, with
    8 / 12 / 16 planes per 32 hash functions (as few as the data needs, at least min_plane_bits) when
    n <= 131068; raw 32 planes otherwise or with min_plane_bits=32.  Synchronises the current stream once on
    the dictionary route.  `planes` / `work` may be preallocated: planes_words(n, n_hash) int32 and
    planes_workspace_bytes(n, n_hash) uint8."""
    lib = _capi.load()
    _require_cuda(sig, "signatures")
    if isinstance(planes, Planes):
        planes = planes.tensor
    if planes is None:
        planes = torch.empty(max(planes_words(n, n_hash), 4), dtype=torch.int32, device=sig.device)
    if work is None:
        work = torch.empty(planes_workspace_bytes(n, n_hash), dtype=torch.uint8, device=sig.device)
    bits = ctypes.c_int(0)
    _call(lib.da_dev_mh_planes, sig.data_ptr(), sig.stride(0), n, int(n_hash), int(min_plane_bits),
                                     work.data_ptr(), work.numel(), planes.data_ptr(), planes.numel(),
                                     ctypes.byref(bits), _stream())
    return Planes(planes, bits.value)


def _alloc_out(rows, n, kind, device, out):
    if out is not None:
        return out
    dt = torch.float64 if kind == DA_OUT_F64 else torch.int16  # int16 holds the uint16 bit pattern
    try:
        return torch.empty((max(rows, 1), max(n, 1)), dtype=dt, device=device)
    except torch.OutOfMemoryError:
        # buffers the library parked for its next call are invisible to torch's allocator: hand them back and try once more
        torch.cuda.empty_cache()
        _capi.load().da_release_device_memory()
        return torch.empty((max(rows, 1), max(n, 1)), dtype=dt, device=device)


def mh_compare(planes, n, n_hash, row_begin=0, row_end=None, symmetric=None, kind=DA_OUT_F64, out=None):
    """K2.  `planes` is the Planes operand from minhash_signatures.  Rows [row_begin,row_end) of the
    n x n similarity (float64) or match-count (uint16 in an int16 tensor) matrix."""
    lib = _capi.load()
    _require_cuda(planes, "bit planes")
    row_end = n if row_end is None else row_end
    if symmetric is None:
        symmetric = (row_begin == 0 and row_end == n)
    out = _alloc_out(row_end - row_begin, n, kind, planes.device, out)
    _call(lib.da_dev_mh_compare, planes.data_ptr(), planes.bits, n, int(n_hash), row_begin,
                                      row_end, 1 if symmetric else 0, kind, out.data_ptr(), out.stride(0), _stream())
    return out


def nw_encode(ds):
    """K0.  Fills ds.codes; returns the int32 flag tensor (0 = all residues valid)."""
    lib = _capi.load()
    _require_cuda(ds.residues, "residues")
    ds.codes = torch.empty_like(ds.residues)
    bad = torch.zeros(1, dtype=torch.int32, device=ds.residues.device)
    _call(lib.da_dev_nw_encode, ds.residues.data_ptr(), ds.total, ds.codes.data_ptr(), bad.data_ptr(), _stream())
    return bad


def decode_bad_position(flag_value):
    """int from nw_encode's flag -> byte position of the first invalid residue, or None."""
    return None if flag_value == 0 else (2 ** 31 - 1) - int(flag_value)


def nw(ds, matrix_name="BLOSUM62", gap_open=10, gap_ext=4, row_begin=0, row_end=None, symmetric=None,
       kind=DA_OUT_F64, out=None, score=None):
    """K3.  ds.codes must be filled (nw_encode)."""
    lib = _capi.load()
    if ds.codes is None:
        raise ValueError("call nw_encode(ds) first")
    mid = lib.da_matrix_id(matrix_name.encode("latin-1"))
    if mid < 0:
        _capi.check(_capi.DA_ERR_BAD_MATRIX)
    n = ds.n
    row_end = n if row_end is None else row_end
    if symmetric is None:
        symmetric = (row_begin == 0 and row_end == n)
    out = _alloc_out(row_end - row_begin, n, kind, ds.residues.device, out)
    _call(lib.da_dev_nw, ds.codes.data_ptr(), ds.offsets.data_ptr(), n, ds.max_len, mid, int(gap_open),
                              int(gap_ext), row_begin, row_end, 1 if symmetric else 0, kind, out.data_ptr(),
                              out.stride(0), None if score is None else score.data_ptr(),
                              0 if score is None else score.stride(0), _stream())
    return out


def symmetrize(mat, n, kind=DA_OUT_F64):
    _call(_capi.load().da_dev_symmetrize, mat.data_ptr(), n, mat.stride(0), kind, _stream())
    return mat


def widen(compact, is_nw, n_hash=0, out=None):
    """uint16 block (int16 tensor) -> float64 with the reference's divide."""
    if out is None:
        out = torch.empty(compact.shape, dtype=torch.float64, device=compact.device)
    assert compact.is_contiguous() and out.is_contiguous()
    _call(_capi.load().da_dev_widen, compact.data_ptr(), out.data_ptr(), compact.numel(), 1 if is_nw else 0,
                                          int(n_hash), _stream())
    return out


def upper_histogram(compact, n, nbins):
    """uint64 histogram (int64 tensor) of the strict upper triangle of an n x n uint16 count matrix."""
    hist = torch.zeros(nbins, dtype=torch.int64, device=compact.device)
    _call(_capi.load().da_dev_upper_histogram, compact.data_ptr(), compact.stride(0), n, int(nbins),
                                                    hist.data_ptr(), _stream())
    return hist


def extract_edges(compact, n, keep, capacity, include_diagonal=True):
    """Append (i, j, value) of the upper-triangle entries whose value v has keep[v] != 0.
    Returns (i, j, v, count): int32, int32, int16 tensors of `capacity` slots and the int64 count tensor."""
    dev = compact.device
    keep_t = torch.as_tensor(np.ascontiguousarray(keep, np.uint8)).to(dev)
    ei = torch.empty(max(capacity, 1), dtype=torch.int32, device=dev)
    ej = torch.empty(max(capacity, 1), dtype=torch.int32, device=dev)
    ev = torch.empty(max(capacity, 1), dtype=torch.int16, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    _call(_capi.load().da_dev_extract_edges, compact.data_ptr(), compact.stride(0), n, keep_t.data_ptr(),
                                                  keep_t.numel(), 1 if include_diagonal else 0, ei.data_ptr(),
                                                  ej.data_ptr(), ev.data_ptr(), int(capacity), cnt.data_ptr(), _stream())
    return ei, ej, ev, cnt


def unique_table(planes, unique, n_hash):
    """symmetric uint16 count table of the plan's unique strings (K2 on `unique` rows) with 16-byte aligned rows -- what
    unique_rows / expand_unique read in 16-byte units"""
    ld = -(-int(unique) // 8) * 8
    buf = torch.empty((int(unique), ld), dtype=torch.int16, device=planes.device)
    return mh_compare(planes, int(unique), n_hash, 0, int(unique), True, DA_OUT_COMPACT, out=buf)


def unique_rows(table, plan, out=None):
    """the n x n uint16 matrix without its duplicate rows (da_dev_unique_rows): [unique][ceil8(n)] int16; row i of the full matrix
    = row uidx[i] of it for the columns j >= i"""
    lib = _capi.load()
    ld = -(-plan.n // 8) * 8
    if out is None:
        out = torch.empty((plan.unique, ld), dtype=torch.int16, device=table.device)
    _call(lib.da_dev_unique_rows, table.data_ptr(), table.stride(0), plan.ptr(), out.data_ptr(), _stream())
    return out


def upper_histogram_rows(rows, plan, nbins):
    """upper_histogram of the full matrix read through the plan's row map"""
    hist = torch.zeros(nbins, dtype=torch.int64, device=rows.device)
    _call(_capi.load().da_dev_upper_histogram_rows, rows.data_ptr(), rows.stride(0), plan.c.d_uidx, plan.n, int(nbins),
                                                         hist.data_ptr(), _stream())
    return hist


def extract_edges_rows(rows, plan, keep, capacity, include_diagonal=True):
    """extract_edges of the full matrix read through the plan's row map"""
    dev = rows.device
    keep_t = torch.as_tensor(np.ascontiguousarray(keep, np.uint8)).to(dev)
    ei = torch.empty(max(capacity, 1), dtype=torch.int32, device=dev)
    ej = torch.empty(max(capacity, 1), dtype=torch.int32, device=dev)
    ev = torch.empty(max(capacity, 1), dtype=torch.int16, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    _call(_capi.load().da_dev_extract_edges_rows, rows.data_ptr(), rows.stride(0), plan.c.d_uidx, plan.n, keep_t.data_ptr(),
                                                       keep_t.numel(), 1 if include_diagonal else 0, ei.data_ptr(), ej.data_ptr(),
                                                       ev.data_ptr(), int(capacity), cnt.data_ptr(), _stream())
    return ei, ej, ev, cnt


def edges_to_csr(ei, ej, ev, n_edges, n):
    """(i <= j, code) device edge list -> symmetric CSR on the device (da_dev_edges_to_csr): returns (ptr int64[n+1], adj int32[nnz],
    codes int16[nnz] (uint16 bit pattern), loops int16[n] (0xFFFF = no self-loop)) as device tensors; nnz = ptr[n]"""
    lib = _capi.load()
    dev = ei.device
    m, n = int(n_edges), int(n)
    nbytes = int(lib.da_dev_edges_to_csr_bytes(m, n))
    work = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    adj = torch.empty(max(2 * m, 1), dtype=torch.int32, device=dev)
    codes = torch.empty(max(2 * m, 1), dtype=torch.int16, device=dev)
    loops = torch.empty(max(n, 1), dtype=torch.int16, device=dev)
    _call(lib.da_dev_edges_to_csr, ei.data_ptr(), ej.data_ptr(), ev.data_ptr(), m, n, work.data_ptr(), nbytes, ptr.data_ptr(),
                                        adj.data_ptr(), codes.data_ptr(), loops.data_ptr(), _stream())
    nnz = int(ptr[n].item())
    return ptr, adj[:nnz], codes[:nnz], loops[:n]


def similarity_mh(ds, k, n_hash, seeds, out=None):
    """similarityMH (src/minHash.cpp:119-188) on a device-resident set, one C call: K1 + K1b + K2, with byte-identical
    sequences collapsed first when that pays (da_dev_similarity_mh).  Returns the (n, n) float64 tensor."""
    lib = _capi.load()
    if not torch.is_tensor(seeds):
        seeds = torch.from_numpy(np.ascontiguousarray(seeds, np.uint32).view(np.int32).copy()).to(ds.residues.device)
    _require_cuda(ds.residues, "residues")
    n = ds.n
    if out is None:
        out = torch.empty((max(n, 1), max(n, 1)), dtype=torch.float64, device=ds.residues.device)
    _call(lib.da_dev_similarity_mh, ds.residues.data_ptr(), ds.offsets.data_ptr(), n, ds.total, int(k), int(n_hash),
                                         seeds.data_ptr(), out.data_ptr(), out.stride(0), _stream())
    return out


def mh_last_route():
    """what this thread's last similarity_mh call did: dict(n, unique, dedup, plane_bits, plan_ms, codes_ms, k2_ms, gather_ms, expand_ms, border_ms)"""
    n, u, t, b = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
    ms = (ctypes.c_double * 6)()
    _capi.check(_capi.load().da_mh_last_route(ctypes.addressof(n), ctypes.addressof(u), ctypes.addressof(t), ctypes.addressof(b),
                                              ctypes.addressof(ms)))
    # route 1 = duplicate-collapsing, 2 = sparse (signatures rarely agree: matching incidences bucketed per tile; then `unique` carries their
    # number, k2_ms the bucket phase and expand_ms the tile pass), 0 = the direct kernels
    # 4 = route 1 with the ROW expansion (k_expand_stream: no gathered copy; gather_ms = the copy lists, expand_ms = that kernel, border_ms = 0)
    # 3 = route 1 in its pipelined form (table compared band by band on a side stream while finished row bands are expanded: k2_ms is the
    # compare's span, gather_ms / expand_ms are sums over the chunk launches, all three overlap)
    sparse = t.value == 2
    ch, el = ctypes.c_int(0), ctypes.c_int(0)
    _capi.check(_capi.load().da_mh_last_route_chunks(ctypes.addressof(ch), ctypes.addressof(el)))
    rare, pb0 = ctypes.c_int64(0), ctypes.c_int(0)
    _capi.check(_capi.load().da_mh_last_route_split(ctypes.addressof(rare), ctypes.addressof(pb0)))
    # heavy / rare split of the column dictionaries: 8 dense planes + `rare_pairs` incidences added from lists (plane_bits_without: what it replaced)
    return {"split": rare.value >= 0, "rare_pairs": max(rare.value, 0), "plane_bits_without": pb0.value,
            "chunks": ch.value, "expand_launches": el.value, "n": n.value, "unique": n.value if sparse else u.value, "dedup": t.value in (1, 3, 4, 5), "pipelined": t.value in (3, 5),
            "expansion": {1: "tiles", 3: "tiles, pipelined", 4: "rows", 5: "rows, pipelined"}.get(t.value, ""), "sparse": sparse,
            "sparse_pairs": u.value if sparse else 0, "plane_bits": b.value, "plan_ms": ms[0], "codes_ms": ms[1],
            "k2_ms": ms[2], "gather_ms": ms[3], "expand_ms": ms[4], "border_ms": ms[5]}


class UniquePlan:
    """da_dev_unique_plan: byte-identical strings collapsed.  Keeps the workspace tensor the plan's device pointers point into."""

    def __init__(self, bytes_t, offsets_t, n, total):
        lib = _capi.load()
        _require_cuda(bytes_t, "sequence bytes")
        nbytes = int(lib.da_dev_unique_plan_bytes(int(n), int(total)))
        self.work = torch.empty(nbytes, dtype=torch.uint8, device=bytes_t.device)
        self.c = _capi.DaUniquePlan()
        self.c.struct_size = ctypes.sizeof(_capi.DaUniquePlan)
        _call(lib.da_dev_unique_plan, bytes_t.data_ptr(), offsets_t.data_ptr(), int(n), int(total), self.work.data_ptr(), nbytes,
                                           ctypes.addressof(self.c), _stream())
        self.n, self.unique = int(self.c.n), int(self.c.unique)

    def ptr(self):
        return ctypes.addressof(self.c)


def shards_to_table(gathered, ld_g, n, world, value_bits, out=None):
    """gathered MinHash shards (uint16 blocks: value_bits = 0; packed: their bit count) -> symmetric uint16 table [n][ld]"""
    ld = -(-int(n) // 8) * 8
    if out is None:
        out = torch.empty((int(n), ld), dtype=torch.int16, device=gathered.device)
    _call(_capi.load().da_dev_shards_to_table, gathered.data_ptr(), int(ld_g), int(n), int(world), int(value_bits), out.data_ptr(),
                                                    out.stride(0), _stream())
    return out


def nw_unique_rows(plan, max_len, matrix_name, gap_open, gap_ext, rank, world, out_rows):
    """rank `rank` of `world`'s cyclic 128-row units of the ordered unique NW table into out_rows (int16 [Q * 128][ld >= unique])"""
    lib = _capi.load()
    mid = lib.da_matrix_id(matrix_name.encode("latin-1"))
    if mid < 0:
        _capi.check(_capi.DA_ERR_BAD_MATRIX)
    _call(lib.da_dev_nw_unique_rows, plan.ptr(), int(max_len), mid, int(gap_open), int(gap_ext), int(rank), int(world),
                                          out_rows.data_ptr(), out_rows.stride(0), _stream())


def expand_workspace_bytes(n, unique, is_nw, n_hash=0, nw_max_len=0):
    return int(_capi.load().da_dev_expand_workspace_bytes(int(n), int(unique), 1 if is_nw else 0, int(n_hash), int(nw_max_len)))


def expand_unique(table, plan, is_nw, n_hash, nw_max_len, out, table_world=1, work=None):
    """dense float64 n x n from the table of the unique strings (da_dev_expand_unique)"""
    if work is None:
        work = torch.empty(expand_workspace_bytes(plan.n, plan.unique, is_nw, n_hash, nw_max_len), dtype=torch.uint8, device=table.device)
    _call(_capi.load().da_dev_expand_unique, table.data_ptr(), table.stride(0), int(table_world), plan.ptr(), 1 if is_nw else 0,
                                                  int(n_hash), int(nw_max_len), work.data_ptr(), work.numel(), out.data_ptr(), out.stride(0),
                                                  _stream())
    return out


def nw_last_route():
    """what this thread's last whole-matrix NW call did: dict(n, unique, dedup, plan_ms, dp_ms, expand_ms)"""
    n, u, t = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int(0)
    ms = (ctypes.c_double * 3)()
    _capi.check(_capi.load().da_nw_last_route(ctypes.addressof(n), ctypes.addressof(u), ctypes.addressof(t), ctypes.addressof(ms)))
    return {"n": n.value, "unique": u.value, "dedup": bool(t.value), "plan_ms": ms[0], "dp_ms": ms[1], "expand_ms": ms[2]}
