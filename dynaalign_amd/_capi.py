"""ctypes binding of libdynaalign_hip.so -- the C ABI declared in include/dynaalign.h.

This module is the Python-side equivalent of the Rcpp glue in r_glue/: it only
marshals arguments.  All compute happens in the HIP library; if the library (or
a GPU) is missing, calls fail loudly -- there is no CPU fallback.
"""
import ctypes as C
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libdynaalign_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "dynaalign.h")

DA_OK = 0
DA_ERR_EMPTY_INPUT, DA_ERR_BAD_K, DA_ERR_BAD_NHASH, DA_ERR_BAD_MATRIX = 1, 2, 3, 4
DA_ERR_BAD_RESIDUE_SEQ1, DA_ERR_BAD_RESIDUE_SEQ2, DA_ERR_NOMEM, DA_ERR_NO_DEVICE = 5, 6, 7, 8
DA_ERR_HIP, DA_ERR_UNSUPPORTED, DA_ERR_BAD_ARG = 9, 10, 11
DA_OUT_F64, DA_OUT_COMPACT, DA_OUT_PACK32 = 0, 1, 2

_vp, _i64, _i32, _u32, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_uint32, C.c_size_t

# name -> (restype, argtypes).  Must list every symbol include/dynaalign.h declares
# (tests/test_abi.py checks the two against each other).
SIGNATURES = {
    "da_last_error": (C.c_char_p, []),
    "da_status_message": (C.c_char_p, [_i32]),
    "da_abi_version": (_i32, []),
    "da_device_count": (_i32, []),
    "da_release_device_memory": (_sz, []),
    "da_config_reload": (None, []),
    "da_debug_comm_cache_state": (C.c_int, [C.c_int, C.c_void_p]),
    "da_hash_family_seeds": (_i32, [_u32, _i32, _vp]),
    "da_random_seed": (_u32, []),
    "da_similarity_mh": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "da_similarity_mh_opts": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "da_similarity_nw_opts": (_i32, [_vp, _vp, _i64, C.c_char_p, _i32, _i32, _vp, _vp]),
    "da_rccl_available": (_i32, []),
    "da_minhash_signatures": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "da_mh_counts": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _i64, _i64, _vp]),
    "da_similarity_nw": (_i32, [_vp, _vp, _i64, C.c_char_p, _i32, _i32, _vp]),
    "da_nw_pairs": (_i32, [_vp, _vp, _i64, C.c_char_p, _i32, _i32, _i64, _i64, _vp, _vp, _vp]),
    "da_sig_ld": (_i64, [_i32]),
    "da_dev_minhash_signatures": (_i32, [_vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _vp, _i64, _vp]),
    "da_mh_planes_words": (_i64, [_i64, _i32]),
    "da_mh_planes_workspace_bytes": (_sz, [_i64, _i32]),
    "da_dev_mh_planes": (_i32, [_vp, _i64, _i64, _i32, _i32, _vp, _sz, _vp, _i64, _vp, _vp]),
    "da_dev_mh_compare": (_i32, [_vp, _i32, _i64, _i32, _i64, _i64, _i32, _i32, _vp, _i64, _vp]),
    "da_nw_last_route": (_i32, [_vp, _vp, _vp, _vp]),
    "da_dev_similarity_mh": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _i64, _vp]),
    "da_mh_last_route": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "da_mh_last_route_chunks": (_i32, [_vp, _vp]),
    "da_mh_last_route_split": (_i32, [_vp, _vp]),
    "da_dev_unique_plan_bytes": (_sz, [_i64, _i64]),
    "da_dev_unique_plan": (_i32, [_vp, _vp, _i64, _i64, _vp, _sz, _vp, _vp]),
    "da_dev_shards_to_table": (_i32, [_vp, _i64, _i64, _i32, _i32, _vp, _i64, _vp]),
    "da_dev_nw_unique_rows": (_i32, [_vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp]),
    "da_dev_expand_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32, _i32]),
    "da_dev_unique_rows_bytes": (_sz, [_i64, _i64]),
    "da_dev_unique_rows": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "da_dev_upper_histogram_rows": (_i32, [_vp, _i64, _vp, _i64, _i32, _vp, _vp]),
    "da_dev_extract_edges_rows": (_i32, [_vp, _i64, _vp, _i64, _vp, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _vp]),
    "da_dev_expand_unique": (_i32, [_vp, _i64, _i32, _vp, _i32, _i32, _i32, _vp, _sz, _vp, _i64, _vp]),
    "da_dev_nw_encode": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "da_dev_nw": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _i32, _i64, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp]),
    "da_similarity_mh_edges": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, C.c_double, _vp, _vp, _i64, _vp, _vp, _vp]),
    "da_similarity_nw_edges": (_i32, [_vp, _vp, _i64, C.c_char_p, _i32, _i32, C.c_double, _vp, _vp, _i64, _vp, _vp, _vp]),
    "da_similarity_mh_edges_begin": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, C.c_double, _vp, _vp, _vp]),
    "da_similarity_nw_edges_begin": (_i32, [_vp, _vp, _i64, C.c_char_p, _i32, _i32, C.c_double, _vp, _vp, _vp]),
    "da_edges_fetch": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "da_edges_free": (None, [_vp]),
    "da_quantile_type7": (_i32, [_vp, _vp, _i32, C.c_double, _vp]),
    "da_dev_upper_histogram": (_i32, [_vp, _i64, _i64, _i32, _vp, _vp]),
    "da_dev_extract_edges": (_i32, [_vp, _i64, _i64, _vp, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _vp]),
    "da_dev_shard_histogram": (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp]),
    "da_dev_shard_extract_edges": (_i32, [_vp, _i64, _i64, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _vp]),
    "da_louvain": (_i32, [_i64, _i64, _vp, _vp, _vp, C.c_double, _u32, _vp, _vp, _vp]),
    "da_louvain_csr": (_i32, [_i64, _vp, _vp, _vp, _vp, _vp, _i32, C.c_double, _u32, _vp, _vp, _vp]),
    "da_dev_edges_to_csr_bytes": (_sz, [_i64, _i64]),
    "da_dev_edges_to_csr": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "da_matrix_id": (_i32, [C.c_char_p]),
    "da_shard_rows": (_i64, [_i64, _i32, _i32]),
    "da_shard_ld": (_i64, [_i64, _i32, _i32]),
    "da_shard_packed_bytes": (_i64, [_i64, _i32, _i32]),
    "da_dev_pack_shard": (_i32, [_vp, _i64, _i64, _i32, _i32, _vp, _vp]),
    "da_dev_finalize_shards_packed": (_i32, [_vp, _i64, _i32, _i32, _i32, _vp, _i64, _vp]),
    "da_dev_mh_compare_shard": (_i32, [_vp, _i32, _i64, _i32, _i32, _i32, _vp, _i64, _vp]),
    "da_dev_nw_shard": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp]),
    "da_dev_finalize_shards": (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _i64, _vp]),
    "da_dev_symmetrize": (_i32, [_vp, _i64, _i64, _i32, _vp]),
    "da_dev_widen": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp]),
}

DA_EXCHANGE = {"rows": 0, "allgather": 1, "peercopy": 2}
DA_PHASES = ("setup", "compute", "exchange", "finalize", "d2h", "total")


class DaUniquePlan(C.Structure):
    """struct da_unique_plan (include/dynaalign.h): device pointers into the plan's workspace"""
    _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_int32), ("n", C.c_int64), ("unique", C.c_int64),
                ("d_uidx", C.c_void_p), ("d_ufirst", C.c_void_p), ("d_ulast", C.c_void_p), ("d_ubytes", C.c_void_p),
                ("d_uoffsets", C.c_void_p), ("d_minfirst", C.c_void_p), ("d_maxlast", C.c_void_p)]


class DaOpts(C.Structure):
    """struct da_opts (include/dynaalign.h)"""
    _fields_ = [("struct_size", C.c_uint32), ("n_devices", C.c_int32), ("devices", C.POINTER(C.c_int32)),
                ("exchange", C.c_int32), ("reserved", C.c_uint32), ("phase_ms", C.POINTER(C.c_double))]


def make_opts(devices=None, exchange="rows"):
    """-> (DaOpts, keepalive): the struct for da_similarity_*_opts; keepalive[1] receives the phase times (ms)"""
    devs = [] if devices is None else [int(d) for d in devices]
    if exchange not in DA_EXCHANGE:
        raise ValueError("exchange must be one of %s" % sorted(DA_EXCHANGE))
    dev_arr = (C.c_int32 * max(len(devs), 1))(*devs)
    phases = (C.c_double * len(DA_PHASES))()
    o = DaOpts(C.sizeof(DaOpts), len(devs), C.cast(dev_arr, C.POINTER(C.c_int32)), DA_EXCHANGE[exchange], 0,
               C.cast(phases, C.POINTER(C.c_double)))
    return o, (dev_arr, phases)


_lib = None


class DynaAlignError(RuntimeError):
    """Raised for any non-zero status; .code is the da_status, str() the library's message
    (for the reference's own error conditions: the reference's message text)."""

    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


def header_symbols(path=HEADER_PATH):
    """Function names declared in include/dynaalign.h."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(da_[a-z0-9_]+)\s*\(", txt)))


def load(path=None):
    """dlopen the HIP library (once) and set prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("DYNAALIGN_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise DynaAlignError(
            DA_ERR_NO_DEVICE,
            "libdynaalign_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(dynaalign_amd has no CPU fallback)" % path)
    try:  # if torch is around, let its bundled libamdhip64.so.7 be the one HIP runtime in the process
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = _Library(lib)
    return _lib


def _switches():
    return tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("DYNAALIGN_")))


class _Library:
    """The loaded library.  The C side parses its DYNAALIGN_* switches once (da_common.hpp Config); tests, bench.py and the tools flip
    switches inside one process, so this front end calls the library's reload hook whenever the process's DYNAALIGN_* environment
    differs from what it was at the previous call -- the library itself never looks at the environment again."""

    def __init__(self, cdll):
        object.__setattr__(self, "_cdll", cdll)
        object.__setattr__(self, "_seen", _switches())

    def __getattr__(self, name):
        fn = getattr(self._cdll, name)
        if name != "da_config_reload":
            now = _switches()
            if now != self._seen:
                self._cdll.da_config_reload()
                object.__setattr__(self, "_seen", now)
        return fn


def check(rc):
    if rc != DA_OK:
        raise DynaAlignError(rc, load().da_last_error().decode("latin-1"))


def ptr(a):
    """numpy array -> void* (None passes NULL)"""
    return None if a is None else a.ctypes.data
