"""ctypes access to the CPU oracle (oracle/liborc.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product package (dynaalign_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")

ERR_BAD_MATRIX, ERR_BAD_RES1, ERR_BAD_RES2 = 4, 5, 6


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liborc.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "liborc.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_murmur3_32.restype = C.c_uint32
        L.orc_murmur3_32.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
        L.orc_mt19937_seeds.restype = None
        L.orc_mt19937_seeds.argtypes = [C.c_uint32, C.c_int, u32p]
        L.orc_num_kmers.restype = C.c_int64
        L.orc_num_kmers.argtypes = [C.c_int64, C.c_int]
        L.orc_minhash_signatures.restype = C.c_int
        L.orc_minhash_signatures.argtypes = [u8p, i64p, C.c_int64, C.c_int, C.c_int, u32p, u32p]
        L.orc_mh_counts_rows.restype = None
        L.orc_mh_counts_rows.argtypes = [u32p, C.c_int64, C.c_int, C.c_int64, C.c_int64, u16p]
        L.orc_similarity_mh.restype = C.c_int
        L.orc_similarity_mh.argtypes = [u8p, i64p, C.c_int64, C.c_int, C.c_int, u32p, f64p]
        L.orc_similarity_mh_rowptr.restype = C.c_int
        L.orc_similarity_mh_rowptr.argtypes = [u8p, i64p, C.c_int64, C.c_int, C.c_int, u32p, f64p]
        L.orc_matrix_id.restype = C.c_int
        L.orc_matrix_id.argtypes = [C.c_char_p]
        L.orc_aa_index.restype = C.c_int
        L.orc_aa_index.argtypes = [C.c_uint8]
        L.orc_nw_pair.restype = C.c_int
        L.orc_nw_pair.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                  C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_uint8)]
        L.orc_matrix_table.restype = C.c_void_p
        L.orc_matrix_table.argtypes = [C.c_int]
        L.orc_similarity_nw.restype = C.c_int
        L.orc_similarity_nw.argtypes = [u8p, i64p, C.c_int64, C.c_char_p, C.c_int, C.c_int, f64p,
                                        C.c_char_p, C.c_size_t]
        L.orc_nw_rows.restype = C.c_int
        L.orc_nw_rows.argtypes = [u8p, i64p, C.c_int64, C.c_int64, C.c_int64, C.c_char_p, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
        L.orc_num_threads.restype = C.c_int
        _LIB = L
    return _LIB


def pack(seqs):
    """list of str/bytes -> (residues uint8[total], offsets int64[n+1])"""
    bs = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in seqs]
    off = np.zeros(len(bs) + 1, np.int64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs])
    res = np.frombuffer(b"".join(bs), np.uint8).copy() if off[-1] else np.zeros(1, np.uint8)
    return res, off


def murmur3(key, seed):
    key = key.encode("latin-1") if isinstance(key, str) else key
    return lib().orc_murmur3_32(key, len(key), seed)


def seeds(seed, n):
    out = np.zeros(max(n, 1), np.uint32)
    lib().orc_mt19937_seeds(seed, n, out)
    return out[:n]


def signatures(seqs, k, n_hash, seedvec):
    res, off = pack(seqs)
    sig = np.zeros((len(seqs), n_hash), np.uint32)
    rc = lib().orc_minhash_signatures(res, off, len(seqs), k, n_hash, np.ascontiguousarray(seedvec, np.uint32), sig)
    assert rc == 0, rc
    return sig


def mh_counts(sig, row_begin=0, row_end=None):
    n, n_hash = sig.shape
    row_end = n if row_end is None else row_end
    out = np.zeros((row_end - row_begin, n), np.uint16)
    lib().orc_mh_counts_rows(np.ascontiguousarray(sig), n, n_hash, row_begin, row_end, out)
    return out


def similarity_mh(seqs, k, n_hash, seedvec, rowptr=False):
    """returns (rc, matrix); rowptr=True: the variant with the reference's data structures (orc_similarity_mh_rowptr)"""
    res, off = pack(seqs)
    n = len(seqs)
    out = np.zeros((max(n, 1), max(n, 1)), np.float64)
    sv = np.ascontiguousarray(seedvec, np.uint32) if len(seedvec) else np.zeros(1, np.uint32)
    fn = lib().orc_similarity_mh_rowptr if rowptr else lib().orc_similarity_mh
    rc = fn(res, off, n, k, n_hash, sv, out)
    return rc, out[:n, :n]


def nw_pair(a, b, matrix="BLOSUM62", go=10, ge=4):
    """returns (rc, matches, alen, score, bad_char)"""
    L = lib()
    a = a.encode("latin-1") if isinstance(a, str) else a
    b = b.encode("latin-1") if isinstance(b, str) else b
    mid = L.orc_matrix_id(matrix.encode())
    assert mid >= 0
    nm, ln, sc, bad = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_uint8(0)
    rc = L.orc_nw_pair(a, len(a), b, len(b), L.orc_matrix_table(mid), go, ge,
                       C.byref(nm), C.byref(ln), C.byref(sc), C.byref(bad))
    return rc, nm.value, ln.value, sc.value, chr(bad.value)


def similarity_nw(seqs, matrix="BLOSUM62", go=10, ge=4):
    """returns (rc, matrix, message)"""
    res, off = pack(seqs)
    n = len(seqs)
    out = np.zeros((max(n, 1), max(n, 1)), np.float64)
    buf = C.create_string_buffer(256)
    rc = lib().orc_similarity_nw(res, off, n, matrix.encode(), go, ge, out, buf, 256)
    return rc, out[:n, :n], buf.value.decode("latin-1")


def nw_rows(seqs, row_begin=0, row_end=None, matrix="BLOSUM62", go=10, ge=4):
    """returns (rc, matches, alen, score, message); arrays [rows][n] int32"""
    res, off = pack(seqs)
    n = len(seqs)
    row_end = n if row_end is None else row_end
    r = row_end - row_begin
    nm = np.zeros((r, n), np.int32)
    ln = np.zeros((r, n), np.int32)
    sc = np.zeros((r, n), np.int32)
    buf = C.create_string_buffer(256)
    rc = lib().orc_nw_rows(res, off, n, row_begin, row_end, matrix.encode(), go, ge,
                           nm.ctypes.data, ln.ctypes.data, sc.ctypes.data, buf, 256)
    return rc, nm, ln, sc, buf.value.decode("latin-1")


def num_threads():
    return lib().orc_num_threads()
