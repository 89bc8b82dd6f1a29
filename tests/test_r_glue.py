"""A compiler pass -- and a run -- for r_glue/src/similarity_glue.cpp in an image without R.

The glue is the binding a maintainer adds to the reference package; R and Rcpp are absent here, so it is
compiled against tests/rcpp_stub/Rcpp.h, a TEST-ONLY header that declares exactly the Rcpp surface the glue touches, and
driven by tests/rcpp_stub/glue_driver.cpp.  What this guards: typos, argument order / types against include/dynaalign.h,
the glue's own logic (packing, seed option, dimnames, 1-based edges, error mapping).  What it is NOT: parity evidence --
the stub is not Rcpp and says nothing about the reference (matches src/RcppExports.cpp:15-39 only in shape)."""
import os
import struct
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
STUB = os.path.join(HERE, "rcpp_stub")
GLUE = os.path.join(ROOT, "r_glue", "src", "similarity_glue.cpp")
LIBDIR = os.path.join(ROOT, "dynaalign_amd", "lib")
DRIVER = os.path.join(STUB, "build", "glue_driver")


def test_glue_passes_the_compiler_against_the_c_abi():
    """g++ -fsyntax-only with warnings as errors: every da_* call in the glue matches include/dynaalign.h"""
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I" + STUB,
                           "-I" + os.path.join(ROOT, "include"), GLUE])


@pytest.fixture(scope="module")
def driver(built):
    os.makedirs(os.path.dirname(DRIVER), exist_ok=True)
    srcs = [GLUE, os.path.join(STUB, "glue_driver.cpp")]
    deps = srcs + [os.path.join(STUB, "Rcpp.h"), os.path.join(ROOT, "include", "dynaalign.h")]
    if not os.path.exists(DRIVER) or any(os.path.getmtime(d) > os.path.getmtime(DRIVER) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + STUB, "-I" + os.path.join(ROOT, "include")] + srcs +
                              ["-o", DRIVER, "-L" + LIBDIR, "-ldynaalign_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])
    return DRIVER


def run(driver, args, seqs, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([driver] + [str(a) for a in args], input=("".join(s + "\n" for s in seqs)).encode("latin-1"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=600)
    return p.returncode, p.stdout, p.stderr.decode("latin-1").strip()


def test_reference_error_texts_surface_as_rcpp_stop(driver):
    """validation happens before any device work, so the reference's messages (src/minHash.cpp:121-131,
    src/pairwiseSeqAlign.cpp:204) come through the glue's check() -> Rcpp::stop on a box without a GPU too"""
    assert run(driver, ["mh", 4, 50], []) == (3, b"", "ERROR: Input sequences vector cannot be empty")
    assert run(driver, ["mh", 0, 50], ["AAAA"]) == (3, b"", "ERROR: 'k' must be a positive integer")
    assert run(driver, ["mh", 4, 0], ["AAAA"]) == (3, b"", "ERROR: Number of hash functions must be positive")
    assert run(driver, ["mh", -1, -1], []) == (3, b"", "ERROR: Input sequences vector cannot be empty")     # order of the checks
    assert run(driver, ["nw", "PAM250", 10, 4], ["AAAA"]) == (3, b"", "ERROR: Invalid substitution matrix name: PAM250")
    rc, out, err = run(driver, ["mh", 4, 50], ["ACDEF"], {"GLUE_OPTION_EXCHANGE": "ring"})
    assert rc == 3 and "DynaAlign.exchange" in err


def test_glue_fails_loudly_without_a_gpu(driver):
    from dynaalign_amd import _capi
    if _capi.load().da_device_count() > 0:
        pytest.skip("a GPU is present")
    rc, out, err = run(driver, ["mh", 4, 50], ["ACDEFGHIKL", "ACDEFGHIKM"])
    assert rc == 3 and out == b"" and "no usable HIP device" in err


def _matrix(blob):
    n = struct.unpack_from("<q", blob)[0]
    m = np.frombuffer(blob, np.float64, n * n, 8).reshape(n, n)
    ok = struct.unpack_from("<q", blob, 8 + 8 * n * n)[0]
    assert len(blob) == 16 + 8 * n * n
    return m, ok


def _edges(blob):
    thr = struct.unpack_from("<d", blob)[0]
    m = struct.unpack_from("<q", blob, 8)[0]
    i = np.frombuffer(blob, np.int32, m, 16)
    j = np.frombuffer(blob, np.int32, m, 16 + 4 * m)
    w = np.frombuffer(blob, np.float64, m, 16 + 8 * m)
    assert len(blob) == 16 + 16 * m
    return thr, i, j, w


@pytest.mark.gpu
def test_glue_results_equal_the_ctypes_path(driver):
    """the glue's four exports, run: same bits as dynaalign_amd (which the parity tests compare with the oracle)"""
    import dynaalign_amd as da
    from dynaalign_amd import synth
    seqs = synth.to_strings(*synth.h3n2_like(700, 20))
    seqs[3] = seqs[2]
    rc, out, err = run(driver, ["mh", 4, 200], seqs, {"GLUE_OPTION_SEED": "12345"})
    assert rc == 0, err
    m, dimnames_ok = _matrix(out)
    assert dimnames_ok == 1
    want = np.asarray(da.similarityMH(seqs, 4, 200, seed=12345))
    assert np.array_equal(m.view(np.uint64), want.view(np.uint64))
    # seed reduced modulo 2^32 like the glue documents; DYNAALIGN_SEED is the environment form
    rc, out2, err = run(driver, ["mh", 4, 200], seqs, {"GLUE_OPTION_SEED": str(12345 + 2 ** 32)})
    assert rc == 0 and out2 == out
    rc, out3, err = run(driver, ["mh", 4, 200], seqs, {"DYNAALIGN_SEED": "12345"})
    assert rc == 0 and out3 == out
    # options(DynaAlign.devices = c(0, 0)): the multi-device entry point, row blocks, same matrix
    rc, out4, err = run(driver, ["mh", 4, 200], seqs, {"GLUE_OPTION_SEED": "12345", "GLUE_OPTION_DEVICES": "0,0"})
    assert rc == 0 and out4 == out, err
    rc, out, err = run(driver, ["nw", "BLOSUM80", 7, 2], seqs)
    assert rc == 0, err
    m, dimnames_ok = _matrix(out)
    assert dimnames_ok == 1
    want = np.asarray(da.similarityNW(seqs, "BLOSUM80", 7, 2))
    assert np.array_equal(m.view(np.uint64), want.view(np.uint64))
    rc, out, err = run(driver, ["nw", "BLOSUM62", 10, 4], seqs[:5] + ["ACDJ"])
    assert (rc, out) == (3, b"") and err == "ERROR: Invalid amino acid in sequence2: J"
    rc, out, err = run(driver, ["mh_edges", 4, 200, 0.8], seqs, {"GLUE_OPTION_SEED": "12345"})
    assert rc == 0, err
    thr, i, j, w = _edges(out)
    thr_w, i_w, j_w, w_w = da.similarityMH_edges(seqs, 4, 200, 0.8, seed=12345)
    assert thr == thr_w and np.array_equal(i, np.asarray(i_w) + 1) and np.array_equal(j, np.asarray(j_w) + 1)    # R is 1-based
    assert np.array_equal(w.view(np.uint64), np.asarray(w_w, np.float64).view(np.uint64))
    rc, out, err = run(driver, ["nw_edges", "BLOSUM62", 10, 4, 0.9], seqs)
    assert rc == 0, err
    thr, i, j, w = _edges(out)
    thr_w, i_w, j_w, w_w = da.similarityNW_edges(seqs, "BLOSUM62", 10, 4, 0.9)
    assert thr == thr_w and np.array_equal(i, np.asarray(i_w) + 1) and np.array_equal(j, np.asarray(j_w) + 1)
    assert np.array_equal(w.view(np.uint64), np.asarray(w_w, np.float64).view(np.uint64))
