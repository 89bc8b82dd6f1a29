"""CPU tests: the oracle against every known answer we hold (SURVEY.md A.3 values recorded
from the reference, published MurmurHash3 / mt19937 vectors), against an independent Python
model, and against the committed oracle_vectors.npz (drift detector)."""
import json
import math
import os

import numpy as np
import pytest

import oracle_lib as O
from nw_model import nw_identity

HERE = os.path.dirname(os.path.abspath(__file__))


def test_murmur3_known_answers(kats):
    for kat in kats["murmur3"]:
        assert O.murmur3(kat["key"], kat["seed"]) == kat["hash"], kat


def test_mt19937_known_answers(kats):
    for kat in kats["mt19937"]:
        if "first" in kat:
            assert O.seeds(kat["seed"], len(kat["first"])).tolist() == kat["first"]
        else:
            assert int(O.seeds(kat["seed"], kat["index"] + 1)[kat["index"]]) == kat["value"]


def test_num_kmers(kats):
    for kat in kats["num_kmers"]:
        assert O.lib().orc_num_kmers(kat["len"], kat["k"]) == kat["count"]


def test_signature_known_answers(kats):
    for kat in kats["signatures"]:
        sig = O.signatures([kat["sequence"]], kat["k"], kat["n_hash"], O.seeds(kat["seed"], kat["n_hash"]))
        assert sig[0].tolist() == kat["sig"]


def test_nw_4x4_known_answer(kats):
    q = kats["nw_4x4"]
    rc, M, _ = O.similarity_nw(q["sequences"], q["matrix"], q["gap_open"], q["gap_ext"])
    assert rc == 0
    assert np.round(M, 6).tolist() == q["rounded6"]
    for key, (mt, ln) in q["exact_fractions"].items():
        i, j = (int(v) for v in key.split(","))
        assert M[i, j] == mt / ln and M[j, i] == mt / ln


def test_nw_order_asymmetry(kats):
    q = kats["nw_asymmetric"]
    rc, mt, ln, _, _ = O.nw_pair(q["a"], q["b"])
    assert (rc, mt, ln) == (0, *q["ab"])
    rc, mt, ln, _, _ = O.nw_pair(q["b"], q["a"])
    assert (rc, mt, ln) == (0, *q["ba"])
    # the driver always evaluates calc(seq[min], seq[max]) and mirrors
    rc, M, _ = O.similarity_nw([q["b"], q["a"]])
    assert M[0, 1] == M[1, 0] == q["ba"][0] / q["ba"][1]


def test_nw_edges(kats):
    rc, M, _ = O.similarity_nw(["", "A"])
    assert rc == 0 and math.isnan(M[0, 0]) and M[0, 1] == 0.0 and M[1, 1] == 1.0
    rc, mt, ln, sc, _ = O.nw_pair("A", "")
    assert (mt, ln, sc) == (0, 1, -(2 ** 30))
    rc, M, msg = O.similarity_nw(["AJ", "AA"])
    assert rc == O.ERR_BAD_RES2 or rc == O.ERR_BAD_RES1
    # pair (0,0) is visited first: sequence1[0]='A' is fine, the scan of sequence2 meets 'J'
    assert msg == "Invalid amino acid in sequence2: J"
    rc, _, _, _, bad = O.nw_pair("AJ", "AA")
    assert rc == O.ERR_BAD_RES1 and bad == "J"  # calc("AJ","AA")
    rc, _, msg = O.similarity_nw(["AA"], "PAM250")
    assert rc == O.ERR_BAD_MATRIX and msg == kats["nw_bad_matrix"]["error"]
    # an empty sequence1 validates nothing (reference loops do not run)
    rc, mt, ln, _, _ = O.nw_pair("", "J")
    assert (rc, mt, ln) == (0, 0, 1)


def test_nw_evp_checksum(kats, evp):
    q = kats["nw_evp"]
    assert len(evp) == q["n"]
    rc, W, _ = O.similarity_nw(evp)
    assert rc == 0
    assert abs(W[0, 1] - q["w01"]) < 1e-10 and abs(W[0, 2] - q["w02"]) < 1e-10
    assert abs(W.sum() - q["sum_all"]) < 1e-6
    assert np.array_equal(W, W.T) and np.all(np.diag(W) == 1.0)


def test_mh_validation_order(kats):
    s = np.zeros(1, np.uint32)
    assert O.similarity_mh([], 0, 0, s)[0] == 1
    assert O.similarity_mh(["AAAA"], 0, 0, s)[0] == 2
    assert O.similarity_mh(["AAAA"], 4, 0, s)[0] == 3


def test_mh_semantics():
    seeds = O.seeds(42, 64)
    seqs = ["ACDEFGHIKL", "ACDEFGHIKL", "ACD", "AC", "WWWWWWWWWW", ""]
    rc, M = O.similarity_mh(seqs, 4, 64, seeds)
    assert rc == 0
    assert M[0, 1] == 1.0                       # duplicates
    assert M[2, 3] == 1.0 and M[3, 5] == 1.0    # no k-mers at all: all-UINT32_MAX signatures agree
    assert M[0, 2] == 0.0
    assert np.all(np.diag(M) == 1.0) and np.array_equal(M, M.T)
    sig = O.signatures(seqs, 4, 64, seeds)
    assert np.all(sig[2] == 0xFFFFFFFF)
    # repeated k-mers do not change the min
    assert np.array_equal(O.signatures(["ACDEACDE"], 4, 64, seeds)[0],
                          O.signatures(["ACDEACDEACDE"], 4, 64, seeds)[0])


def test_blosum_tables_match_fixture():
    meta = json.load(open(os.path.join(HERE, "golden", "blosum_tables.json")))
    L = O.lib()
    import ctypes as C
    for name, info in meta["tables"].items():
        mid = L.orc_matrix_id(name.encode())
        assert mid >= 0
        tab = np.ctypeslib.as_array(C.cast(L.orc_matrix_table(mid), C.POINTER(C.c_int8)), (576,))
        assert tab.tolist() == info["values"]
        t = tab.reshape(24, 24)
        assert np.array_equal(t, t.T)           # all six verified symmetric
        assert (t.min(), t.max()) == (info["min"], info["max"])
    for i, c in enumerate(meta["order"]):
        assert L.orc_aa_index(ord(c)) == i
    for c in "JUOabc -1":
        assert L.orc_aa_index(ord(c)) == -1


def test_oracle_vs_independent_python_model():
    """full-matrix + traceback (C oracle) == forward-propagated matches/len (Python model)."""
    rng = np.random.RandomState(3)
    meta = json.load(open(os.path.join(HERE, "golden", "blosum_tables.json")))["tables"]
    aa = "ARNDCQEGHILKMFPSTWYVBZX*"
    names = list(meta)
    for it in range(400):
        name = names[it % len(names)]
        go, ge = (int(rng.randint(0, 15)), int(rng.randint(0, 8))) if it % 3 else (10, 4)
        alpha = aa if it % 2 else "AGW"
        a = "".join(alpha[i] for i in rng.randint(0, len(alpha), rng.randint(0, 25)))
        b = "".join(alpha[i] for i in rng.randint(0, len(alpha), rng.randint(0, 25)))
        rc, mt, ln, sc, _ = O.nw_pair(a, b, name, go, ge)
        assert rc == 0
        assert (mt, ln, sc) == nw_identity(a, b, meta[name]["values"], go, ge), (a, b, name, go, ge)


def test_committed_vectors_still_reproduce(golden):
    """oracle drift detector"""
    mix = [str(s) for s in golden["mh_sequences"]]
    for k, n_hash in ((1, 8), (4, 500), (7, 50)):
        sig = O.signatures(mix, k, n_hash, O.seeds(12345, n_hash))
        assert np.array_equal(sig, golden["mh_sig_k%d_h%d" % (k, n_hash)])
        assert np.array_equal(O.mh_counts(sig), golden["mh_cnt_k%d_h%d" % (k, n_hash)])
    nws = [str(s) for s in golden["nw_sequences"]]
    rc, mt, ln, sc, _ = O.nw_rows(nws, 0, None, "BLOSUM50", 12, 2)
    assert rc == 0
    assert np.array_equal(mt, golden["nw_BLOSUM50_12_2_matches"])
    assert np.array_equal(ln, golden["nw_BLOSUM50_12_2_len"])
    assert np.array_equal(sc, golden["nw_BLOSUM50_12_2_score"])


def test_rows_api_consistent_with_full_matrix():
    seqs = ["ACDEFGHIKL", "ACDEYGHIKL", "WWWWACDE", "", "MKV"]
    seeds = O.seeds(9, 32)
    sig = O.signatures(seqs, 3, 32, seeds)
    cnt = O.mh_counts(sig)
    rc, M = O.similarity_mh(seqs, 3, 32, seeds)
    assert np.array_equal(cnt / 32.0, M)
    assert np.array_equal(O.mh_counts(sig, 1, 3), cnt[1:3])
    rc, mt, ln, sc, _ = O.nw_rows(seqs)
    rc2, W, _ = O.similarity_nw(seqs)
    with np.errstate(invalid="ignore"):
        R = mt / ln.astype(np.float64)
    assert np.array_equal(np.isnan(R), np.isnan(W)) and np.array_equal(R[~np.isnan(R)], W[~np.isnan(W)])


def test_oracle_is_clean_under_asan_ubsan():
    """CPU sanitizer run of the oracle (GPU sanitizers are not available on the pool)"""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", O.ORACLE_DIR, "selftest_asan"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", OMP_NUM_THREADS="2")
    out = subprocess.run([os.path.join(O.ORACLE_DIR, "selftest_asan")], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "oracle selftest: ok" in out.stdout


def test_reference_structured_mh_variant_has_the_same_bits():
    """orc_similarity_mh_rowptr (row-pointer signatures, copied k-mers, column-major element stores: the CPU baseline SURVEY 8(d)
    names) == the flat port, bit for bit, on ragged input incl. empty / shorter-than-k sequences; same validation order"""
    rng = np.random.RandomState(5)
    alpha = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    seqs = ["".join(map(chr, alpha[rng.randint(0, 20, rng.randint(0, 30))])) for _ in range(257)]
    seqs[7] = seqs[6]
    for k, n_hash in ((4, 500), (2, 50), (7, 33)):
        sv = O.seeds(12345, n_hash)
        rc_a, a = O.similarity_mh(seqs, k, n_hash, sv)
        rc_b, b = O.similarity_mh(seqs, k, n_hash, sv, rowptr=True)
        assert rc_a == rc_b == 0
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    assert O.similarity_mh([], 4, 50, O.seeds(1, 50), rowptr=True)[0] == 1
    assert O.similarity_mh(["AAAA"], 0, 50, O.seeds(1, 50), rowptr=True)[0] == 2
    assert O.similarity_mh(["AAAA"], 4, 0, np.zeros(0, np.uint32), rowptr=True)[0] == 3
