"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs, against the committed golden fixtures, and
-- at BASELINE sizes -- through size-independent properties.

Bars: bit-exact for signatures, match counts, NW (matches, length, score); bit-exact float64
(compared as uint64) for the similarity matrices -- they are one IEEE divide of two exactly
representable integers (reference src/minHash.cpp:174, src/pairwiseSeqAlign.cpp:311)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def da(built):
    import dynaalign_amd
    from dynaalign_amd import _capi
    lib = _capi.load()
    assert lib.da_device_count() > 0, "no HIP device visible: the product has no CPU fallback"
    return dynaalign_amd


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def assert_same_f64(got, want):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape
    assert np.array_equal(bits(got), bits(want)), "float64 bit patterns differ at %s" % (
        np.argwhere(bits(got) != bits(want))[:5].tolist(),)


# ---------------------------------------------------------------- MinHash

def test_native_library_is_loaded(da):
    from dynaalign_amd import _capi
    maps = open("/proc/self/maps").read()
    assert "libdynaalign_hip.so" in maps
    assert _capi.load().da_device_count() >= 1


def test_signature_known_answers_on_gpu(da, kats):
    for kat in kats["signatures"]:
        sig = da.minhash_signatures([kat["sequence"]], kat["k"], kat["n_hash"], seed=kat["seed"])
        assert sig[0].tolist() == kat["sig"]


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 7])
@pytest.mark.parametrize("n_hash", [8, 50, 500])
def test_signatures_and_counts_match_golden(da, golden, k, n_hash):
    """edge cases: empty, L<k, L=k, repeated k-mers, non-AA bytes, duplicates"""
    seqs = [str(s) for s in golden["mh_sequences"]]
    sig = da.minhash_signatures(seqs, k, n_hash, seed=12345)
    assert np.array_equal(sig, golden["mh_sig_k%d_h%d" % (k, n_hash)])
    cnt = da.mh_counts(seqs, k, n_hash, seed=12345)
    assert np.array_equal(cnt, golden["mh_cnt_k%d_h%d" % (k, n_hash)])
    M = da.similarityMH(seqs, k, n_hash, seed=12345)
    assert_same_f64(M, golden["mh_cnt_k%d_h%d" % (k, n_hash)].astype(np.float64) / n_hash)


def test_more_hash_functions_than_16_bit_counters(da):
    """n_hash > 65535: the reference has no upper limit (src/minHash.cpp:129-131); the host entry point processes the hash
    functions in chunks of 65504 and sums the counts in 32 bits.  n = 300 has interior tiles; duplicates give count == n_hash."""
    from dynaalign_amd import synth
    n, n_hash = 300, 70000
    seqs = synth.to_strings(*synth.h3n2_like(n, 20))
    seqs[200:220] = seqs[0:20]
    seeds = da.hash_family_seeds(5, n_hash)
    rc, want = O.similarity_mh(seqs, 4, n_hash, seeds)
    assert rc == 0 and want[0, 200] == 1.0
    assert_same_f64(da.similarityMH(seqs, 4, n_hash, seed=5), want)
    assert_same_f64(da.similarityMH(seqs, 4, n_hash, seed=5, devices=[0]), want)
    with pytest.raises(da.DynaAlignError, match="one device only"):
        da.similarityMH(seqs, 4, n_hash, seed=5, devices=[0, 0])
    with pytest.raises(da.DynaAlignError, match="uint16 match counts cannot hold"):
        da.mh_counts(seqs, 4, n_hash, seed=5)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6, 9])
def test_bytes_above_0x7f_are_hashed_as_they_are(da, k):
    """the reference hashes the raw bytes of the CHARSXP: 4-byte blocks through a uint32 read, the tail through uint8
    (src/minHash.cpp:32,43) -- no sign extension, no re-encoding.  Latin-1 strings with bytes 0x80..0xFF (and control bytes)
    through the packed-byte boundary: signatures, counts and the matrix against the oracle."""
    rng = np.random.RandomState(k)
    seqs = ["".join(chr(int(b)) for b in rng.randint(1, 256, int(rng.randint(0, 30)))) for _ in range(200)]
    seqs += [chr(255) * 12, "".join(chr(c) for c in (128, 129, 130, 131, 132)), chr(255) * 12, "A" + chr(233) + chr(232) + "B" + chr(252) + "C"]
    seeds = da.hash_family_seeds(777, 100)
    want_sig = O.signatures(seqs, k, 100, seeds)
    assert np.array_equal(da.minhash_signatures(seqs, k, 100, seed=777), want_sig)
    rc, want = O.similarity_mh(seqs, k, 100, seeds)
    assert rc == 0
    assert_same_f64(da.similarityMH(seqs, k, 100, seed=777), want)


@pytest.mark.parametrize("n,k,n_hash", [(1, 4, 50), (2, 4, 500), (127, 4, 500), (128, 2, 50), (129, 4, 33),
                                        (300, 3, 7), (641, 2, 50), (1000, 4, 500), (1025, 6, 129)])
def test_similarity_mh_matches_oracle(da, evp, n, k, n_hash):
    from dynaalign_amd import synth
    if n == 641:
        seqs = evp                      # config 1: the bundled evp_peparray probes, k=2 n_hash=50
    else:
        seqs = synth.to_strings(*synth.h3n2_like(n, 20))
    seeds = da.hash_family_seeds(12345, n_hash)
    rc, want = O.similarity_mh(seqs, k, n_hash, seeds)
    assert rc == 0
    got = da.similarityMH(seqs, k, n_hash, seed=12345)
    assert_same_f64(got, want)
    assert got.dimnames[0][0] == "1" and got.dimnames[1][-1] == str(n)
    assert np.all(np.diag(got) == 1.0)


def test_mh_long_and_ragged_sequences(da):
    rng = np.random.RandomState(11)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    seqs = ["".join(aa[i] for i in rng.randint(0, 20, L)) for L in (0, 3, 566, 1500, 2500, 40, 567, 1, 4, 1024, 1027)]
    for k, n_hash in ((4, 50), (5, 300), (9, 64), (2, 257)):
        seeds = da.hash_family_seeds(99, n_hash)
        assert np.array_equal(da.minhash_signatures(seqs, k, n_hash, seed=99), O.signatures(seqs, k, n_hash, seeds))
        rc, want = O.similarity_mh(seqs, k, n_hash, seeds)
        assert_same_f64(da.similarityMH(seqs, k, n_hash, seed=99), want)


def test_mh_extreme_parameters(da):
    """large n_hash (count->double table switched off, many plane groups), large k, k > L"""
    rng = np.random.RandomState(21)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    seqs = ["".join(aa[i] for i in rng.randint(0, 20, L)) for L in (40, 41, 300, 64, 33, 200, 40, 1200)]
    seqs += [seqs[0], seqs[2][:150]]
    for k, n_hash in ((4, 7000), (33, 64), (64, 31), (301, 16), (2, 6143), (3, 6144)):
        seeds = da.hash_family_seeds(5, n_hash)
        sig = O.signatures(seqs, k, n_hash, seeds)
        assert np.array_equal(da.minhash_signatures(seqs, k, n_hash, seed=5), sig), (k, n_hash)
        assert np.array_equal(da.mh_counts(seqs, k, n_hash, seed=5), O.mh_counts(sig)), (k, n_hash)
        rc, want = O.similarity_mh(seqs, k, n_hash, seeds)
        assert_same_f64(da.similarityMH(seqs, k, n_hash, seed=5), want)
    # beyond the 16-bit counters of the compare kernels: chunked by the host entry point (the reference has no upper limit)
    seeds = da.hash_family_seeds(5, 70000)
    rc, want = O.similarity_mh(seqs, 4, 70000, seeds)
    assert_same_f64(da.similarityMH(seqs, 4, 70000, seed=5), want)


def test_mh_row_blocks_and_forced_streaming(da, monkeypatch):
    """row-sharded entry (rect tiles) and the host path's row-block streaming give the same bits"""
    from dynaalign_amd import synth
    seqs = synth.to_strings(*synth.h3n2_like(700, 20))
    seeds = da.hash_family_seeds(12345, 500)
    sig = O.signatures(seqs, 4, 500, seeds)
    want = O.mh_counts(sig)
    for r0, r1 in ((0, 700), (0, 128), (128, 384), (130, 131), (515, 700), (699, 700)):
        got = da.mh_counts(seqs, 4, 500, seed=12345, row_begin=r0, row_end=r1)
        assert np.array_equal(got, want[r0:r1]), (r0, r1)
    monkeypatch.setenv("DYNAALIGN_BLOCK_BYTES", str(700 * 8 * 200))   # forces 128-row blocks
    got = da.similarityMH(seqs, 4, 500, seed=12345)
    assert_same_f64(got, want.astype(np.float64) / 500)


def test_mh_default_seed_is_random_but_valid(da):
    seqs = ["ACDEFGHIKL", "ACDEFGHIKL", "WWWWWYYYYY"]
    a = da.similarityMH(seqs, 4, 64)
    assert a[0, 1] == 1.0 and a[0, 2] == 0.0 and np.array_equal(a, a.T)
    da.set_option("seed", 77)
    try:
        b = da.similarityMH(seqs, 4, 64)
        rc, want = O.similarity_mh(seqs, 4, 64, da.hash_family_seeds(77, 64))
        assert_same_f64(b, want)
    finally:
        da.set_option("seed", None)


def test_device_divide_table_is_ieee(da):
    """count/n_hash computed on the device == host IEEE divide for every count"""
    import torch
    from dynaalign_amd import device
    for n_hash in (1, 3, 50, 500, 4095, 4096, 65535):
        c = np.arange(0, n_hash + 1, max(1, n_hash // 5000)).astype(np.uint16)
        t = torch.from_numpy(c.view(np.int16).copy()).cuda()
        got = device.widen(t, False, n_hash).cpu().numpy()
        assert_same_f64(got, c.astype(np.float64) / n_hash)
    v = np.array([(m << 8) | l for l in range(1, 256) for m in range(0, min(l, 127) + 1)], np.uint16)
    t = torch.from_numpy(v.view(np.int16).copy()).cuda()
    got = device.widen(t, True).cpu().numpy()
    assert_same_f64(got, (v >> 8).astype(np.float64) / (v & 255).astype(np.float64))


# ---------------------------------------------------------------- NW

NW_CASES = [("BLOSUM62", 10, 4), ("BLOSUM45", 10, 4), ("BLOSUM50", 12, 2), ("BLOSUM80", 0, 0),
            ("BLOSUM90", 5, 1), ("BLOSUM100", 10, 4), ("BLOSUM62", 1, 7), ("BLOSUM62", 3, 0)]


@pytest.mark.parametrize("name,go,ge", NW_CASES)
def test_nw_integers_match_golden(da, golden, name, go, ge):
    """ragged lengths 0..30, low-complexity strings (gaps), B/Z/X/*, all six matrices"""
    seqs = [str(s) for s in golden["nw_sequences"]]
    mt, ln, sc = da.nw_pairs(seqs, name, go, ge)
    tag = "nw_%s_%d_%d" % (name, go, ge)
    assert np.array_equal(mt, golden[tag + "_matches"])
    assert np.array_equal(ln, golden[tag + "_len"])
    assert np.array_equal(sc, golden[tag + "_score"])


def test_nw_known_answers_on_gpu(da, kats):
    q = kats["nw_4x4"]
    M = da.similarityNW(q["sequences"], q["matrix"], q["gap_open"], q["gap_ext"])
    assert np.round(np.asarray(M), 6).tolist() == q["rounded6"]
    a = kats["nw_asymmetric"]
    M = da.similarityNW([a["a"], a["b"]])
    assert M[0, 1] == M[1, 0] == a["ab"][0] / a["ab"][1]
    M = da.similarityNW([a["b"], a["a"]])          # lower index is sequence1
    assert M[0, 1] == M[1, 0] == a["ba"][0] / a["ba"][1]


def test_nw_empty_strings_and_nan_bits(da):
    seqs = ["", "A", "", "ACD"]
    rc, want, _ = O.similarity_nw(seqs)
    got = da.similarityNW(seqs)
    assert np.isnan(got[0, 0]) and np.isnan(got[0, 2]) and got[0, 1] == 0.0
    assert_same_f64(got, want)                     # NaN payload/sign identical to the host divide
    mt, ln, sc = da.nw_pairs(seqs)
    rc, omt, oln, osc, _ = O.nw_rows(seqs)
    assert np.array_equal(mt, omt) and np.array_equal(ln, oln) and np.array_equal(sc, osc)
    assert da.similarityNW([]).shape == (0, 0)


def test_nw_evp_config1(da, kats, evp):
    """config 1 input (641 bundled 12-mers) against the reference-recorded checksum"""
    q = kats["nw_evp"]
    W = np.asarray(da.similarityNW(evp))
    assert abs(W[0, 1] - q["w01"]) < 1e-10 and abs(W[0, 2] - q["w02"]) < 1e-10
    assert abs(W.sum() - q["sum_all"]) < 1e-6
    rc, want, _ = O.similarity_nw(evp)
    assert_same_f64(W, want)


@pytest.mark.parametrize("n,L", [(1, 20), (63, 20), (64, 20), (65, 8), (200, 12), (333, 20), (130, 24), (100, 32)])
def test_similarity_nw_matches_oracle(da, n, L):
    from dynaalign_amd import synth
    seqs = synth.to_strings(*synth.h3n2_like(n, L, parent_len=100))
    rc, want, _ = O.similarity_nw(seqs)
    assert rc == 0
    got = da.similarityNW(seqs)
    assert_same_f64(got, want)
    assert got.dimnames[0][-1] == str(n)


def test_nw_row_blocks(da):
    from dynaalign_amd import synth
    seqs = synth.to_strings(*synth.h3n2_like(300, 20))
    seqs[17] = ""                                  # ragged
    seqs[250] = seqs[250][:7]
    rc, omt, oln, osc, _ = O.nw_rows(seqs)
    for r0, r1 in ((0, 300), (0, 64), (64, 192), (70, 71), (130, 300), (299, 300)):
        mt, ln, sc = da.nw_pairs(seqs, row_begin=r0, row_end=r1)
        assert np.array_equal(mt, omt[r0:r1]) and np.array_equal(ln, oln[r0:r1]) and np.array_equal(sc, osc[r0:r1]), (r0, r1)


def test_nw_large_gap_penalties(da):
    """penalties far outside the usual range still agree (int32 arithmetic, NEG sentinel)"""
    from dynaalign_amd import synth
    seqs = synth.to_strings(*synth.uniform_peptides(80, 16, seed=3))
    for go, ge in ((1000, 1000), (0, 50), (100000, 1), (7, 0), (600, 100), (6999, 0), (7001, 0), (-3, 2), (5, -1)):
        rc, omt, oln, osc, _ = O.nw_rows(seqs, 0, None, "BLOSUM62", go, ge)
        mt, ln, sc = da.nw_pairs(seqs, "BLOSUM62", go, ge)
        assert np.array_equal(mt, omt) and np.array_equal(ln, oln) and np.array_equal(sc, osc), (go, ge)


def test_nw_int32_kernel_still_agrees(da, golden, monkeypatch):
    """the general int32 kernel (used when penalties do not fit the combined-key fast path)"""
    monkeypatch.setenv("DYNAALIGN_NW_INT32", "1")
    seqs = [str(s) for s in golden["nw_sequences"]]
    for name, go, ge in (("BLOSUM62", 10, 4), ("BLOSUM80", 0, 0), ("BLOSUM50", 12, 2)):
        mt, ln, sc = da.nw_pairs(seqs, name, go, ge)
        tag = "nw_%s_%d_%d" % (name, go, ge)
        assert np.array_equal(mt, golden[tag + "_matches"]) and np.array_equal(ln, golden[tag + "_len"])
        assert np.array_equal(sc, golden[tag + "_score"])


def _rand_seqs(rng, lengths, alphabet="ARNDCQEGHILKMFPSTWYVBZX*"):
    return ["".join(alphabet[i] for i in rng.randint(0, len(alphabet), L)) for L in lengths]


@pytest.mark.parametrize("lengths", [
    (33, 40, 64, 50, 1, 0, 63, 64),                      # W = 1 (<= 64)
    (100, 128, 65, 90, 20, 127),                         # W = 2
    (130, 192, 150, 7),                                  # W = 3
    (200, 256, 230, 31, 255),                            # W = 4
    (300, 384, 350, 2),                                  # W = 6
    (500, 512, 450, 64, 0),                              # W = 8
    (566, 567, 520, 566, 12),                            # W = 9: the length of the bundled HA sequences
    (700, 768, 650),                                     # W = 12
    (1000, 1024, 900, 33),                               # W = 16
])
def test_nw_long_sequences_match_oracle(da, lengths):
    """wavefront-per-pair anti-diagonal kernel (k_nw_long): ragged lengths, every column-per-lane width"""
    rng = np.random.RandomState(sum(lengths))
    seqs = _rand_seqs(rng, lengths)
    seqs.append(seqs[0])                                  # an exact duplicate
    mutated = list(seqs[1])
    for k in range(0, len(mutated), 7):
        mutated[k] = "W"
    seqs.append("".join(mutated[: max(1, len(mutated) - 3)]))   # near duplicate with a length change -> gaps
    rc, omt, oln, osc, _ = O.nw_rows(seqs)
    assert rc == 0
    mt, ln, sc = da.nw_pairs(seqs)
    assert np.array_equal(mt, omt) and np.array_equal(ln, oln) and np.array_equal(sc, osc)
    rc, want, _ = O.similarity_nw(seqs)
    assert_same_f64(da.similarityNW(seqs), want)
    mt2, ln2, sc2 = da.nw_pairs(seqs, row_begin=1, row_end=len(seqs) - 1)       # row-block path
    assert np.array_equal(mt2, omt[1:-1]) and np.array_equal(ln2, oln[1:-1]) and np.array_equal(sc2, osc[1:-1])


def test_nw_long_other_matrices_and_gaps(da):
    rng = np.random.RandomState(8)
    seqs = _rand_seqs(rng, (120, 80, 200, 150, 90), "AGW") + _rand_seqs(rng, (140, 60))
    for name, go, ge in (("BLOSUM45", 10, 4), ("BLOSUM100", 3, 1), ("BLOSUM62", 0, 0), ("BLOSUM80", 25, 9)):
        rc, omt, oln, osc, _ = O.nw_rows(seqs, 0, None, name, go, ge)
        mt, ln, sc = da.nw_pairs(seqs, name, go, ge)
        assert np.array_equal(mt, omt) and np.array_equal(ln, oln) and np.array_equal(sc, osc), (name, go, ge)


@pytest.mark.parametrize("lengths", [(33, 48, 40, 20, 0, 47, 48, 35), (64, 49, 60, 5, 64, 33)])
def test_nw_mid_length_lane_per_pair(da, lengths):
    """33..64 residues: lane-per-pair kernel with the 15-bit-score combined key; penalties outside its
    range fall back to the wavefront-per-pair kernel -- both must agree with the oracle"""
    rng = np.random.RandomState(len(lengths))
    seqs = _rand_seqs(rng, lengths) + _rand_seqs(rng, lengths[:3], "AGW")
    seqs.append(seqs[0][:-2] + "WW")
    for name, go, ge in (("BLOSUM62", 10, 4), ("BLOSUM100", 0, 0), ("BLOSUM45", 2400, 0), ("BLOSUM80", 12, 19),
                         ("BLOSUM62", 10, 20), ("BLOSUM50", 3000, 1)):
        rc, omt, oln, osc, _ = O.nw_rows(seqs, 0, None, name, go, ge)
        mt, ln, sc = da.nw_pairs(seqs, name, go, ge)
        assert np.array_equal(mt, omt) and np.array_equal(ln, oln) and np.array_equal(sc, osc), (name, go, ge)
    rc, want, _ = O.similarity_nw(seqs)
    assert_same_f64(da.similarityNW(seqs), want)


@pytest.mark.parametrize("lengths", [(1025, 1024, 1027, 40, 0), (1500, 2500, 2049, 2048, 1, 300), (3100, 2047, 1024, 3073)])
def test_nw_sequences_beyond_1024_residues(da, lengths):
    """k_nw_xlong: column blocks of 1024 with the block boundary spilled to HBM (the reference is O(m n) for any length,
    src/pairwiseSeqAlign.cpp:216-219): block edges exactly at / next to multiples of 1024, ragged partners, both
    directions (long row / short column and the reverse), (matches, length, score) and the float64 matrix vs the oracle"""
    rng = np.random.RandomState(sum(lengths))
    seqs = _rand_seqs(rng, lengths, "ARNDCQEGHILKMFPSTWYV")
    seqs.append(seqs[0])                                  # exact duplicate of a long one
    mutated = list(seqs[1])
    for k in range(0, len(mutated), 11):
        mutated[k] = "W"
    del mutated[500:517]                                  # a deletion -> long gap
    seqs.append("".join(mutated))
    rc, omt, oln, osc, _ = O.nw_rows(seqs)
    assert rc == 0
    mt, ln, sc = da.nw_pairs(seqs)
    assert np.array_equal(mt, omt) and np.array_equal(ln, oln) and np.array_equal(sc, osc)
    rc, want, _ = O.similarity_nw(seqs)
    assert_same_f64(da.similarityNW(seqs), want)
    mt2, ln2, sc2 = da.nw_pairs(seqs, "BLOSUM45", 3, 1, row_begin=1, row_end=len(seqs) - 1)       # row-block path, other penalties
    rc, omt2, oln2, osc2, _ = O.nw_rows(seqs, 1, len(seqs) - 1, "BLOSUM45", 3, 1)
    assert np.array_equal(mt2, omt2) and np.array_equal(ln2, oln2) and np.array_equal(sc2, osc2)


def test_nw_too_long_sequences_fail_loudly(da):
    """the alignment length travels in 16 bits: m + n <= 65535, i.e. sequences up to 32767 residues"""
    with pytest.raises(da.DynaAlignError) as ei:
        da.similarityNW(["A" * 32768, "C" * 10])
    assert ei.value.code == 10 and "32767" in str(ei.value)


# ---------------------------------------------------------------- BASELINE sizes (properties)

def _workload_10k(name):
    from dynaalign_amd import synth
    res, off = (synth.uniform_peptides(10000, 20, seed=7) if name == "uniform" else synth.h3n2_like(10000, 20))
    return res, off, synth.to_strings(res, off)


class _env:
    """route switches of the library, for the duration of a call"""
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        import os
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update({k: str(v) for k, v in self.kv.items()})

    def __exit__(self, *exc):
        import os
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("workload", ["uniform", "h3n2like"])
def test_mh_10k_whole_matrix_against_the_oracle(da, workload):
    """BASELINE config 2 (10k synthetic 20-mers, k=4, n_hash=500), nothing sampled: the ENTIRE 10 000 x 10 000 float64 result of the
    reference-boundary call and of every route of the device call (direct kernels, duplicate route, sparse route) against the oracle's
    similarityMH bit for bit (src/minHash.cpp:160-178), all signatures, and all 10^8 uint16 match counts against the oracle's compare
    loop run on the ORACLE's signatures."""
    import torch
    from dynaalign_amd import device
    res, off, seqs = _workload_10k(workload)
    n, k, n_hash = 10000, 4, 500
    seeds = da.hash_family_seeds(12345, n_hash)
    rc, want = O.similarity_mh(seqs, k, n_hash, seeds)                       # the oracle's whole call
    assert rc == 0
    wb = bits(want)
    osig = O.signatures(seqs, k, n_hash, seeds)
    ocnt = O.mh_counts(osig)                                                 # [n][n] uint16, diagonal = n_hash
    assert np.array_equal(bits(ocnt.astype(np.float64) / n_hash), wb)        # the oracle agrees with itself (count / n_hash IS the matrix)

    assert np.array_equal(da.minhash_signatures(seqs, k, n_hash, seed=12345), osig)
    assert np.array_equal(da.mh_counts(seqs, k, n_hash, seed=12345), ocnt)
    M = np.asarray(da.similarityMH(seqs, k, n_hash, seed=12345))             # what R sees (host-pointer boundary)
    assert M.shape == (n, n) and np.array_equal(bits(M), wb)
    del M

    ds = device.DeviceSequences(res, off)
    routes = {}

    def dev_call(**env):
        with _env(**env):
            out = device.similarity_mh(ds, k, n_hash, seeds)
            torch.cuda.synchronize()
        r = device.mh_last_route()
        got = out.cpu().numpy()
        del out
        assert np.array_equal(bits(got), wb), "route %s differs from the oracle" % (r,)
        return r
    r = dev_call()                                                           # the route the library picks by itself
    routes["default"] = r
    r = dev_call(DYNAALIGN_MH_NO_DEDUP=1, DYNAALIGN_MH_NO_SPARSE=1)
    assert not r["dedup"] and not r["sparse"]
    if workload == "uniform":
        assert routes["default"]["sparse"], routes                           # uniform 10k: signatures rarely agree -> sparse route
    else:
        assert routes["default"]["dedup"], routes                            # h3n2-like 10k: duplicates collapsed
        for form in ("rows", "tiles", "pipe", "rowspipe"):                    # every expansion form of the duplicate route
            r = dev_call(DYNAALIGN_MH_EXPAND=form)
            assert r["dedup"], (form, r)
    # device compare in its uint16 form on the planes (what the sharded / edge paths consume)
    sig, planes = device.minhash_signatures(ds, k, n_hash, seeds)
    assert np.array_equal(sig[:, :n_hash].cpu().numpy().view(np.uint32), osig)
    from dynaalign_amd import _capi
    cnt = device.mh_compare(planes, n, n_hash, kind=_capi.DA_OUT_COMPACT)
    assert np.array_equal(cnt.cpu().numpy().view(np.uint16), ocnt)


@pytest.mark.parametrize("workload", ["uniform", "h3n2like"])
def test_nw_10k_whole_matrix_against_the_oracle(da, workload):
    """BASELINE config 3 (10k synthetic 20-mers, BLOSUM62 / 10 / 4, full N x N), nothing sampled: (matches, length, score) of all 10^8
    ordered elements against the oracle's DP + traceback (src/pairwiseSeqAlign.cpp:209-313 under the driver's calc(seq[min], seq[max]),
    :340-352), and the float64 ratio matrix as uint64 -- through the host-pointer call and through the device call's direct and
    duplicate routes."""
    import torch
    from dynaalign_amd import device, _capi
    res, off, seqs = _workload_10k(workload)
    n = 10000
    rc, omt, oln, osc, _ = O.nw_rows(seqs, 0, n)                             # OpenMP over rows: ~10 s on the box's 16 threads
    assert rc == 0
    want = omt / oln.astype(np.float64)
    wb = bits(want)
    W = np.asarray(da.similarityNW(seqs))
    assert W.shape == (n, n) and np.array_equal(bits(W), wb)
    del W
    half = n // 2                                                            # integers in two row blocks (3 x 200 MB each)
    for r0 in (0, half):
        mt, ln, sc = da.nw_pairs(seqs, row_begin=r0, row_end=r0 + half)
        assert np.array_equal(mt, omt[r0:r0 + half]) and np.array_equal(ln, oln[r0:r0 + half]) and np.array_equal(sc, osc[r0:r0 + half])
    ds = device.DeviceSequences(res, off)
    assert int(device.nw_encode(ds).item()) == 0
    seen = set()
    for env in ({}, {"DYNAALIGN_NW_NO_DEDUP": 1}, {"DYNAALIGN_NW_DEDUP_MIN_N": 1}):
        with _env(**env):
            out = device.nw(ds, "BLOSUM62", 10, 4, 0, n, True, _capi.DA_OUT_F64)
            torch.cuda.synchronize()
        r = device.nw_last_route()
        seen.add(bool(r["dedup"]))
        assert np.array_equal(bits(out.cpu().numpy()), wb), "NW route %s differs from the oracle" % (r,)
        del out
    if workload == "h3n2like":
        assert seen == {True, False}                                         # both the duplicate route and the direct kernel were hit


def test_100k_headline_workload_properties(da):
    """configs[3] (the bench workload): 100 000 h3n2-like 20-mers through the device API, everything
    resident in HBM.  Whole-matrix properties + oracle compares on row samples; torch is used only
    to reduce/compare on the device (checker side)."""
    import torch
    from dynaalign_amd import device, synth, _capi
    n, n_hash = 100000, 500
    res, off = synth.h3n2_like(n, 20)
    seqs = synth.to_strings(res, off)
    seeds = da.hash_family_seeds(12345, n_hash)
    ds = device.DeviceSequences(res, off)
    sig, planes = device.minhash_signatures(ds, 4, n_hash, seeds)
    sig_h = sig[:, :n_hash].cpu().numpy().view(np.uint32)
    for r0 in (0, 31337, 99900):                                   # bit-exact signatures on samples
        assert np.array_equal(sig_h[r0:r0 + 50], O.signatures(seqs[r0:r0 + 50], 4, n_hash, seeds))
    cnt = device.mh_compare(planes, n, n_hash, kind=_capi.DA_OUT_COMPACT)          # int16 tensor, 20 GB
    # sum of all match counts without walking pairs: sum_h sum_v multiplicity(v in column h)^2
    want_total = 0
    for h in range(n_hash):
        _, c = np.unique(sig_h[:, h], return_counts=True)
        want_total += int((c.astype(np.int64) ** 2).sum())
    got_total = 0
    for r0 in range(0, n, 10000):
        got_total += int(cnt[r0:r0 + 10000].to(torch.int32).sum(dtype=torch.int64).item())
    assert got_total == want_total
    assert bool((torch.diagonal(cnt) == n_hash).all())
    for (a, b) in ((0, 70000), (12800, 12928), (99000, 500)):      # symmetry on sampled blocks
        assert torch.equal(cnt[a:a + 900, b:b + 900], cnt[b:b + 900, a:a + 900].T)
    for r0 in (0, 49999, 99980):                                   # oracle compare on row samples
        want = np.stack([(sig_h[i][None, :] == sig_h).sum(1) for i in range(r0, r0 + 12)]).astype(np.int16)
        assert np.array_equal(cnt[r0:r0 + 12].cpu().numpy(), want)
    out = device.mh_compare(planes, n, n_hash, kind=_capi.DA_OUT_F64)              # float64, 80 GB
    ratio = torch.from_numpy(np.arange(n_hash + 1, dtype=np.float64) / n_hash).cuda()   # host IEEE divide
    for r0 in range(0, n, 5000):                                   # f64 matrix == counts / n_hash everywhere
        assert torch.equal(out[r0:r0 + 5000], ratio[cnt[r0:r0 + 5000].long()])
    # the one-call route bench.py times (duplicates collapsed: K1 / K1b / K2 on the 44 931 unique strings + index expansion)
    # must reproduce the whole 80 GB matrix bit for bit
    out.fill_(-1.0)
    device.similarity_mh(ds, 4, n_hash, seeds, out=out)
    route = device.mh_last_route()
    assert route["dedup"] and route["unique"] == len(set(seqs))
    for r0 in range(0, n, 5000):
        assert torch.equal(out[r0:r0 + 5000], ratio[cnt[r0:r0 + 5000].long()])
    del cnt
    # NW on the same set: sampled rows against the oracle, symmetry, diagonal
    assert int(device.nw_encode(ds).item()) == 0
    device.nw(ds, out=out)
    assert bool((torch.diagonal(out) == 1.0).all())
    for (a, b) in ((0, 70000), (6400, 6464), (99000, 500)):
        assert torch.equal(out[a:a + 900, b:b + 900], out[b:b + 900, a:a + 900].T)
    for r0 in (0, 50001, 99995):
        rc, mt, ln, _, _ = O.nw_rows(seqs, r0, r0 + 4)
        assert rc == 0
        assert_same_f64(out[r0:r0 + 4].cpu().numpy(), mt / ln.astype(np.float64))


def test_host_widening_and_device_float64_routes_agree(da, monkeypatch):
    """host-pointer float64 results: by default the device hands over uint16 codes and the HOST widens while copying (a quarter
    of the PCIe bytes; same IEEE divide); DYNAALIGN_NO_HOST_WIDEN=1 copies the device's float64 matrix.  Same bits, including
    the NaN of empty-vs-empty NW pairs, on an input large enough for the pinned-ring pipeline (> 64 MiB of codes)."""
    from dynaalign_amd import synth
    seqs = synth.to_strings(*synth.h3n2_like(6000, 20))
    seqs[17] = ""
    seqs[4321] = ""
    a = np.asarray(da.similarityMH(seqs, 4, 300, seed=3))
    w = np.asarray(da.similarityNW(seqs))
    monkeypatch.setenv("DYNAALIGN_NO_HOST_WIDEN", "1")
    b = np.asarray(da.similarityMH(seqs, 4, 300, seed=3))
    v = np.asarray(da.similarityNW(seqs))
    assert_same_f64(a, b)
    assert_same_f64(w, v)
    assert np.isnan(w[17, 4321]) and np.isnan(w[17, 17]) and w[17, 0] == 0.0
    rc, mt, ln, _, _ = O.nw_rows(seqs, 17, 18)
    with np.errstate(invalid="ignore", divide="ignore"):
        assert_same_f64(w[17], mt[0].astype(np.float64) / ln[0].astype(np.float64))
