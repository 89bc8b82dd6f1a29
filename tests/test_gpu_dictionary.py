"""GPU tests of K1b (dict_kernels.hip): the exact 16-bit re-coding of the signatures.

The compare kernel must give the same match counts from the dictionary codes (16 planes per 32
hash functions) as from the raw signature bits (32 planes), and both must equal the oracle
(reference src/minHash.cpp:160-178).  Inputs are chosen to stress what the re-coding relies on:
singletons (uniform peptides: most values occur once), repeated values (h3n2-like windows,
duplicated sequences), the all-equal column (sequences shorter than k: UINT32_MAX everywhere),
row blocks that cut the diagonal at arbitrary offsets, and the id-count worst case n = 131068
with every value occurring exactly twice."""
import os

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def da(built):
    import dynaalign_amd
    from dynaalign_amd import _capi
    assert _capi.load().da_device_count() > 0
    return dynaalign_amd


def _counts(sig_h):
    return np.stack([(row[None, :] == sig_h).sum(1) for row in sig_h]).astype(np.uint16)


def _bits_needed(sig_h):
    """planes the library should pick: ids for the values seen >= 2 times in a column, plus 2 singleton codes"""
    most = 0
    for h in range(sig_h.shape[1]):
        _, c = np.unique(sig_h[:, h], return_counts=True)
        most = max(most, int((c >= 2).sum()))
    return 8 if most + 2 <= 256 else 12 if most + 2 <= 4096 else 14 if most + 2 <= 16384 else 15 if most + 2 <= 32768 else 16


def _planes_both(da, res, off, k, n_hash, seed=12345, min_bits=0):
    from dynaalign_amd import device
    seeds = da.hash_family_seeds(seed, n_hash)
    ds = device.DeviceSequences(res, off)
    sig, pc = device.minhash_signatures(ds, k, n_hash, seeds, min_plane_bits=min_bits)
    _, p32 = device.minhash_signatures(ds, k, n_hash, seeds, raw_planes=True)
    assert pc.bits in (8, 12, 14, 15, 16) and pc.bits >= min_bits and p32.bits == 32
    return ds.n, sig[:, :n_hash].cpu().numpy().view(np.uint32), pc, p32


@pytest.mark.parametrize("gen,n", [("uniform_peptides", 1), ("uniform_peptides", 2), ("uniform_peptides", 63),
                                   ("uniform_peptides", 64), ("uniform_peptides", 65), ("uniform_peptides", 1000),
                                   ("h3n2_like", 129), ("h3n2_like", 1500), ("uniform_peptides", 9000)])
@pytest.mark.parametrize("n_hash", [500, 31, 64])
@pytest.mark.parametrize("min_bits", [0, 12, 14, 15, 16])
def test_codes_give_the_counts_of_the_raw_signatures(da, gen, n, n_hash, min_bits):
    """8 / 12 / 16 code planes (whatever the data needs, or forced up) == 32 raw planes == numpy"""
    from dynaalign_amd import device, synth, _capi
    if n >= 9000 and n_hash != 500:
        pytest.skip("one large case is enough")
    res, off = getattr(synth, gen)(n, 20)
    n, sig_h, p16, p32 = _planes_both(da, res, off, 4, n_hash, min_bits=min_bits)
    if min_bits == 0:
        assert p16.bits == _bits_needed(sig_h)
    c16 = device.mh_compare(p16, n, n_hash, kind=_capi.DA_OUT_COMPACT).cpu().numpy().view(np.uint16)
    c32 = device.mh_compare(p32, n, n_hash, kind=_capi.DA_OUT_COMPACT).cpu().numpy().view(np.uint16)
    assert np.array_equal(c16, c32)
    if n <= 1500:
        assert np.array_equal(c16, _counts(sig_h))
    f16 = device.mh_compare(p16, n, n_hash).cpu().numpy()
    f32 = device.mh_compare(p32, n, n_hash).cpu().numpy()
    assert np.array_equal(f16.view(np.uint64), f32.view(np.uint64))
    assert np.all(np.diagonal(f16) == 1.0)


def test_duplicates_short_sequences_and_identical_input(da):
    """duplicated sequences (every value repeated), sequences shorter than k (UINT32_MAX in every
    column, reference src/minHash.cpp:98-103,140) and an input of identical sequences"""
    import dynaalign_amd as dam
    rng = np.random.RandomState(3)
    alpha = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    base = ["".join(map(chr, alpha[rng.randint(0, 20, 20)])) for _ in range(150)]
    seqs = base + base[:70] + ["AC", "A", "", "ACD"] * 5 + ["MKTIIALSYIFCLVFA"] * 40
    rng.shuffle(seqs)
    seeds = dam.hash_family_seeds(99, 200)
    rc, want = O.similarity_mh(seqs, 4, 200, seeds)
    assert rc == 0
    got = dam.similarityMH(seqs, 4, 200, seed=99)
    assert np.array_equal(np.asarray(got).view(np.uint64), want.view(np.uint64))
    same = ["MKTIIALSYIFCLVFAQK"] * 300
    got = dam.similarityMH(same, 4, 50, seed=1)
    assert np.all(np.asarray(got) == 1.0)


@pytest.mark.parametrize("row_begin,row_end", [(0, 700), (5, 133), (127, 129), (128, 640), (300, 301), (693, 700)])
def test_row_blocks_keep_the_diagonal(da, row_begin, row_end):
    """rectangular mode: the forced diagonal must follow arbitrary row offsets"""
    from dynaalign_amd import device, synth, _capi
    res, off = synth.uniform_peptides(700, 20)
    n, sig_h, p16, p32 = _planes_both(da, res, off, 4, 100)
    assert p16.bits == _bits_needed(sig_h)
    want = _counts(sig_h)[row_begin:row_end]
    for p in (p16, p32):
        got = device.mh_compare(p, n, 100, row_begin, row_end, False, _capi.DA_OUT_COMPACT)
        assert np.array_equal(got.cpu().numpy().view(np.uint16), want)


def test_host_path_same_with_raw_planes(da):
    """DYNAALIGN_PLANE_BITS=32 (raw signature bits) and the default (dictionary codes) agree bit for bit"""
    from dynaalign_amd import synth
    res, off = synth.h3n2_like(2500, 20)
    seqs = synth.to_strings(res, off)
    a = np.asarray(da.similarityMH(seqs, 4, 500, seed=12345))
    thr_a, ei_a, ej_a, ew_a = da.similarityMH_edges(seqs, 4, 500, 0.8, seed=12345)
    os.environ["DYNAALIGN_PLANE_BITS"] = "32"
    try:
        b = np.asarray(da.similarityMH(seqs, 4, 500, seed=12345))
        thr_b, ei_b, ej_b, ew_b = da.similarityMH_edges(seqs, 4, 500, 0.8, seed=12345)
    finally:
        del os.environ["DYNAALIGN_PLANE_BITS"]
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    assert thr_a == thr_b and np.array_equal(ei_a, ei_b) and np.array_equal(ej_a, ej_b) and np.array_equal(ew_a, ew_b)


def test_worst_case_id_count(da):
    """n = 131068 (the largest n the 16-bit codes are guaranteed for), every sequence present exactly
    twice: close to n/2 repeated values per column.  Checked through properties on the device."""
    import torch
    from dynaalign_amd import device, synth, _capi
    n, n_hash = 131068, 64
    res, off = synth.uniform_peptides(n // 2, 20, seed=21)
    res2 = np.concatenate([res[:off[-1]], res[:off[-1]]])
    off2 = np.concatenate([off, off[1:] + off[-1]])
    seeds = da.hash_family_seeds(5, n_hash)
    ds = device.DeviceSequences(res2, off2)
    assert ds.n == n
    # k = 8: 13 shingles out of 20^8 per sequence, so nearly every sequence has its own value in a column
    sig, p16 = device.minhash_signatures(ds, 8, n_hash, seeds)
    assert p16.bits == 16
    sig_h = sig[:, :n_hash].cpu().numpy().view(np.uint32)
    cnt = device.mh_compare(p16, n, n_hash, kind=_capi.DA_OUT_COMPACT)        # 34 GB
    want_total = 0
    most = 0
    for h in range(n_hash):
        _, c = np.unique(sig_h[:, h], return_counts=True)
        want_total += int((c.astype(np.int64) ** 2).sum())
        most = max(most, int((c >= 2).sum()))
    assert most > 60000                                                       # the test does exercise large id counts
    got_total = 0
    for r0 in range(0, n, 8192):
        got_total += int(cnt[r0:r0 + 8192].to(torch.int32).sum(dtype=torch.int64).item())
    assert got_total == want_total
    assert bool((torch.diagonal(cnt) == n_hash).all())
    half = n // 2
    assert bool((torch.diagonal(cnt, offset=half) == n_hash).all())           # a sequence and its copy
    for r0 in (0, 65530, 131000):
        want = np.stack([(sig_h[i][None, :] == sig_h).sum(1) for i in range(r0, r0 + 8)]).astype(np.int16)
        assert np.array_equal(cnt[r0:r0 + 8].cpu().numpy(), want)
    del cnt
    # beyond the guarantee the library must switch to the raw planes by itself
    res3 = np.concatenate([res2, res[:20 * 4]])
    off3 = np.concatenate([off2, off2[-1] + 20 * np.arange(1, 5)])
    ds3 = device.DeviceSequences(res3, off3)
    _, p = device.minhash_signatures(ds3, 8, n_hash, seeds)
    assert p.bits == 32


def _unmix(z):
    """inverse of the dictionary's key mixer (murmur3's 32-bit finaliser)"""
    z = int(z)
    z ^= z >> 16
    z = (z * 0x7ED1B41D) & 0xFFFFFFFF          # inverse of 0xc2b2ae35 mod 2^32
    z ^= (z >> 13) ^ (z >> 26)
    z = (z * 0xA5CB9243) & 0xFFFFFFFF          # inverse of 0x85ebca6b mod 2^32
    z ^= z >> 16
    return z


def _mix(x):
    x = int(x)
    x ^= x >> 16
    x = (x * 0x85EBCA6B) & 0xFFFFFFFF
    x ^= x >> 13
    x = (x * 0xC2B2AE35) & 0xFFFFFFFF
    x ^= x >> 16
    return x


@pytest.mark.parametrize("n", [2, 8191, 8192, 8193, 16385, 30000])
def test_crafted_signature_columns(da, n):
    """signature matrices written directly (no hashing), one pathology per column: all equal, all distinct,
    two values, 0 / 0xFFFFFFFF, the keys whose mixed value is the LDS table's empty marker of each
    partition (they live in another partition, so they must still be found), heavy duplicates."""
    import torch
    from dynaalign_amd import device, _capi
    assert all(_mix(_unmix(z)) == z for z in (0, 1, 0x12345678, 0xFFFFFFFF, 0x80000000))
    rng = np.random.RandomState(n)
    R = max(2, -(-n // 8192))
    empties = [(((q << 32) + R - 1) // R) for q in range(R)]          # smallest mixed key of partition q
    special = np.array([_unmix(z) for z in empties] + [0, 0xFFFFFFFF, 1, 0x80000000], np.uint64)
    cols = [np.full(n, 7, np.uint64),
            rng.permutation(n).astype(np.uint64) * 40503 % (1 << 32),
            rng.randint(0, 2, n).astype(np.uint64) * 0xFFFFFFFF,
            special[rng.randint(0, len(special), n)],
            np.where(rng.rand(n) < 0.5, special[rng.randint(0, len(special), n)], rng.randint(0, 1 << 32, n).astype(np.uint64)),
            rng.randint(0, max(2, n // 3), n).astype(np.uint64),
            rng.randint(0, 1 << 32, n).astype(np.uint64)]
    n_hash = 70                                                         # 3 groups, the last one partial
    sig_h = np.stack([cols[h % len(cols)] if h < 2 * len(cols) else rng.randint(0, 50, n).astype(np.uint64)
                      for h in range(n_hash)], 1).astype(np.uint32)
    sig = torch.zeros((n, device.sig_ld(n_hash)), dtype=torch.int32, device="cuda")
    sig[:, :n_hash] = torch.from_numpy(sig_h.view(np.int32)).cuda()
    pc = device.mh_planes(sig, n, n_hash)
    p32 = device.mh_planes(sig, n, n_hash, min_plane_bits=32)
    assert pc.bits == _bits_needed(sig_h) and p32.bits == 32
    rows = (0, min(n, 140)) if n > 3000 else (0, n)
    want = np.stack([(sig_h[i][None, :] == sig_h).sum(1) for i in range(*rows)]).astype(np.uint16)
    for p in (pc, p32):
        got = device.mh_compare(p, n, n_hash, rows[0], rows[1], rows == (0, n), _capi.DA_OUT_COMPACT)
        assert np.array_equal(got.cpu().numpy().view(np.uint16), want)
    if n > 3000:                                                        # whole matrix: both operands agree everywhere
        a = device.mh_compare(pc, n, n_hash, kind=_capi.DA_OUT_COMPACT)
        b = device.mh_compare(p32, n, n_hash, kind=_capi.DA_OUT_COMPACT)
        assert torch.equal(a, b)


@pytest.mark.parametrize("n,n_hash", [(1000, 500), (1153, 70), (2048, 33), (640, 31), (2500, 64), (3001, 129), (1300, 511), (1300, 512),
                                      (5000, 500)])
@pytest.mark.parametrize("bits", [12, 14, 15, 16])
def test_hand_scheduled_and_compiled_12_plane_kernels_agree(da, n, n_hash, bits):
    """symmetric 12-plane compares run a hand-scheduled stage loop on interior tiles and the compiled kernel on diagonal /
    border tiles.  Three routes, same operand: the one-tile-per-workgroup k_mh_compare_a12 (default), the PERSISTENT kernel
    k_mh_compare_p12 (DYNAALIGN_K2_PERSIST=1; a workgroup walks a sequence of tiles, ring never drains; n_hash <= 32 and
    float64 n_hash >= 512 fall back to a12 by themselves) and the compiled kernel everywhere (DYNAALIGN_K2_NO_ASM=1).  Both
    output kinds, 1 / 2 / 3 / 5 / 16 stages (ring phases 0, 1, 2 between tiles), n a multiple of 128 or not, few tiles per
    workgroup (n = 1000) and many (n = 5000: 780 tiles over 1024 workgroups -> some get one tile, some none)."""
    import torch
    from dynaalign_amd import device, synth, _capi
    res, off = synth.h3n2_like(n, 20)
    n, sig_h, p12, p32 = _planes_both(da, res, off, 4, n_hash, min_bits=bits)
    assert p12.bits == bits          # 16: k_mh_compare_a16 on the padded twin of the operand (8 steps per stage, ring of two stages); 14: its seven-step form
    want = _counts(sig_h) if n <= 3100 else None
    got = {}
    for tag, env in (("persistent", "DYNAALIGN_K2_PERSIST"), ("one_tile", None), ("compiled", "DYNAALIGN_K2_NO_ASM")):
        if env:
            os.environ[env] = "1"
        try:
            got[tag] = (device.mh_compare(p12, n, n_hash, kind=_capi.DA_OUT_COMPACT).cpu().numpy().view(np.uint16),
                        device.mh_compare(p12, n, n_hash).cpu().numpy())
        finally:
            if env:
                os.environ.pop(env, None)
    for tag in ("persistent", "one_tile"):
        assert np.array_equal(got[tag][0], got["compiled"][0]), tag
        assert np.array_equal(got[tag][1].view(np.uint64), got["compiled"][1].view(np.uint64)), tag
    if want is not None:
        assert np.array_equal(got["compiled"][0], want)
        assert np.array_equal(got["compiled"][1], want.astype(np.float64) / n_hash)


@pytest.mark.parametrize("wg_per_cu", [1, 2, 3])
def test_persistent_kernel_with_few_workgroups(da, wg_per_cu):
    """fewer resident workgroups (as when the occupancy query answers less than 4): every workgroup walks a longer tile
    sequence; results must not depend on the grid"""
    import subprocess, sys
    code = (
        "import os, sys; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "import numpy as np, torch, dynaalign_amd as da\n"
        "from dynaalign_amd import device, synth, _capi\n"
        "res, off = synth.h3n2_like(9000, 20)\n"
        "ds = device.DeviceSequences(res, off)\n"
        "sig, p = device.minhash_signatures(ds, 4, 500, da.hash_family_seeds(12345, 500), min_plane_bits=12)\n"
        "a = device.mh_compare(p, 9000, 500); a16 = device.mh_compare(p, 9000, 500, kind=_capi.DA_OUT_COMPACT)\n"
        "os.environ['DYNAALIGN_K2_NO_ASM'] = '1'\n"
        "b = device.mh_compare(p, 9000, 500); b16 = device.mh_compare(p, 9000, 500, kind=_capi.DA_OUT_COMPACT)\n"
        "assert torch.equal(a.view(torch.int64), b.view(torch.int64)) and torch.equal(a16, b16)\n"
        "print('ok')\n") % (ROOT, ROOT)
    env = dict(os.environ, DYNAALIGN_K2_WG_PER_CU=str(wg_per_cu), DYNAALIGN_K2_PERSIST="1")
    out = subprocess.check_output([sys.executable, "-c", code], env=env, timeout=300)
    assert out.decode().strip().endswith("ok")


@pytest.mark.parametrize("n_hash", [4607, 4608, 7000])
def test_large_n_hash_with_interior_tiles(da, n_hash):
    """The hand-scheduled 12-plane kernel keeps its count -> double table (n_hash + 1 entries) in its 36 KiB ring:
    4608 doubles.  n_hash = 4607 is the last that fits; larger ones must take the general kernel (direct divide).
    n = 400 gives interior off-diagonal tiles; duplicated sequences give match counts == n_hash off the diagonal,
    i.e. reads at the very end of the table."""
    from dynaalign_amd import device, synth, _capi
    res, off = synth.h3n2_like(400, 20)
    res = res.reshape(400, 20).copy()
    res[300:400] = res[0:100]                                   # rows 300.. duplicate rows 0..: count == n_hash in tile (0, 2/3)
    res = res.reshape(-1)
    n, sig_h, p12, p32 = _planes_both(da, res, off, 4, n_hash, min_bits=12)
    assert p12.bits == 12
    want = _counts(sig_h)
    assert want[0, 300] == n_hash
    c12 = device.mh_compare(p12, n, n_hash, kind=_capi.DA_OUT_COMPACT).cpu().numpy().view(np.uint16)
    assert np.array_equal(c12, want)
    f12 = device.mh_compare(p12, n, n_hash).cpu().numpy()
    want_f = want.astype(np.float64) / n_hash                  # src/minHash.cpp:174
    assert np.array_equal(f12.view(np.uint64), want_f.view(np.uint64))
