"""GPU tests of the duplicate-collapsing route of similarityMH (api.cpp mh_full_symmetric = da_dev_similarity_mh): byte-identical
sequences are collapsed, K1 / K1b / K2 run on the table of unique strings and the n x n matrix is an index expansion of the
U x U count table (minhash_kernels.hip: k_gather_columns + k_expand_rows for interior tiles, k_expand_unique for diagonal and
border tiles).  Must be bit-identical to the three kernels run on all n rows and to the oracle (reference src/minHash.cpp:119-188)."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def da(built):
    import dynaalign_amd
    from dynaalign_amd import _capi
    assert _capi.load().da_device_count() > 0
    return dynaalign_amd


@pytest.fixture()
def small_n_route(monkeypatch):
    monkeypatch.setenv("DYNAALIGN_MH_DEDUP_MIN_N", "1")


def oracle(seqs, k, n_hash, seed=12345):
    rc, m = O.similarity_mh(seqs, k, n_hash, O.seeds(seed, n_hash))
    assert rc == 0
    return m


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def run(seqs, k, n_hash, seed=12345, direct=False):
    import torch
    from dynaalign_amd import device, synth
    import dynaalign_amd as da_
    res, off = O.pack(seqs)
    ds = device.DeviceSequences(np.asarray(res, np.uint8), np.asarray(off, np.int64))
    seeds = da_.hash_family_seeds(seed, n_hash)
    if direct:
        os.environ["DYNAALIGN_MH_NO_DEDUP"] = "1"
    try:
        out = device.similarity_mh(ds, k, n_hash, seeds)
        torch.cuda.synchronize()
        route = device.mh_last_route()
    finally:
        os.environ.pop("DYNAALIGN_MH_NO_DEDUP", None)
    return out.cpu().numpy(), route


def duplicated_set(rng, n_pool, n_draw, n_single, lo=0, hi=41):
    alpha = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    mk = lambda: "".join(map(chr, alpha[rng.randint(0, 20, rng.randint(lo, hi))]))
    pool = [mk() for _ in range(n_pool)]
    seqs = [pool[q] for q in rng.randint(0, n_pool, n_draw)] + [mk() for _ in range(n_single)]
    rng.shuffle(seqs)
    return seqs


@pytest.mark.parametrize("seed,n_hash,k", [(1, 50, 4), (2, 500, 4), (3, 33, 2)])
def test_route_matches_direct_and_oracle(da, small_n_route, seed, n_hash, k):
    """few distinct strings drawn many times in random order + singletons, ragged lengths (some shorter than k, some empty)"""
    rng = np.random.RandomState(seed)
    seqs = duplicated_set(rng, 50, 600, 130)
    got, route = run(seqs, k, n_hash)
    assert route["dedup"] and route["unique"] == len(set(seqs)) and route["n"] == len(seqs)
    direct, droute = run(seqs, k, n_hash, direct=True)
    assert not droute["dedup"]
    want = oracle(seqs, k, n_hash)
    assert same(direct, want)
    assert same(got, want)


def test_interior_tiles_take_the_two_pass_expansion(da, small_n_route):
    """n = 1000 (7 full 128-tiles + a border): interior off-diagonal tiles go through k_gather_columns + k_expand_rows, the rest
    through k_expand_unique; an odd n puts the last column on the border path"""
    rng = np.random.RandomState(7)
    for n_draw, n_single in ((800, 200), (900, 101)):
        seqs = duplicated_set(rng, 120, n_draw, n_single, 8, 25)
        got, route = run(seqs, 4, 100)
        assert route["dedup"]
        direct, _ = run(seqs, 4, 100, direct=True)
        assert same(got, direct)
        want = oracle(seqs, 4, 100)
        assert same(got, want)


def test_route_not_taken_without_duplicates(da, small_n_route):
    rng = np.random.RandomState(11)
    seqs = duplicated_set(rng, 1, 0, 400, 15, 25)
    got, route = run(seqs, 4, 64)
    assert not route["dedup"] and route["unique"] >= 0.6 * len(seqs)
    assert same(got, oracle(seqs, 4, 64))


def test_h3n2_like_12000_properties(da):
    """the headline generator at a size the default threshold (n >= 2048) reaches: both routes agree bit for bit; symmetric, unit diagonal"""
    import torch
    from dynaalign_amd import device, synth
    import dynaalign_amd as da_
    n = 12000
    res, off = synth.h3n2_like(n, 20)
    ds = device.DeviceSequences(res, off)
    seeds = da_.hash_family_seeds(12345, 500)
    out = device.similarity_mh(ds, 4, 500, seeds)
    route = device.mh_last_route()
    assert route["dedup"] and route["unique"] == len(set(synth.to_strings(res, off)))
    os.environ["DYNAALIGN_MH_NO_DEDUP"] = "1"
    try:
        direct = device.similarity_mh(ds, 4, 500, seeds)
    finally:
        del os.environ["DYNAALIGN_MH_NO_DEDUP"]
    assert torch.equal(out.view(torch.int64), direct.view(torch.int64))
    assert torch.equal(out, out.T) and bool((out.diagonal() == 1.0).all())


@pytest.mark.parametrize("pool,n,n_hash,env", [(3000, 8000, 100, {"DYNAALIGN_MH_PIPE_STEP": "1"}),
                                               (3000, 8000, 100, {"DYNAALIGN_MH_PIPE_STEP": "1", "DYNAALIGN_MH_PIPE_ONE_STREAM": "1"}),
                                               (4200, 9000, 64, {"DYNAALIGN_MH_PIPE_WG": "1"}), (2500, 5000, 33, {"DYNAALIGN_MH_PIPE_WG": "4", "DYNAALIGN_MH_PIPE_STEP": "2"}),
                                               (2000, 2600, 500, {"DYNAALIGN_MH_PIPE_STEP": "1"}),
                                               (3000, 8000, 600, {"DYNAALIGN_MH_PIPE_STEP": "1"})])     # 10-bit counts: the unpacked LDS row, one K2 ring beside it
def test_pipelined_form_equals_one_stream_form_and_oracle(da, monkeypatch, pool, n, n_hash, env):
    """api.cpp "PIPELINED form": the unique table compared band by band on a side stream while finished output row bands are gathered
    and expanded -- several chunks (U > 1024), every schedule switch; bit-identical to the one-stream form and the oracle
    (reference src/minHash.cpp:119-188)"""
    from dynaalign_amd import synth
    rng = np.random.RandomState(pool + n)
    base = sorted(set(synth.to_strings(*synth.h3n2_like(pool, 20))))
    seqs = [base[i] for i in rng.randint(0, len(base), n)]
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    monkeypatch.setenv("DYNAALIGN_PLANE_BITS", "12")          # sets this small get 8 code planes by themselves; the banded compare is the 12-plane kernel
    monkeypatch.setenv("DYNAALIGN_MH_EXPAND", "pipe")
    out, route = run(seqs, 4, n_hash)
    assert route["dedup"] and route["pipelined"] and route["unique"] == len(set(seqs)) and route["unique"] > 1024 and route["chunks"] >= 2
    monkeypatch.setenv("DYNAALIGN_MH_EXPAND", "tiles")
    one, route1 = run(seqs, 4, n_hash)
    assert route1["dedup"] and route1["expansion"] == "tiles"
    assert same(out, one)
    monkeypatch.setenv("DYNAALIGN_MH_EXPAND", "rows")
    rows, route2 = run(seqs, 4, n_hash)
    assert route2["dedup"] and route2["expansion"] == "rows"
    assert same(out, rows)
    monkeypatch.delenv("DYNAALIGN_MH_EXPAND")                 # the default: the row expansion pipelined with K2
    monkeypatch.setenv("DYNAALIGN_MH_PIPE_HEAD", "1")
    rp, route3 = run(seqs, 4, n_hash)
    assert route3["dedup"] and route3["expansion"] == "rows, pipelined" and route3["chunks"] >= 2
    assert same(out, rp)
    assert same(out, oracle(seqs, 4, n_hash))


def test_pipelined_forms_on_a_caller_stream_and_from_two_host_threads(da, monkeypatch):
    """the pipelined forms borrow side streams and events from a per-device pool: (a) the caller's stream is a non-default torch stream with
    work queued ahead of the call, (b) two host threads call da_dev_similarity_mh on the same device at once with different inputs"""
    import threading
    import torch
    from dynaalign_amd import device, synth
    import dynaalign_amd as da_
    monkeypatch.setenv("DYNAALIGN_PLANE_BITS", "12")
    monkeypatch.setenv("DYNAALIGN_MH_PIPE_HEAD", "1")
    monkeypatch.setenv("DYNAALIGN_MH_PIPE_STEP", "1")
    sets = []
    for seed in (1, 2):
        rng = np.random.RandomState(seed)
        base = sorted(set(synth.to_strings(*synth.h3n2_like(2600 + 400 * seed, 20))))
        seqs = [base[i] for i in rng.randint(0, len(base), 6000 + 2 * seed)]
        res, off = O.pack(seqs)
        sets.append((seqs, device.DeviceSequences(np.asarray(res, np.uint8), np.asarray(off, np.int64))))
    seeds = da_.hash_family_seeds(12345, 96)
    want = [oracle(seqs, 4, 96) for seqs, _ in sets]
    # (a)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        junk = torch.zeros(64 << 20, device="cuda")
        for _ in range(20):
            junk.add_(1.0)                                       # the call's first kernel has to queue behind these
        out = device.similarity_mh(sets[0][1], 4, 96, seeds)
        route = device.mh_last_route()
    side.synchronize()
    assert route["expansion"] == "rows, pipelined" and route["chunks"] >= 2
    assert same(out.cpu().numpy(), want[0])
    # (b)
    results, errors = [None, None], []

    def work(t):
        try:
            for _ in range(3):
                o = device.similarity_mh(sets[t][1], 4, 96, seeds)
                torch.cuda.synchronize()
                results[t] = (o.cpu().numpy(), device.mh_last_route())
        except Exception as e:                                  # pragma: no cover
            errors.append(e)
    threads = [threading.Thread(target=work, args=(t,)) for t in (0, 1)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors
    for t in (0, 1):
        assert results[t][1]["expansion"] == "rows, pipelined"
        assert same(results[t][0], want[t])


def test_pipelined_form_needs_two_stages_of_twelve_planes(da, monkeypatch):
    """n_hash <= 32 (a single stage: the persistent compare does not take it) stays with the one-stream form"""
    from dynaalign_amd import synth
    base = synth.to_strings(*synth.h3n2_like(1500, 20))
    seqs = base + base[::-1]
    monkeypatch.setenv("DYNAALIGN_PLANE_BITS", "12")
    monkeypatch.setenv("DYNAALIGN_MH_EXPAND", "pipe")
    out, route = run(seqs, 4, 32)
    assert route["dedup"] and route["expansion"] == "tiles"
    assert same(out, oracle(seqs, 4, 32))


@pytest.mark.parametrize("form", ["rows", "rowspipe", "tiles", "pipe"])
def test_expansion_forms_on_awkward_sets(da, small_n_route, monkeypatch, form):
    """the expansions of the unique table (api.cpp: rows = k_expand_stream, tiles = gather + k_expand_rows, rowspipe / pipe = these pipelined with K2):
    odd n, one string with hundreds of copies (more than ES_COPIES = 4 per work item) next to single-copy strings, empty and shorter-than-k
    strings, n_hash not a multiple of 32 -- each form against the oracle (reference src/minHash.cpp:119-188)"""
    rng = np.random.RandomState(17)
    singles = duplicated_set(rng, 1, 0, 140, 15, 30)
    seqs = singles[:70] + [singles[3]] * 333 + ["", "AC", "ACD"] * 7 + singles[70:] + [singles[100]] * 5 + [singles[101]] * 4 + [singles[102]] * 6
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]
    assert len(seqs) % 2 == 1
    import torch
    from dynaalign_amd import device
    import dynaalign_amd as da_
    monkeypatch.setenv("DYNAALIGN_MH_EXPAND", form)
    monkeypatch.setenv("DYNAALIGN_PLANE_BITS", "12")          # (the pipelined forms need the 12-plane compare; a set this small gets 8 planes by itself)
    n = len(seqs)
    res, off = O.pack(seqs)
    ds = device.DeviceSequences(np.asarray(res, np.uint8), np.asarray(off, np.int64))
    for n_hash in (70, 500, 640):                              # (640: counts need 10 bits -- the row expansion's unpacked LDS row)
        buf = torch.full((n, n + 1), -1.0, dtype=torch.float64, device="cuda")     # odd n: an even leading dimension for the 16-byte stores
        device.similarity_mh(ds, 4, n_hash, da_.hash_family_seeds(12345, n_hash), out=buf[:, :n])
        route = device.mh_last_route()
        assert route["dedup"] and route["unique"] == len(set(seqs))
        assert route["expansion"] == {"rows": "rows", "rowspipe": "rows, pipelined", "tiles": "tiles", "pipe": "tiles, pipelined"}[form]
        assert same(buf[:, :n].cpu().numpy(), oracle(seqs, 4, n_hash)) and bool((buf[:, n] == -1.0).all())


@pytest.mark.parametrize("case", range(16))
def test_expansion_forms_on_random_sets(da, small_n_route, monkeypatch, case):
    """seeded random sets through every form of the duplicate route's expansion against the oracle (reference src/minHash.cpp:119-188): random
    size (even and odd), duplicate share, copy distribution (a few heavy strings), ragged lengths, and n_hash at the switches of the row
    expansion -- 33 (two stages: the banded compare's minimum), 511 / 512 (packed / unpacked LDS row), 2047 (its maximum)"""
    import torch
    from dynaalign_amd import device
    import dynaalign_amd as da_
    rng = np.random.RandomState(1000 + case)
    n_hash = [33, 64, 511, 512, 513, 2047, 100, 500][case % 8]
    n = int(rng.randint(300, 1400 if n_hash > 600 else 2600))
    n_unique = int(n * rng.uniform(0.15, 0.6))
    alpha = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    pool = sorted({"".join(map(chr, alpha[rng.randint(0, 20, rng.randint(0, 33))])) for _ in range(n_unique)})
    heavy = rng.randint(0, len(pool), 3)
    picks = np.where(rng.uniform(size=n) < 0.25, heavy[rng.randint(0, 3, n)], rng.randint(0, len(pool), n))
    seqs = [pool[i] for i in picks]
    if len(set(seqs)) * 100 > n * 60:
        seqs = seqs[: n // 2] * 2                                   # keep the duplicate route's own rule satisfied
        n = len(seqs)
    res, off = O.pack(seqs)
    ds = device.DeviceSequences(np.asarray(res, np.uint8), np.asarray(off, np.int64))
    seeds = da_.hash_family_seeds(4242 + case, n_hash)
    want = oracle(seqs, 3, n_hash, seed=4242 + case)
    monkeypatch.setenv("DYNAALIGN_PLANE_BITS", "12")
    monkeypatch.setenv("DYNAALIGN_MH_PIPE_HEAD", "1")
    monkeypatch.setenv("DYNAALIGN_MH_PIPE_STEP", "1")
    ld = n + (n & 1)
    for form in ("rowspipe", "rows", "pipe", "tiles"):
        monkeypatch.setenv("DYNAALIGN_MH_EXPAND", form)
        buf = torch.full((n, ld + 2), -1.0, dtype=torch.float64, device="cuda")
        device.similarity_mh(ds, 3, n_hash, seeds, out=buf[:, :n])
        route = device.mh_last_route()
        assert route["dedup"] and route["unique"] == len(set(seqs)), (form, route)
        assert route["expansion"].startswith("rows" if form.startswith("rows") else "tiles"), (form, route)
        assert same(buf[:, :n].cpu().numpy(), want), (form, n, n_hash)
        assert bool((buf[:, n:] == -1.0).all())


def test_row_expansion_quotient_equals_the_divide_for_every_count(da):
    """k_expand_stream forms count / n_hash as RN(q0 + (c - q0 n) r), r = RN(1 / n), q0 = RN(c r) (minhash_kernels.hip es_ratio) instead of
    dividing (src/minHash.cpp:174): the same double for EVERY 0 <= c <= n_hash, n_hash = 1 .. 2047 and a few larger -- run on the device"""
    import ctypes
    import torch
    from dynaalign_amd import _capi
    lib = _capi.load()
    lib.da_debug_ratio_check.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.da_debug_ratio_check.restype = ctypes.c_int
    sizes = list(range(1, 2048)) + [4096, 9999, 65535]
    buf = torch.zeros(sum(s_ + 1 for s_ in sizes), dtype=torch.float64, device="cuda")
    pos = 0
    for n_hash in sizes:
        assert lib.da_debug_ratio_check(n_hash, buf.data_ptr() + 8 * pos, None) == 0
        pos += n_hash + 1
    got = buf.cpu().numpy()
    want = np.concatenate([np.arange(s_ + 1, dtype=np.float64) / float(s_) for s_ in sizes])
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_row_expansion_needs_an_even_leading_dimension(da, small_n_route):
    """16-byte stores: an odd ld or an unaligned result takes the tile forms (which have their own element-wise fallback)"""
    import torch
    from dynaalign_amd import device
    import dynaalign_amd as da_
    rng = np.random.RandomState(5)
    seqs = duplicated_set(rng, 50, 400, 80, 12, 25)
    res, off = O.pack(seqs)
    ds = device.DeviceSequences(np.asarray(res, np.uint8), np.asarray(off, np.int64))
    n = len(seqs)
    seeds = da_.hash_family_seeds(12345, 64)
    buf = torch.full((n, n + 1), -1.0, dtype=torch.float64, device="cuda")
    device.similarity_mh(ds, 4, 64, seeds, out=buf[:, :n])
    route = device.mh_last_route()
    assert route["dedup"] and not route["expansion"].startswith("rows")
    assert same(buf[:, :n].cpu().numpy(), oracle(seqs, 4, 64)) and bool((buf[:, n] == -1.0).all())
    buf2 = torch.full((n, n + 2), -1.0, dtype=torch.float64, device="cuda")
    device.similarity_mh(ds, 4, 64, seeds, out=buf2[:, :n])
    assert device.mh_last_route()["expansion"] in ("rows", "rows, pipelined")     # (8-plane dictionaries: the banded compare exists since round 4)
    assert same(buf2[:, :n].cpu().numpy(), oracle(seqs, 4, 64)) and bool((buf2[:, n:] == -1.0).all())


def test_degenerate_unique_tables(da, small_n_route):
    """U = 1 (every sequence identical), U = 2, and a set whose duplicates all sit at the end / the start"""
    a, b = "ACDEFGHIKLMNPQRSTVWY", "YWVTSRQPNMLKIHGFEDCA"
    rng = np.random.RandomState(3)
    singles = duplicated_set(rng, 1, 0, 300, 18, 24)
    for seqs in ([a] * 300, [a] * 257 + [b] * 3, [a, b] * 200, singles + [singles[0]] * 320, [singles[5]] * 330 + singles):
        got, route = run(seqs, 4, 96)
        assert route["dedup"] and route["unique"] == len(set(seqs))
        assert same(got, oracle(seqs, 4, 96))


@pytest.mark.parametrize("n", [255, 256, 257, 383, 384, 385, 511])
def test_sizes_around_the_tile_edges(da, small_n_route, n):
    """n < 256: no interior tile exists, the route is not taken (direct kernels); n = 256 / 384: all tiles full; +-1: border column / row"""
    rng = np.random.RandomState(n)
    seqs = duplicated_set(rng, 40, n - 60, 60, 10, 22)
    got, route = run(seqs, 3, 40)
    assert route["dedup"] == (n >= 256)
    assert same(got, oracle(seqs, 3, 40))


def test_odd_leading_dimension_and_unaligned_output(da, small_n_route):
    """the wide-store kernels need an even ld and a 16-byte aligned matrix; anything else takes the scalar kernels -- same values"""
    import torch
    from dynaalign_amd import device
    import dynaalign_amd as da_
    rng = np.random.RandomState(21)
    seqs = duplicated_set(rng, 60, 500, 100, 10, 22)
    n = len(seqs)
    want = oracle(seqs, 4, 64)
    res, off = O.pack(seqs)
    ds = device.DeviceSequences(res, off)
    seeds = da_.hash_family_seeds(12345, 64)
    for ld, shift in ((n + 1, 0), (n + 2, 1), (n, 1)):
        buf = torch.zeros(n * ld + 8, dtype=torch.float64, device="cuda")
        out = buf[shift:shift + n * ld].view(n, ld)[:, :n]
        device.similarity_mh(ds, 4, 64, seeds, out=out)
        torch.cuda.synchronize()
        assert device.mh_last_route()["dedup"]
        assert same(out.cpu().numpy(), want)


@pytest.mark.parametrize("n_draw,n_single", [(400, 100), (900, 401)])
def test_threshold_and_edges_through_the_row_map(da, n_draw, n_single):
    """SURVEY 8(f)-1 on the duplicate route: histogram + edge extraction read the n x n count matrix through the plan's row map
    (da_dev_unique_rows + *_rows calls) -- identical histogram and identical edge set to the dense uint16 matrix"""
    import torch
    from dynaalign_amd import device, _capi
    import dynaalign_amd as da_
    rng = np.random.RandomState(n_draw)
    seqs = duplicated_set(rng, 80, n_draw, n_single, 10, 22)
    n, n_hash, k = len(seqs), 120, 3
    res, off = O.pack(seqs)
    ds = device.DeviceSequences(res, off)
    seeds = da_.hash_family_seeds(12345, n_hash)
    # dense reference on the device: all n rows
    _, planes = device.minhash_signatures(ds, k, n_hash, seeds)
    dense = device.mh_compare(planes, n, n_hash, kind=_capi.DA_OUT_COMPACT)
    h_dense = device.upper_histogram(dense, n, n_hash + 1).cpu().numpy()
    # duplicate route
    up = device.UniquePlan(ds.residues, ds.offsets, ds.n, ds.total)
    from dynaalign_amd.sharding import UniqueSequences
    _, uplanes = device.minhash_signatures(UniqueSequences(up, ds.total, ds.max_len), k, n_hash, seeds)
    table = device.unique_table(uplanes, up.unique, n_hash)
    rows = device.unique_rows(table, up)
    h_rows = device.upper_histogram_rows(rows, up, n_hash + 1).cpu().numpy()
    assert np.array_equal(h_rows, h_dense)
    cs = np.cumsum(h_dense[::-1])[::-1]
    thr_bin = int(np.argmax(cs <= 0.2 * cs[0])) or 1
    keep = np.zeros(n_hash + 1, np.uint8)
    keep[max(thr_bin, 1):] = 1
    cap = int(h_dense[keep != 0].sum()) + n
    got = device.extract_edges_rows(rows, up, keep, cap)
    want = device.extract_edges(dense, n, keep, cap)
    def as_set(t):
        c = int(t[3].item())
        a = torch.stack([t[0][:c].long(), t[1][:c].long(), t[2][:c].long() & 0xFFFF], 1).cpu().numpy()
        return c, a[np.lexsort((a[:, 1], a[:, 0]))]
    cg, ag = as_set(got)
    cw, aw = as_set(want)
    assert cg == cw == cap and np.array_equal(ag, aw)


# ---- the SPARSE route (round 3): inputs without duplicates whose signatures rarely agree -----------------------------------------

def _rand_peptides(n, lo, hi, seed, alphabet=b"ACDEFGHIKLMNPQRSTVWY"):
    rng = np.random.RandomState(seed)
    alpha = np.frombuffer(alphabet, np.uint8)
    return ["".join(map(chr, alpha[rng.randint(0, len(alpha), rng.randint(lo, hi + 1))])) for _ in range(n)]


@pytest.mark.parametrize("n,lo,hi,k,n_hash", [(2048, 20, 20, 4, 500), (2500, 12, 30, 4, 500), (4099, 20, 20, 4, 33), (3000, 8, 24, 5, 128),
                                              (2600, 20, 20, 4, 2047)])
def test_sparse_route_equals_the_oracle(da, n, lo, hi, k, n_hash):
    """da_dev_similarity_mh on duplicate-free random peptides takes the sparse route (matching incidences enumerated from the dictionary
    codes, bucketed per output tile, every tile written once): bit-identical to the oracle -- n a multiple of 128 or not (border tiles),
    ragged lengths incl. sequences shorter than k (their all-ones signatures agree with each other in EVERY column: a large class),
    n_hash from one stage to the table limit"""
    import torch
    from dynaalign_amd import device
    seqs = _rand_peptides(n, lo, hi, seed=n + n_hash)
    for t in (5, 77, 1234):
        seqs[t] = seqs[t][:max(k - 1, 0)]                     # shorter than k: no k-mers (src/minHash.cpp:98-103)
    res, off = O.pack(seqs)
    ds = device.DeviceSequences(res, off)
    seeds = da.hash_family_seeds(12345, n_hash)
    out = device.similarity_mh(ds, k, n_hash, seeds)
    torch.cuda.synchronize()
    route = device.mh_last_route()
    assert route["sparse"] and not route["dedup"], route
    rc, want = O.similarity_mh(seqs, k, n_hash, seeds)
    assert rc == 0 and np.array_equal(out.cpu().numpy().view(np.uint64), want.view(np.uint64))
    assert route["sparse_pairs"] == int(np.triu(np.round(want * n_hash).astype(np.int64), 1).sum())


def test_sparse_route_admission_rule_and_switch(da, monkeypatch):
    """dense similarity structure (k = 1: every sequence shares letters with every other) is refused by the route's admission rule;
    DYNAALIGN_MH_SPARSE_MAX_PAIRS / DYNAALIGN_MH_NO_SPARSE steer it; every way the same matrix"""
    import torch
    from dynaalign_amd import device
    seqs = _rand_peptides(2300, 20, 20, seed=3)
    res, off = O.pack(seqs)
    ds = device.DeviceSequences(res, off)
    seeds = da.hash_family_seeds(7, 200)
    a = device.similarity_mh(ds, 1, 200, seeds)
    torch.cuda.synchronize()
    assert not device.mh_last_route()["sparse"]                # k = 1: 2300 x 2300 x 200 incidences, classes of hundreds
    rc, want = O.similarity_mh(seqs, 1, 200, seeds)
    assert np.array_equal(a.cpu().numpy().view(np.uint64), want.view(np.uint64))
    b = device.similarity_mh(ds, 4, 200, seeds)
    torch.cuda.synchronize()
    assert device.mh_last_route()["sparse"]
    monkeypatch.setenv("DYNAALIGN_MH_SPARSE_MAX_PAIRS", "10")
    c = device.similarity_mh(ds, 4, 200, seeds)
    torch.cuda.synchronize()
    assert not device.mh_last_route()["sparse"]
    monkeypatch.delenv("DYNAALIGN_MH_SPARSE_MAX_PAIRS")
    monkeypatch.setenv("DYNAALIGN_MH_NO_SPARSE", "1")
    d = device.similarity_mh(ds, 4, 200, seeds)
    torch.cuda.synchronize()
    assert not device.mh_last_route()["sparse"]
    assert torch.equal(b.view(torch.int64), c.view(torch.int64)) and torch.equal(b.view(torch.int64), d.view(torch.int64))
