"""Full-size parity on ONE MI355X: BASELINE configs[3] (100 000 h3n2-like 20-mers) and its uniform twin, nothing sampled
where the oracle can afford it, and the N > 1 legs of the sharded path rehearsed with virtual ranks at N = 100 000.

  * every one of the 100 000 signature rows against the oracle (src/minHash.cpp:140-157);
  * thousands of whole rows of the match-count matrix against the oracle's own compare loop run on the oracle's own
    signatures (src/minHash.cpp:160-178) -- the GPU's signatures are not an input of the expected values;
  * 128 whole rows of similarityNW against the oracle's DP + traceback (src/pairwiseSeqAlign.cpp:209-313, :340-352);
  * the NW duplicate route against the direct kernel over all 10^10 elements, bit for bit;
  * world in {2, 3, 8}: every rank's shard computed in turn on the one GPU, the concatenation (= what the ONE all-gather
    delivers) finalized, and the result compared bit for bit with the single-GPU matrix -- MinHash packed direct, MinHash
    uint16 direct, MinHash duplicate route, NW duplicate route, NW direct.  Every index of those paths is exercised where
    rows x leading dimension exceeds 2^31.

Two float64 N x N buffers (2 x 80 GB) live for the whole module; torch is the checker's plumbing (allocation, compares)."""
import os

import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu

N, N_HASH, K, SEED = 100000, 500, 4, 12345


@pytest.fixture(scope="module")
def da(built):
    import dynaalign_amd
    from dynaalign_amd import _capi
    assert _capi.load().da_device_count() > 0
    return dynaalign_amd


@pytest.fixture(scope="module")
def bufs(da):
    """the reference matrix and the matrix under test: 2 x 80 GB of the 288 GB"""
    from dynaalign_amd import _capi
    torch.cuda.empty_cache()
    _capi.load().da_release_device_memory()          # buffers the library parked for earlier tests are invisible to torch's allocator
    a = torch.empty((N, N), dtype=torch.float64, device="cuda")
    b = torch.empty((N, N), dtype=torch.float64, device="cuda")
    yield a, b
    del a, b
    torch.cuda.empty_cache()
    _capi.load().da_release_device_memory()


@pytest.fixture(scope="module")
def h3n2(da):
    from dynaalign_amd import device, synth
    res, off = synth.h3n2_like(N, 20)
    seqs = synth.to_strings(res, off)
    seeds = da.hash_family_seeds(SEED, N_HASH)
    ds = device.DeviceSequences(res, off)
    want_sig = O.signatures(seqs, K, N_HASH, seeds)            # the oracle's signatures: every expected value below derives from them
    return {"seqs": seqs, "seeds": seeds, "ds": ds, "sig": want_sig,
            "d_seeds": torch.from_numpy(seeds.view(np.int32).copy()).cuda()}


def _same_bits(a, b, step=4000):
    """float64 device matrices equal bit for bit (NaN patterns included)"""
    for r0 in range(0, a.shape[0], step):
        if not torch.equal(a[r0:r0 + step].view(torch.int64), b[r0:r0 + step].view(torch.int64)):
            bad = (a[r0:r0 + step].view(torch.int64) != b[r0:r0 + step].view(torch.int64)).nonzero()[0].tolist()
            raise AssertionError("first difference at (%d, %d): %r vs %r" % (r0 + bad[0], bad[1], a[r0 + bad[0], bad[1]].item(),
                                                                             b[r0 + bad[0], bad[1]].item()))


def _row_blocks(n, nblocks, rows):
    """row ranges spread over the matrix, tile edges (multiples of 128 +- 1) included, the last rows included"""
    starts = sorted({0, 127, n - rows} | {((n * b // nblocks) // 128) * 128 + (b % 3) - 1 for b in range(1, nblocks)})
    return [(max(0, s), min(n, max(0, s) + rows)) for s in starts]


def _check_count_rows(cnt_rows_of, sig, nblocks=24, rows=96):
    """whole rows of the GPU's count matrix == the oracle's compare loop on the oracle's signatures (the diagonal is forced)"""
    checked = 0
    for a, b in _row_blocks(sig.shape[0], nblocks, rows):
        want = O.mh_counts(sig, a, b)
        assert np.array_equal(cnt_rows_of(a, b), want), "match counts differ from the oracle in rows %d..%d" % (a, b)
        checked += b - a
    return checked


def test_every_signature_row_and_2000_count_rows_against_the_oracle(da, h3n2, bufs):
    from dynaalign_amd import device, _capi
    ds, seeds, want_sig = h3n2["ds"], h3n2["seeds"], h3n2["sig"]
    sig, planes = device.minhash_signatures(ds, K, N_HASH, seeds)
    got = sig[:, :N_HASH].cpu().numpy().view(np.uint32)
    assert got.shape == want_sig.shape and np.array_equal(got, want_sig)      # all 100 000 x 500 min-hashes
    ref, out = bufs
    device.mh_compare(planes, N, N_HASH, kind=_capi.DA_OUT_F64, out=ref)       # the direct kernels (no duplicate collapse)
    ratio = np.arange(N_HASH + 1, dtype=np.float64) / N_HASH                   # the reference's divide, on the host

    def rows_of(a, b):
        blk = ref[a:b].cpu().numpy()
        cnt = np.searchsorted(ratio, blk)                                       # exact: every element must BE one of the ratios
        assert np.array_equal(ratio[cnt], blk)
        return cnt.astype(np.uint16)
    assert _check_count_rows(rows_of, want_sig) >= 2000
    # the one-call route bench.py times (duplicates collapsed) over the whole matrix against the direct kernels
    out.fill_(-1.0)
    device.similarity_mh(ds, K, N_HASH, seeds, out=out)
    route = device.mh_last_route()
    assert route["dedup"] and route["expansion"] == "rows, pipelined" and route["unique"] == len(set(h3n2["seqs"]))
    _same_bits(out, ref)
    assert bool((torch.diagonal(out) == 1.0).all())
    # ... and the route's other forms: the row expansion after K2, the tile expansion pipelined with K2 and one kernel after the other
    for form, name in (("rows", "rows"), ("pipe", "tiles, pipelined"), ("tiles", "tiles")):
        out.fill_(-1.0)
        os.environ["DYNAALIGN_MH_EXPAND"] = form
        try:
            device.similarity_mh(ds, K, N_HASH, seeds, out=out)
        finally:
            del os.environ["DYNAALIGN_MH_EXPAND"]
        route = device.mh_last_route()
        assert route["dedup"] and route["expansion"] == name
        _same_bits(out, ref)
    # the one call without the duplicate collapse: the heavy / rare split (8 dense planes + 7.2e6 incidences from lists; round 4) and the full-width
    # compare; and the split inside the duplicate route (opt-in) -- whole matrix each
    for switches, dedup, split in (({"DYNAALIGN_MH_NO_DEDUP": "1"}, False, True), ({"DYNAALIGN_MH_NO_DEDUP": "1", "DYNAALIGN_MH_NO_HYBRID": "1"}, False, False),
                                   ({"DYNAALIGN_MH_HYBRID_DEDUP": "1"}, True, True)):
        out.fill_(-1.0)
        os.environ.update(switches)
        try:
            device.similarity_mh(ds, K, N_HASH, seeds, out=out)
        finally:
            for key in switches:
                del os.environ[key]
        route = device.mh_last_route()
        assert route["dedup"] == dedup and route["split"] == split and route["plane_bits"] == (8 if split else 12), route
        assert not split or (route["rare_pairs"] > 0 and route["plane_bits_without"] == 12)
        _same_bits(out, ref)


def test_row_expansion_with_more_than_49152_unique_strings(da, h3n2, bufs):
    """the duplicate route on a set with 51 000+ unique strings of 100 000 (k_expand_stream's deepest prefetch, <true, 8>; K2's table has 50 bands),
    and with n_hash = 640 (counts need 10 bits: the unpacked LDS row, one K2 workgroup per CU in the pipeline): rows / rows pipelined / tiles over
    the whole matrix, and whole rows of the result against the oracle's compare loop"""
    from dynaalign_amd import device, synth
    rng = np.random.RandomState(99)
    alpha = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    seqs = list(h3n2["seqs"][:88000]) + ["".join(map(chr, alpha[rng.randint(0, 20, 20)])) for _ in range(12000)]
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]
    unique = len(set(seqs))
    assert 49152 < unique <= 60000
    res, off = O.pack(seqs)
    ds = device.DeviceSequences(np.asarray(res, np.uint8), np.asarray(off, np.int64))
    ref, out = bufs
    for n_hash in (500, 640):
        seeds = da.hash_family_seeds(SEED, n_hash)
        got = {}
        for form, name in (("tiles", "tiles"), ("rows", "rows"), ("rowspipe", "rows, pipelined")):
            os.environ["DYNAALIGN_MH_EXPAND"] = form
            try:
                dst = ref if form == "tiles" else out
                dst.fill_(-1.0)
                device.similarity_mh(ds, K, n_hash, seeds, out=dst)
            finally:
                del os.environ["DYNAALIGN_MH_EXPAND"]
            route = device.mh_last_route()
            # (the added random peptides push the table's dictionaries past 12 code planes: the banded compare exists for 12 and 8 planes, so the route
            #  pipelines when the heavy / rare split brings the compare down to 8 planes -- round 4 -- and runs one kernel after the other otherwise)
            if form == "rowspipe" and route["plane_bits"] not in (8, 12):
                name = "rows"
            assert route["dedup"] and route["expansion"] == name and route["unique"] == unique
            assert route["split"] == (route["plane_bits"] == 8)
            if route["split"]:
                assert route["plane_bits_without"] > 12 and route["rare_pairs"] > 0
            if form != "tiles":
                _same_bits(out, ref)
        sig = O.signatures(seqs, K, n_hash, seeds) if n_hash == 640 else None
        if sig is not None:                                                     # whole rows against the oracle (its signatures, its compare loop)
            ratio = np.arange(n_hash + 1, dtype=np.float64) / n_hash

            def rows_of(a, b):
                blk = ref[a:b].cpu().numpy()
                cnt = np.searchsorted(ratio, blk)
                assert np.array_equal(ratio[cnt], blk)
                return cnt.astype(np.uint16)
            assert _check_count_rows(rows_of, sig, nblocks=6, rows=48) >= 200


def test_uniform_100k_count_rows_against_the_oracle(da, bufs):
    """no duplicates to collapse, 15-bit dictionary codes -> the hand-scheduled 16-plane kernel"""
    from dynaalign_amd import device, synth
    res, off = synth.uniform_peptides(N, 20, seed=7)
    seqs = synth.to_strings(res, off)
    seeds = da.hash_family_seeds(SEED, N_HASH)
    want_sig = O.signatures(seqs, K, N_HASH, seeds)
    ds = device.DeviceSequences(res, off)
    _, out = bufs
    out.fill_(-1.0)
    device.similarity_mh(ds, K, N_HASH, seeds, out=out)
    route = device.mh_last_route()
    # uniform random peptides: 1.4e8 matching (pair, hash function) incidences -> the SPARSE route (round 3)
    assert not route["dedup"] and route["sparse"] and route["sparse_pairs"] > 0
    ratio = np.arange(N_HASH + 1, dtype=np.float64) / N_HASH

    def rows_of(a, b):
        blk = out[a:b].cpu().numpy()
        cnt = np.searchsorted(ratio, blk)
        assert np.array_equal(ratio[cnt], blk)
        return cnt.astype(np.uint16)
    assert _check_count_rows(rows_of, want_sig) >= 2000
    # whole-matrix properties: symmetric, unit diagonal, and the count total that never walks the pair loop
    assert bool((torch.diagonal(out) == 1.0).all())
    for r0 in range(0, N, 10000):
        assert torch.equal(out[r0:r0 + 10000, r0:r0 + 10000], out[r0:r0 + 10000, r0:r0 + 10000].T)
    assert torch.equal(out[:5000, 60000:], out[60000:, :5000].T)
    total = 0
    for h in range(N_HASH):
        _, c = np.unique(want_sig[:, h], return_counts=True)
        total += int((c.astype(np.int64) ** 2).sum())
    # sum of all match counts, diagonal forced to n_hash: sum_h sum_v mult^2 counts the diagonal as n_hash per row too
    got = 0
    for r0 in range(0, N, 5000):
        got += int(torch.round(out[r0:r0 + 5000] * N_HASH).to(torch.int64).sum().item())
    assert got == total
    assert route["sparse_pairs"] == (total - N * N_HASH) // 2          # the route's incidence count is the same sum, off the diagonal, i < j
    # ... and the dense kernels (15-bit dictionary codes -> the hand-scheduled 16-plane kernel) over the whole matrix, bit for bit
    ref, _ = bufs
    os.environ["DYNAALIGN_MH_NO_SPARSE"] = "1"
    try:
        ref.fill_(-1.0)
        device.similarity_mh(ds, K, N_HASH, seeds, out=ref)
        r2 = device.mh_last_route()
    finally:
        del os.environ["DYNAALIGN_MH_NO_SPARSE"]
    assert not r2["sparse"] and not r2["dedup"] and r2["plane_bits"] in (14, 15, 16)
    _same_bits(out, ref)


def _virtual_mh_direct(h3n2, planes, world, packed, out):
    from dynaalign_amd import sharding
    plan0 = sharding.Plan(N, 0, world, sharding.MH_TILE)
    work = sharding.PackedWorkspace(plan0, N_HASH) if packed else sharding.Workspace(plan0)
    for r in range(world):
        plan = sharding.Plan(N, r, world, sharding.MH_TILE)
        work.local.fill_(0x7FFF)                                               # poison what the rank must not rely on
        sharding.mh_local_block(plan, work, planes, N_HASH)
        if packed:
            sharding.pack_local_block(plan, work)
            work.gathered[r * work.block_bytes:(r + 1) * work.block_bytes].copy_(work.packed)
        else:
            work.gathered[r * plan.local_rows:(r + 1) * plan.local_rows].copy_(work.local)
    assert work.gathered.numel() * work.gathered.element_size() > 2 ** 31      # the index math the test is after
    out.fill_(-1.0)
    if packed:
        sharding.finalize_shards_packed(plan0, work, work.gathered, N_HASH, out)
    else:
        sharding.finalize_shards(plan0, work.gathered, False, N_HASH, out)
    return out


@pytest.mark.parametrize("world", [2, 3, 8])
def test_mh_virtual_ranks_at_full_size(da, h3n2, bufs, world):
    """config 4's N > 1 legs, MinHash: shard compare (+ pack) per rank, concatenation, finalize == the single-GPU matrix"""
    from dynaalign_amd import device, sharding, _capi
    ref, out = bufs
    ds, seeds = h3n2["ds"], h3n2["seeds"]
    _, planes = device.minhash_signatures(ds, K, N_HASH, seeds)
    device.mh_compare(planes, N, N_HASH, kind=_capi.DA_OUT_F64, out=ref)
    _same_bits(_virtual_mh_direct(h3n2, planes, world, True, out), ref)         # 9-bit packed exchange (what bench.py --gpus N runs)
    if world != 3:
        _same_bits(_virtual_mh_direct(h3n2, planes, world, False, out), ref)    # plain uint16 blocks
    # the duplicate route on the shards: packed shards of the unique strings' table -> table -> column gather + expansion
    uplan = device.UniquePlan(ds.residues, ds.offsets, ds.n, ds.total)
    assert uplan.unique == len(set(h3n2["seqs"])) and sharding.dedup_worth(N, uplan.unique, False, N_HASH)
    blocks, last = [], None
    for r in range(world):
        plan, work = sharding.mh_unique_local(uplan, ds, K, N_HASH, h3n2["d_seeds"], r, world)
        blocks.append(work.packed.clone())
        last = (plan, work)
    out.fill_(-1.0)
    sharding.mh_unique_finish(uplan, last[0], last[1], torch.cat(blocks), N_HASH, out)
    _same_bits(out, ref)


def test_nw_128_rows_against_the_oracle_and_duplicate_route_against_direct(da, h3n2, bufs, monkeypatch):
    from dynaalign_amd import device
    ref, out = bufs
    ds, seqs = h3n2["ds"], h3n2["seqs"]
    assert int(device.nw_encode(ds).item()) == 0
    monkeypatch.setenv("DYNAALIGN_NW_NO_DEDUP", "1")
    ref.fill_(-1.0)
    device.nw(ds, out=ref)                                                      # the direct kernel: every pair's DP
    assert not device.nw_last_route()["dedup"]
    monkeypatch.delenv("DYNAALIGN_NW_NO_DEDUP")
    for a, b in _row_blocks(N, 8, 16):                                          # 128+ whole rows: the oracle's DP + traceback
        rc, mt, ln, _, _ = O.nw_rows(seqs, a, b)
        assert rc == 0
        got = ref[a:b].cpu().numpy()
        assert np.array_equal(got.view(np.uint64), (mt / ln.astype(np.float64)).view(np.uint64)), "NW rows %d..%d" % (a, b)
    out.fill_(-1.0)
    device.nw(ds, out=out)                                                      # ordered unique table + two-pass expansion
    route = device.nw_last_route()
    assert route["dedup"] and route["unique"] == len(set(seqs))
    _same_bits(out, ref)                                                        # all 10^10 elements


@pytest.mark.parametrize("world", [2, 8])
def test_nw_virtual_ranks_at_full_size(da, h3n2, bufs, world, monkeypatch):
    """config 4's N > 1 legs, NW: the direct shards (folded uint16 blocks) and the ordered unique table in cyclic row blocks"""
    from dynaalign_amd import device, sharding
    ref, out = bufs
    ds, seqs = h3n2["ds"], h3n2["seqs"]
    assert int(device.nw_encode(ds).item()) == 0
    monkeypatch.setenv("DYNAALIGN_NW_NO_DEDUP", "1")
    device.nw(ds, out=ref)
    monkeypatch.delenv("DYNAALIGN_NW_NO_DEDUP")
    # duplicate route: rank r's cyclic 128-row units of the ORDERED table, concatenated, read through table_row()
    uplan = device.UniquePlan(ds.codes, ds.offsets, ds.n, ds.total)
    assert uplan.unique == len(set(seqs))
    blocks = [sharding.nw_unique_rows_local(uplan, ds.max_len, r, world) for r in range(world)]
    out.fill_(-1.0)
    device.expand_unique(torch.cat(blocks, 0), uplan, True, 0, ds.max_len, out, table_world=world)
    del blocks
    _same_bits(out, ref)
    # direct shards
    plan0 = sharding.Plan(N, 0, world, sharding.NW_TILE)
    work = sharding.Workspace(plan0)
    for r in range(world):
        plan = sharding.Plan(N, r, world, sharding.NW_TILE)
        work.local.fill_(0x7FFF)
        sharding.nw_local_block(plan, work, ds)
        work.gathered[r * plan.local_rows:(r + 1) * plan.local_rows].copy_(work.local)
    assert work.gathered.numel() * 2 > 2 ** 31
    out.fill_(-1.0)
    sharding.finalize_shards(plan0, work.gathered, True, 0, out)
    _same_bits(out, ref)
