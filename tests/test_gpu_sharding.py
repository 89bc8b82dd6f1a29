"""GPU tests of the row-sharded path on ONE MI355X: the P ranks of a node are played one
after another on the same device (each rank's HIP shard kernel writes its local block; the
concatenation of the blocks is exactly what the RCCL all-gather would deliver), then the HIP
finalize kernel expands it.  Result must equal the oracle bit for bit on every "rank"."""
import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def da(built):
    import dynaalign_amd
    from dynaalign_amd import _capi
    assert _capi.load().da_device_count() > 0
    return dynaalign_amd


def _gather_virtual(plan_of, fill_local, world, n):
    from dynaalign_amd import sharding
    blocks = []
    for r in range(world):
        plan = plan_of(r)
        work = sharding.Workspace(plan, "cuda")
        work.local.fill_(0x7FFF)          # poison everything the kernel is not supposed to write
        fill_local(plan, work)
        blocks.append(work.local.clone())
    return torch.cat(blocks, 0)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("n", [100, 129, 700])
def test_mh_virtual_ranks(da, world, n):
    from dynaalign_amd import device, sharding, synth
    res, off = synth.h3n2_like(n, 20)
    seqs = synth.to_strings(res, off)
    n_hash = 500
    seeds = da.hash_family_seeds(12345, n_hash)
    ds = device.DeviceSequences(res, off)
    sig, planes = device.minhash_signatures(ds, 4, n_hash, seeds)
    plan_of = lambda r: sharding.Plan(n, r, world, sharding.MH_TILE)
    gathered = _gather_virtual(plan_of, lambda p, w: sharding.mh_local_block(p, w, planes, n_hash), world, n)
    out = torch.full((n, n), -1.0, dtype=torch.float64, device="cuda")
    sharding.finalize_shards(plan_of(0), gathered, False, n_hash, out)
    rc, want = O.similarity_mh(seqs, 4, n_hash, seeds)
    got = out.cpu().numpy()
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


@pytest.mark.parametrize("bits", [12, 14, 15, 16])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("n,n_hash", [(700, 500), (1300, 96), (1025, 33), (2200, 20)])
def test_mh_virtual_ranks_hand_scheduled_shard_kernel(da, monkeypatch, world, n, n_hash, bits):
    """k_mh_compare_s12 / k_mh_compare_s16<14 | 15 | 16> (round 4): the hand-scheduled plane loops behind the shard geometry (cyclic 128-row
    units, folded rows, upper tiles only) -- the code-plane count forced (sets this small get 8), interior tiles by them, border tiles by the
    compiled kernel; every rank's block is bit-identical to the compiled kernel's (poisoned elsewhere), and the finalized matrix to the
    oracle (reference src/minHash.cpp:119-188)"""
    import os
    from dynaalign_amd import device, sharding, synth
    res, off = synth.h3n2_like(n, 20)
    seqs = synth.to_strings(res, off)
    seeds = da.hash_family_seeds(12345, n_hash)
    ds = device.DeviceSequences(res, off)
    monkeypatch.setenv("DYNAALIGN_PLANE_BITS", str(bits))
    sig, planes = device.minhash_signatures(ds, 4, n_hash, seeds)
    assert planes.bits == bits
    plan_of = lambda r: sharding.Plan(n, r, world, sharding.MH_TILE)
    gathered = _gather_virtual(plan_of, lambda p, w: sharding.mh_local_block(p, w, planes, n_hash), world, n)
    monkeypatch.setenv("DYNAALIGN_K2_NO_ASM", "1")
    compiled = _gather_virtual(plan_of, lambda p, w: sharding.mh_local_block(p, w, planes, n_hash), world, n)
    monkeypatch.delenv("DYNAALIGN_K2_NO_ASM")
    assert torch.equal(gathered, compiled)
    out = torch.full((n, n), -1.0, dtype=torch.float64, device="cuda")
    sharding.finalize_shards(plan_of(0), gathered, False, n_hash, out)
    rc, want = O.similarity_mh(seqs, 4, n_hash, seeds)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want.view(np.uint64))


@pytest.mark.parametrize("world", [1, 2, 5, 8])
@pytest.mark.parametrize("n", [63, 130, 400])
def test_nw_virtual_ranks(da, world, n):
    from dynaalign_amd import device, sharding, synth
    res, off = synth.h3n2_like(n, 20)
    seqs = synth.to_strings(res, off)
    ds = device.DeviceSequences(res, off)
    assert int(device.nw_encode(ds).item()) == 0
    plan_of = lambda r: sharding.Plan(n, r, world, sharding.NW_TILE)
    gathered = _gather_virtual(plan_of, lambda p, w: sharding.nw_local_block(p, w, ds), world, n)
    out = torch.full((n, n), -1.0, dtype=torch.float64, device="cuda")
    sharding.finalize_shards(plan_of(0), gathered, True, 0, out)
    rc, want, _ = O.similarity_nw(seqs)
    got = out.cpu().numpy()
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_sharded_step_world1_equals_fused_path(da):
    """world = 1 through the sharded API (no process group needed) == the fused symmetric kernel"""
    from dynaalign_amd import device, sharding, synth, _capi
    n, n_hash = 1000, 500
    res, off = synth.h3n2_like(n, 20)
    ds = device.DeviceSequences(res, off)
    sig, planes = device.minhash_signatures(ds, 4, n_hash, da.hash_family_seeds(12345, n_hash))
    fused = device.mh_compare(planes, n, n_hash)
    plan = sharding.Plan(n, 0, 1)
    out = torch.empty((n, n), dtype=torch.float64, device="cuda")
    sharding.mh_sharded_step(plan, sharding.Workspace(plan), planes, n_hash, out)
    assert torch.equal(out, fused)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_mh_edges_sharded_virtual_ranks(da, world):
    """threshold + sparsify on the shards: sum of the ranks' histograms = dense histogram, union of the
    ranks' edge lists = the dense threshold step (R/clusterbreak.R:219-221) on the oracle matrix"""
    from dynaalign_amd import device, sharding, synth
    n, n_hash, p = 900, 500, 0.8
    res, off = synth.h3n2_like(n, 20)
    seqs = synth.to_strings(res, off)
    seeds = da.hash_family_seeds(12345, n_hash)
    ds = device.DeviceSequences(res, off)
    sig, planes = device.minhash_signatures(ds, 4, n_hash, seeds)
    plans = [sharding.Plan(n, r, world, sharding.MH_TILE) for r in range(world)]
    works = [sharding.Workspace(pl) for pl in plans]
    hists = []
    for pl, w in zip(plans, works):
        w.local.fill_(0x7FFF)
        sharding.mh_local_block(pl, w, planes, n_hash)
        hists.append(sharding.shard_histogram(pl, w.local, n_hash + 1))
    total = torch.stack(hists).sum(0)
    rc, M = O.similarity_mh(seqs, 4, n_hash, seeds)
    cnt = np.round(M * n_hash).astype(np.int64)
    want_hist = np.bincount(cnt[np.triu_indices(n, 1)], minlength=n_hash + 1)
    assert np.array_equal(total.cpu().numpy(), want_hist)
    edges = set()
    thr_seen = set()
    for pl, w, h in zip(plans, works, hists):
        thr, ei, ej, ev, c, cap = sharding.edges_from_histograms(
            pl, h, n_hash, p, lambda t: t.copy_(total),
            lambda keep, capacity: sharding.shard_extract_edges(pl, w.local, keep, capacity))
        m = int(c.item())
        assert m == cap
        thr_seen.add(thr)
        for a, b, v in zip(ei[:m].cpu().numpy(), ej[:m].cpu().numpy(), ev[:m].cpu().numpy().view(np.uint16)):
            key = (int(a), int(b))
            assert key not in edges and pl.owner(int(a)) == pl.rank
            edges.add(key)
            assert v == cnt[a, b]
    thr_d, i_d, j_d, w_d = da.similarityMH_edges(seqs, 4, n_hash, p, seed=12345)
    assert thr_seen == {thr_d}
    assert edges == set(zip(i_d.tolist(), j_d.tolist()))


def test_symmetrize_and_widen_helpers(da):
    """da_dev_symmetrize (lower <- upper) and da_dev_widen on small matrices of both element kinds"""
    from dynaalign_amd import device, _capi
    rng = np.random.RandomState(4)
    for n in (1, 31, 32, 33, 200):
        a = rng.randint(0, 500, (n, n)).astype(np.int16)
        t = torch.from_numpy(a.copy()).cuda()
        device.symmetrize(t, n, _capi.DA_OUT_COMPACT)
        want = np.triu(a) + np.triu(a, 1).T
        assert np.array_equal(t.cpu().numpy(), want)
        f = torch.from_numpy(rng.rand(n, n)).cuda()
        g = f.clone()
        device.symmetrize(g, n, _capi.DA_OUT_F64)
        fw = f.cpu().numpy()
        assert np.array_equal(g.cpu().numpy(), np.triu(fw) + np.triu(fw, 1).T)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("n,n_hash", [(129, 500), (700, 500), (700, 200), (300, 1000)])
def test_mh_packed_exchange_virtual_ranks(da, world, n, n_hash):
    """the MH exchange in bits(n_hash) bits per count: every rank's block packed by the HIP kernel, the
    concatenation (= what the all-gather delivers) expanded by the packed finalize kernel == the oracle"""
    from dynaalign_amd import device, sharding, synth
    res, off = synth.h3n2_like(n, 20)
    seqs = synth.to_strings(res, off)
    seeds = da.hash_family_seeds(12345, n_hash)
    ds = device.DeviceSequences(res, off)
    sig, planes = device.minhash_signatures(ds, 4, n_hash, seeds)
    blocks = []
    for r in range(world):
        plan = sharding.Plan(n, r, world, sharding.MH_TILE)
        work = sharding.PackedWorkspace(plan, n_hash)
        assert work.bits == (8 if n_hash < 256 else 9 if n_hash < 512 else 10)
        work.local.fill_(0x7FFF)
        sharding.mh_local_block(plan, work, planes, n_hash)
        blocks.append(sharding.pack_local_block(plan, work).clone())
    gathered = torch.cat(blocks)
    assert gathered.numel() == world * work.block_bytes < world * plan.local_rows * plan.width * 2
    out = torch.full((n, n), -1.0, dtype=torch.float64, device="cuda")
    sharding.finalize_shards_packed(plan, work, gathered, n_hash, out)
    rc, want = O.similarity_mh(seqs, 4, n_hash, seeds)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want.view(np.uint64))
    if world == 1:
        out2 = torch.empty_like(out)
        sharding.mh_sharded_step_packed(plan, work, planes, n_hash, out2)
        assert torch.equal(out, out2)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("n,lens", [(700, (20, 20)), (333, (1, 40))])
def test_nw_edges_sharded_virtual_ranks(da, world, n, lens):
    """the NW twin of the sharded MinHash edge list: the ranks' code histograms sum to the dense histogram, the union of
    their edge lists is the dense threshold step on the oracle's similarityNW matrix (quantile over RATIOS: codes such as
    1/2 and 2/4 are different bins of equal value)"""
    from dynaalign_amd import device, sharding, synth
    from test_threshold_edges import reference_edges
    rng = np.random.RandomState(n + world)
    if lens[0] == lens[1]:
        res, off = synth.h3n2_like(n, lens[0])
        seqs = synth.to_strings(res, off)
    else:
        alpha = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", np.uint8)
        seqs = ["".join(map(chr, alpha[rng.randint(0, 20, rng.randint(lens[0], lens[1] + 1))])) for _ in range(n)]
        res, off = da.pack_sequences(seqs)
    ds = device.DeviceSequences(res, off)
    assert int(device.nw_encode(ds).item()) == 0
    rc, M, _ = O.similarity_nw(seqs)
    assert rc == 0
    thr_w, iw, jw, ww = reference_edges(M, 0.8)
    plans = [sharding.Plan(n, r, world, sharding.NW_TILE) for r in range(world)]
    works = [sharding.Workspace(pl) for pl in plans]
    values = sharding.nw_code_values(ds.max_len)
    hists = []
    for pl, w in zip(plans, works):
        w.local.fill_(0x7FFF)
        sharding.nw_local_block(pl, w, ds)
        hists.append(sharding.shard_histogram(pl, w.local, len(values)))
    total = torch.stack(hists).sum(0)
    assert int(total.sum().item()) == n * (n - 1) // 2
    edges = {}
    for pl, w, h in zip(plans, works, hists):
        thr, ei, ej, ev, c, cap = sharding.edges_from_histograms(
            pl, h, 0, 0.8, lambda t: t.copy_(total),
            lambda keep, capacity: sharding.shard_extract_edges(pl, w.local, keep, capacity), values=values)
        m = int(c.item())
        assert m == cap and thr == thr_w
        for a, b, v in zip(ei[:m].cpu().numpy(), ej[:m].cpu().numpy(), ev[:m].cpu().numpy().view(np.uint16)):
            key = (int(a), int(b))
            assert key not in edges and pl.owner(int(a)) == pl.rank
            edges[key] = values[v]
    assert sorted(edges) == list(zip(iw.tolist(), jw.tolist()))
    assert [edges[k] for k in sorted(edges)] == ww.tolist()
    if world == 1:                                             # the one-call form agrees
        thr, ei, ej, ev, c, vals = sharding.nw_edges_sharded(plans[0], works[0], ds, 0.8)
        assert thr == thr_w and int(c.item()) == len(iw)


def _dup_set(n_pool, n_draw, n_single, seed, lo=12, hi=24):
    rng = np.random.RandomState(seed)
    alpha = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    mk = lambda: "".join(map(chr, alpha[rng.randint(0, 20, rng.randint(lo, hi))]))
    pool = [mk() for _ in range(n_pool)]
    seqs = [pool[q] for q in rng.randint(0, n_pool, n_draw)] + [mk() for _ in range(n_single)]
    rng.shuffle(seqs)
    return seqs


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("n_draw,n_single", [(500, 140), (900, 380)])
def test_mh_duplicate_route_virtual_ranks(da, world, n_draw, n_single):
    """sharded similarityMH with the duplicates collapsed: every rank's packed shard of the UNIQUE strings' count table, their
    concatenation (= the all-gather), table rebuild + index expansion; bit-identical to the oracle on the full input"""
    from dynaalign_amd import device, sharding
    seqs = _dup_set(90, n_draw, n_single, world * 7 + n_draw)
    n, n_hash = len(seqs), 200
    res, off = O.pack(seqs)
    ds = device.DeviceSequences(res, off)
    seeds = da.hash_family_seeds(12345, n_hash)
    d_seeds = torch.from_numpy(seeds.view(np.int32).copy()).cuda()
    uplan = device.UniquePlan(ds.residues, ds.offsets, ds.n, ds.total)
    assert uplan.unique == len(set(seqs)) and sharding.dedup_worth(n, uplan.unique, False, n_hash, min_n=1)
    blocks, last = [], None
    for r in range(world):
        plan, work = sharding.mh_unique_local(uplan, ds, 4, n_hash, d_seeds, r, world)
        blocks.append(work.packed.clone())
        last = (plan, work)
    gathered = torch.cat(blocks)
    out = torch.full((n, n), -1.0, dtype=torch.float64, device="cuda")
    sharding.mh_unique_finish(uplan, last[0], last[1], gathered, n_hash, out)
    rc, want = O.similarity_mh(seqs, 4, n_hash, seeds)
    assert rc == 0 and np.array_equal(out.cpu().numpy().view(np.uint64), want.view(np.uint64))
    if world == 1:                                             # the step function itself (no collective at world 1)
        out2 = torch.full((n, n), -1.0, dtype=torch.float64, device="cuda")
        sharding.mh_sharded_step_dedup(uplan, ds, 4, n_hash, d_seeds, 0, 1, out2)
        assert torch.equal(out2.view(torch.int64), out.view(torch.int64))


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("n_draw,n_single", [(500, 140), (900, 380)])
def test_nw_duplicate_route_virtual_ranks(da, world, n_draw, n_single):
    """sharded similarityNW with the duplicates collapsed: row blocks of cyclic 128-row units of the ORDERED unique table, their
    concatenation, index expansion; the input holds both orders of asymmetric pairs (random order of a small pool)"""
    from dynaalign_amd import device, sharding
    seqs = _dup_set(90, n_draw, n_single, world * 11 + n_single) + ["YDYIHIYADKQDRIGWLGNT", "MYCEMNVEIQYMATKNMWNT"] * 3
    n = len(seqs)
    res, off = O.pack(seqs)
    ds = device.DeviceSequences(res, off)
    assert int(device.nw_encode(ds).item()) == 0
    uplan = device.UniquePlan(ds.codes, ds.offsets, ds.n, ds.total)
    assert uplan.unique == len(set(seqs)) and sharding.dedup_worth(n, uplan.unique, True, 0, ds.max_len, min_n=1)
    blocks = [sharding.nw_unique_rows_local(uplan, ds.max_len, r, world) for r in range(world)]
    out = torch.full((n, n), -1.0, dtype=torch.float64, device="cuda")
    device.expand_unique(torch.cat(blocks, 0), uplan, True, 0, ds.max_len, out, table_world=world)
    rc, want, _ = O.similarity_nw(seqs)
    assert rc == 0 and np.array_equal(out.cpu().numpy().view(np.uint64), want.view(np.uint64))
    if world == 1:
        out2 = torch.full((n, n), -1.0, dtype=torch.float64, device="cuda")
        sharding.nw_sharded_step_dedup(uplan, ds.max_len, 0, 1, out2)
        assert torch.equal(out2.view(torch.int64), out.view(torch.int64))
