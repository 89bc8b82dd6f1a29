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
