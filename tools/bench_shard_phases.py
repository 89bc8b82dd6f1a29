#!/usr/bin/env python3
"""Per-rank phases of the row-sharded path, measured on ONE GPU (the gather itself needs the
8-GPU node): shard compute for a chosen rank and the finalize kernel at N = 100k."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import dynaalign_amd as da
from dynaalign_amd import device, sharding, synth

def t_ms(f, reps=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    n_hash = 500
    res, off = synth.h3n2_like(n, 20)
    ds = device.DeviceSequences(res, off)
    sig, planes = device.minhash_signatures(ds, 4, n_hash, da.hash_family_seeds(12345, n_hash))
    assert int(device.nw_encode(ds).item()) == 0
    out = torch.empty((n, n), dtype=torch.float64, device="cuda")
    r = {"n": n}
    for world in (2, 4, 8):
        plan = sharding.Plan(n, 0, world, sharding.MH_TILE)
        work = sharding.Workspace(plan)
        r["mh_w%d" % world] = {"shard_ms_rank0": t_ms(lambda: sharding.mh_local_block(plan, work, planes, n_hash)),
                               "finalize_ms": t_ms(lambda: sharding.finalize_shards(plan, work.gathered, False, n_hash, out)),
                               "block_GB": work.local.numel() * 2 / 1e9, "gathered_GB": work.gathered.numel() * 2 / 1e9}
        del work
        pwork = sharding.PackedWorkspace(plan, n_hash)
        sharding.mh_local_block(plan, pwork, planes, n_hash)
        pwork.gathered.zero_()
        r["mh_w%d" % world].update({"pack_ms": t_ms(lambda: sharding.pack_local_block(plan, pwork)),
                                    "finalize_packed_ms": t_ms(lambda: sharding.finalize_shards_packed(plan, pwork, pwork.gathered, n_hash, out)),
                                    "packed_block_GB": pwork.block_bytes / 1e9, "value_bits": pwork.bits})
        del pwork
        nplan = sharding.Plan(n, 0, world, sharding.NW_TILE)
        nwork = sharding.Workspace(nplan)
        r["nw_w%d" % world] = {"shard_ms_rank0": t_ms(lambda: sharding.nw_local_block(nplan, nwork, ds), 1),
                               "finalize_ms": t_ms(lambda: sharding.finalize_shards(nplan, nwork.gathered, True, 0, out))}
        del nwork
    # the duplicate route: per-rank pieces with this GPU playing rank 0 (the all-gather itself needs the 8-GPU node)
    seeds = da.hash_family_seeds(12345, n_hash)
    d_seeds = torch.from_numpy(seeds.view(np.int32).copy()).cuda()
    up = device.UniquePlan(ds.residues, ds.offsets, n, ds.total)
    nup = device.UniquePlan(ds.codes, ds.offsets, n, ds.total)
    r["unique"] = up.unique
    r["plan_ms"] = t_ms(lambda: device.UniquePlan(ds.residues, ds.offsets, n, ds.total))
    for world in (1, 2, 4, 8):
        state = {}
        def local():
            state["pw"] = sharding.mh_unique_local(up, ds, 4, n_hash, d_seeds, 0, world)
        d = {"codes_and_shard_ms_rank0": t_ms(local)}
        plan, work = state["pw"]
        work.gathered.zero_()
        d["packed_block_GB"] = work.block_bytes / 1e9
        d["shards_to_table_ms"] = t_ms(lambda: device.shards_to_table(work.gathered, 0, up.unique, world, work.bits))
        table = device.shards_to_table(work.gathered, 0, up.unique, world, work.bits)
        ework = torch.empty(device.expand_workspace_bytes(n, up.unique, False, n_hash, 0), dtype=torch.uint8, device="cuda")
        d["expand_ms"] = t_ms(lambda: device.expand_unique(table, up, False, n_hash, 0, out, work=ework))
        r["mh_dedup_w%d" % world] = d
        del work, table, ework, state
        nd = {"rows_ms_rank0": t_ms(lambda: sharding.nw_unique_rows_local(nup, ds.max_len, 0, world), 1)}
        blk = sharding.nw_unique_rows_local(nup, ds.max_len, 0, world)
        nd["block_GB"] = blk.numel() * 2 / 1e9
        gathered = torch.zeros((world * blk.shape[0], blk.shape[1]), dtype=torch.int16, device="cuda")
        ework = torch.empty(device.expand_workspace_bytes(n, nup.unique, True, 0, ds.max_len), dtype=torch.uint8, device="cuda")
        nd["expand_ms"] = t_ms(lambda: device.expand_unique(gathered, nup, True, 0, ds.max_len, out, table_world=world, work=ework))
        r["nw_dedup_w%d" % world] = nd
        del blk, gathered, ework
    print(json.dumps(r))

if __name__ == "__main__":
    main()
