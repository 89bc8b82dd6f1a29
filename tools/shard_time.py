#!/usr/bin/env python3
"""Per-rank shard compare (da_dev_mh_compare_shard, one GPU playing rank 0 of `world`) with the hand-scheduled and the compiled kernels.
usage: shard_time.py [n] [gen] [world ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import dynaalign_amd as da
from dynaalign_amd import device, sharding, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
gen = sys.argv[2] if len(sys.argv) > 2 else "uniform_peptides"
worlds = [int(x) for x in sys.argv[3:]] or [2, 4, 8]
res, off = getattr(synth, gen)(n, 20)
ds = device.DeviceSequences(res, off)
sig, planes = device.minhash_signatures(ds, 4, 500, da.hash_family_seeds(12345, 500))
r = {"n": n, "workload": gen, "plane_bits": planes.bits}
for world in worlds:
    plan = sharding.Plan(n, 0, world, sharding.MH_TILE)
    work = sharding.Workspace(plan)
    blocks = {}
    for tag, noasm in (("hand_scheduled", None), ("compiled", "1")):
        if noasm: os.environ["DYNAALIGN_K2_NO_ASM"] = "1"
        else: os.environ.pop("DYNAALIGN_K2_NO_ASM", None)
        sharding.mh_local_block(plan, work, planes, 500); torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            t = time.perf_counter(); sharding.mh_local_block(plan, work, planes, 500); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        r["world%d_%s_ms" % (world, tag)] = round(min(ts), 3)
        blocks[tag] = work.local.clone()
    os.environ.pop("DYNAALIGN_K2_NO_ASM", None)
    r["world%d_equal" % world] = bool(torch.equal(blocks["hand_scheduled"], blocks["compiled"]))
    del work, blocks
print(json.dumps(r))
