#!/usr/bin/env python3
"""Schedule sweep of the pipelined duplicate route inside ONE process (the library re-reads its switches when the environment changes):
da_dev_similarity_mh on 100k h3n2-like for combinations of DYNAALIGN_MH_PIPE_HEAD / _STEP / _WG, 6 calls each after 2 warm-ups.
usage: mh_sweep.py [n]"""
import itertools, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import dynaalign_amd as da
from dynaalign_amd import device, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
res, off = synth.h3n2_like(n, 20)
ds = device.DeviceSequences(res, off)
seeds = da.hash_family_seeds(12345, 500)
out = torch.empty((n, n), dtype=torch.float64, device="cuda")


def run(env):
    keys = ("DYNAALIGN_MH_PIPE_HEAD", "DYNAALIGN_MH_PIPE_STEP", "DYNAALIGN_MH_PIPE_WG")
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    for _ in range(2):
        device.similarity_mh(ds, 4, 500, seeds, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        t = time.perf_counter(); device.similarity_mh(ds, 4, 500, seeds, out=out); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    r = device.mh_last_route()
    return {"ms": round(float(np.mean(ts)), 3), "min": round(min(ts), 3), "k2_span": round(r["k2_ms"], 2), "expand": round(r["expand_ms"], 2), "chunks": r["chunks"]}


print(json.dumps({"default": run({})}))
for head, step, wg in itertools.product((1, 2), (4, 8, 16), (1, 2, 3)):
    print(json.dumps({"head": head, "step": step, "wg": wg, **run({"DYNAALIGN_MH_PIPE_HEAD": head, "DYNAALIGN_MH_PIPE_STEP": step, "DYNAALIGN_MH_PIPE_WG": wg})}), flush=True)
print(json.dumps({"default_again": run({})}))
