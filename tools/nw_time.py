#!/usr/bin/env python3
"""Times similarityNW on a device-resident set with and without the duplicate-collapsing route.  usage: nw_time.py [n] [gen]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from dynaalign_amd import device, synth, _capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
gen = sys.argv[2] if len(sys.argv) > 2 else "h3n2_like"
res, off = getattr(synth, gen)(n, 20)
ds = device.DeviceSequences(res, off)
assert int(device.nw_encode(ds).item()) == 0
out = torch.empty((n, n), dtype=torch.float64, device="cuda")
seqs = synth.to_strings(res, off)
r = {"n": n, "workload": gen, "unique": len(set(seqs))}
for tag, env in (("dedup", None), ("direct", "1")):
    if env: os.environ["DYNAALIGN_NW_NO_DEDUP"] = env
    else: os.environ.pop("DYNAALIGN_NW_NO_DEDUP", None)
    device.nw(ds, out=out); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t = time.perf_counter(); device.nw(ds, out=out); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    r[tag + "_ms"] = [x * 1e3 for x in ts]
    if tag == "dedup": ref = out.clone() if n <= 60000 else out[:2000].clone()
    else: r["equal"] = bool(torch.equal((out if n <= 60000 else out[:2000]).view(torch.int64), ref.view(torch.int64)))
print(json.dumps(r))
