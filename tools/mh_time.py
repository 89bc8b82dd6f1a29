#!/usr/bin/env python3
"""Times da_dev_similarity_mh (one C call: plan + K1 + K1b + K2 + expansion -> dense f64 N x N in HBM) on a device-resident set.
usage: mh_time.py [n] [gen] [calls]   (DYNAALIGN_LIB selects the library build; DYNAALIGN_* switches apply)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import dynaalign_amd as da
from dynaalign_amd import device, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
gen = sys.argv[2] if len(sys.argv) > 2 else "h3n2_like"
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 10
res, off = getattr(synth, gen)(n, 20)
ds = device.DeviceSequences(res, off)
seeds = da.hash_family_seeds(12345, 500)
out = torch.empty((n, n), dtype=torch.float64, device="cuda")
for _ in range(3):
    device.similarity_mh(ds, 4, 500, seeds, out=out)
torch.cuda.synchronize()
ts, ph = [], []
for _ in range(calls):
    t = time.perf_counter(); device.similarity_mh(ds, 4, 500, seeds, out=out); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    ph.append(device.mh_last_route())
r = ph[-1]
chk = int(out[:64].view(torch.int64).sum().item())
print(json.dumps({"n": n, "workload": gen, "ms_mean": round(float(np.mean(ts)), 3), "ms_min": round(min(ts), 3), "route": r["expansion"] or ("sparse" if r["sparse"] else "direct"),
                  "k2_ms": round(float(np.mean([p["k2_ms"] for p in ph])), 3), "expand_ms": round(float(np.mean([p["expand_ms"] for p in ph])), 3),
                  "codes_ms": round(r["codes_ms"], 3), "plan_ms": round(r["plan_ms"], 3), "checksum_first_rows": chk}))
