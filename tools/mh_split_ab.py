#!/usr/bin/env python3
"""A/B of the heavy / rare split of da_dev_similarity_mh (8 dense planes + incidence lists) against the full-width compare:
whole-matrix equality and call times, on the direct route (DYNAALIGN_MH_NO_DEDUP=1) and on the duplicate route.
usage: mh_split_ab.py [n] [gen] [calls]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import dynaalign_amd as da
from dynaalign_amd import device, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
gen = sys.argv[2] if len(sys.argv) > 2 else "h3n2_like"
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 6
res, off = getattr(synth, gen)(n, 20)
ds = device.DeviceSequences(res, off)
seeds = da.hash_family_seeds(12345, 500)
ref = torch.empty((n, n), dtype=torch.float64, device="cuda")
out = torch.empty((n, n), dtype=torch.float64, device="cuda")


def run(buf, env):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update({k: v for k, v in env.items() if v is not None})
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
    try:
        for _ in range(2):
            device.similarity_mh(ds, 4, 500, seeds, out=buf)
        torch.cuda.synchronize()
        ts, ph = [], []
        for _ in range(calls):
            t = time.perf_counter(); device.similarity_mh(ds, 4, 500, seeds, out=buf); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
            ph.append(device.mh_last_route())
        r = ph[-1]
        return {"ms_mean": round(float(np.mean(ts)), 3), "ms_min": round(min(ts), 3), "route": r["expansion"] or ("sparse" if r["sparse"] else "direct"),
                "plane_bits": r["plane_bits"], "split": r["split"], "rare_pairs": r["rare_pairs"], "plane_bits_without": r["plane_bits_without"],
                "k2_ms": round(float(np.mean([p["k2_ms"] for p in ph])), 3), "expand_ms": round(float(np.mean([p["expand_ms"] for p in ph])), 3),
                "codes_ms": round(float(np.mean([p["codes_ms"] for p in ph])), 3), "plan_ms": round(r["plan_ms"], 3)}
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


for label, base in (("direct", {"DYNAALIGN_MH_NO_DEDUP": "1"}), ("duplicate_route", {})):
    a = run(ref, dict(base, DYNAALIGN_MH_NO_HYBRID="1"))
    b = run(out, dict(base, DYNAALIGN_MH_HYBRID_DEDUP="1", DYNAALIGN_MH_HYBRID_MIN_N="2048"))
    same = bool(torch.equal(ref.view(torch.int64), out.view(torch.int64)))
    print(json.dumps({"n": n, "workload": gen, "leg": label, "full_width": a, "split": b, "bit_identical": same}), flush=True)
    if not same:
        d = (ref[:4096].view(torch.int64) != out[:4096].view(torch.int64))
        idx = d.nonzero()[:8].tolist()
        print("first differences (rows < 4096)", [(i, j, ref[i, j].item() * 500, out[i, j].item() * 500) for i, j in idx], "count", int(d.sum().item()), flush=True)
