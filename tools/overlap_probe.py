#!/usr/bin/env python3
"""Do a VALU-bound compare launch and an HBM-bound widen launch overlap when issued on two streams?  (feasibility probe for
splitting K2's float64 stores into a trailing kernel)"""
import json, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import dynaalign_amd as da
from dynaalign_amd import device, synth, _capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n_hash = 500
res, off = synth.h3n2_like(n, 20)
ds = device.DeviceSequences(res, off)
sig, planes = device.minhash_signatures(ds, 4, n_hash, da.hash_family_seeds(12345, n_hash))
cnt_a = torch.empty((n, n), dtype=torch.int16, device="cuda")
cnt_b = torch.empty((n, n), dtype=torch.int16, device="cuda")
out = torch.empty((n, n), dtype=torch.float64, device="cuda")
device.mh_compare(planes, n, n_hash, 0, n, True, _capi.DA_OUT_COMPACT, out=cnt_b)
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def timed(f, reps=3):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    return min(ts)

def k2_u16():
    with torch.cuda.stream(sa):
        device.mh_compare(planes, n, n_hash, 0, n, True, _capi.DA_OUT_COMPACT, out=cnt_a)
def k2_f64():
    with torch.cuda.stream(sa):
        device.mh_compare(planes, n, n_hash, 0, n, True, _capi.DA_OUT_F64, out=out)
def widen():
    with torch.cuda.stream(sb):
        device.widen(cnt_b, False, n_hash, out=out)
def both():
    k2_u16(); widen()
r = {"n": n, "k2_f64_ms": timed(k2_f64), "k2_u16_ms": timed(k2_u16), "widen_ms": timed(widen), "both_streams_ms": timed(both)}
print(json.dumps(r))
