#!/usr/bin/env python3
"""Where does k_mh_compare's time go?  Times the kernel at N=100k for several n_hash
(fixed per-tile cost vs per-group cost) and for both output kinds (store cost)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import dynaalign_amd as da
from dynaalign_amd import device, synth, _capi

def t_ms(f, reps=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    res, off = synth.h3n2_like(n, 20)
    ds = device.DeviceSequences(res, off)
    out = torch.empty((n, n), dtype=torch.float64, device="cuda")
    out16 = out.view(torch.int16)[:, :n] if False else torch.empty((n, n), dtype=torch.int16, device="cuda")
    r = {}
    for n_hash in (32, 64, 256, 480, 500, 512):
        sig, planes = device.minhash_signatures(ds, 4, n_hash, da.hash_family_seeds(12345, n_hash))
        r["f64_h%d" % n_hash] = t_ms(lambda: device.mh_compare(planes, n, n_hash, 0, n, True, _capi.DA_OUT_F64, out=out))
        r["u16_h%d" % n_hash] = t_ms(lambda: device.mh_compare(planes, n, n_hash, 0, n, True, _capi.DA_OUT_COMPACT, out=out16))
    print(json.dumps(r))

if __name__ == "__main__":
    main()
