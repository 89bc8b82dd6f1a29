#!/usr/bin/env python3
"""Throughput of the long-sequence NW kernel (k_nw_long) on HA-like lengths:
N random sequences of length L (default 2000 x 566), dense f64 N x N in HBM."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import dynaalign_amd as da
from dynaalign_amd import device, synth, _capi

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 566
    res, off = synth.uniform_peptides(n, L, seed=5)
    ds = device.DeviceSequences(res, off)
    assert int(device.nw_encode(ds).item()) == 0
    out = torch.empty((n, n), dtype=torch.float64, device="cuda")
    device.nw(ds, out=out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); device.nw(ds, out=out); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3
    pairs = n * (n + 1) // 2
    print(json.dumps({"kernel": "k_nw_long", "n": n, "L": L, "ms": t * 1e3, "pairs_per_s": pairs / t,
                      "gcups": pairs * L * L / t / 1e9}))

if __name__ == "__main__":
    main()
