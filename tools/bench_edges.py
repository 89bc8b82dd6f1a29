#!/usr/bin/env python3
"""similarityMH + clusterbreak's quantile threshold as an edge list, all on the device:
K1 + K2 (uint16 counts) + histogram + exact type-7 quantile + edge extraction, N = 100k."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import dynaalign_amd as da
from dynaalign_amd import device, synth, _capi

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
    n_hash = 500
    res, off = synth.h3n2_like(n, 20)
    ds = device.DeviceSequences(res, off)
    seeds = da.hash_family_seeds(12345, n_hash)
    cnt = torch.empty((n, n), dtype=torch.int16, device="cuda")
    ev = lambda: torch.cuda.Event(enable_timing=True)
    values = np.arange(n_hash + 1) / n_hash
    def run():
        e = [ev() for _ in range(5)]
        e[0].record()
        sig, planes = device.minhash_signatures(ds, 4, n_hash, seeds)
        e[1].record()
        device.mh_compare(planes, n, n_hash, 0, n, True, _capi.DA_OUT_COMPACT, out=cnt)
        e[2].record()
        hist = device.upper_histogram(cnt, n, n_hash + 1)
        e[3].record()
        h = hist.cpu().numpy().astype(np.uint64)
        thr = da.quantile_type7(h, values, p)
        keep = (~(values < thr)) & (np.arange(n_hash + 1) != 0)
        m = int(h[keep].sum()) + n
        ei, ej, evv, c = device.extract_edges(cnt, n, keep, m)
        e[4].record()
        torch.cuda.synchronize()
        assert int(c.item()) == m
        return thr, m, [e[i].elapsed_time(e[i + 1]) for i in range(4)]
    run()
    thr, m, ms = run()
    print(json.dumps({"n": n, "thresh_p": p, "threshold": thr, "edges": m, "edge_fraction": m / (n * (n + 1) / 2),
                      "ms": {"k1": ms[0], "k2_u16": ms[1], "histogram": ms[2], "quantile+extract": ms[3]},
                      "total_ms": sum(ms), "dense_f64_bytes": n * n * 8, "edge_list_bytes": m * 10}))

if __name__ == "__main__":
    main()
