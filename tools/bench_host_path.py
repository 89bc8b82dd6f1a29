#!/usr/bin/env python3
"""Wall time of the host-pointer entry points (what the R glue calls): H2D + kernels + D2H into a
caller-owned float64 n x n host matrix.  PCIe-inclusive; never the bench `value`."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dynaalign_amd as da
from dynaalign_amd import synth

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
    seqs = synth.to_strings(*synth.h3n2_like(n, 20))
    da.similarityMH(seqs[:2000], 4, 500, seed=12345)   # warm up (context, code objects)
    r = {"n": n, "matrix_GB": n * n * 8 / 1e9}
    t = time.perf_counter(); M = da.similarityMH(seqs, 4, 500, seed=12345); r["mh_s"] = time.perf_counter() - t
    t = time.perf_counter(); W = da.similarityNW(seqs); r["nw_s"] = time.perf_counter() - t
    r["mh_pairs_per_s"] = n * (n - 1) / 2 / r["mh_s"]; r["nw_pairs_per_s"] = n * (n + 1) / 2 / r["nw_s"]
    r["mh_GBs"] = r["matrix_GB"] / r["mh_s"]
    print(json.dumps(r))

if __name__ == "__main__":
    main()
