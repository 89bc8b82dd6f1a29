#!/usr/bin/env python3
"""da_dev_similarity_mh on 88 000 h3n2-like + 12 000 uniform random 20-mers (51 000+ unique strings: the duplicate route with a table whose
dictionaries need more than 12 code planes), with and without the heavy / rare split.  usage: mh_mixed_time.py [calls]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import dynaalign_amd as da
from dynaalign_amd import device, synth

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = 100000
res, off = synth.h3n2_like(n, 20)
seqs = synth.to_strings(res, off)[:88000]
rng = np.random.RandomState(99)
alpha = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
seqs += ["".join(map(chr, alpha[rng.randint(0, 20, 20)])) for _ in range(12000)]
seqs = [seqs[i] for i in rng.permutation(len(seqs))]
b = "".join(seqs).encode("latin-1")
ds = device.DeviceSequences(np.frombuffer(b, np.uint8).copy(), np.arange(0, 20 * n + 1, 20, dtype=np.int64))
seeds = da.hash_family_seeds(12345, 500)
bufs = [torch.empty((n, n), dtype=torch.float64, device="cuda") for _ in range(2)]
for i, sw in enumerate(({"DYNAALIGN_MH_NO_HYBRID": "1"}, {})):
    os.environ.update(sw)
    try:
        for _ in range(2):
            device.similarity_mh(ds, 4, 500, seeds, out=bufs[i])
        ts = []
        for _ in range(calls):
            torch.cuda.synchronize(); t = time.perf_counter(); device.similarity_mh(ds, 4, 500, seeds, out=bufs[i]); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        r = device.mh_last_route()
    finally:
        for k in sw:
            del os.environ[k]
    print(json.dumps({"switches": sw, "ms_mean": round(float(np.mean(ts)), 3), "ms_min": round(min(ts), 3), "unique": r["unique"], "route": r["expansion"], "plane_bits": r["plane_bits"],
                      "split": r["split"], "rare_pairs": r["rare_pairs"], "plane_bits_without": r["plane_bits_without"], "k2_ms": round(r["k2_ms"], 3), "expand_ms": round(r["expand_ms"], 3),
                      "codes_ms": round(r["codes_ms"], 3)}), flush=True)
print(json.dumps({"bit_identical": bool(torch.equal(bufs[0].view(torch.int64), bufs[1].view(torch.int64)))}))
