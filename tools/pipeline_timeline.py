#!/usr/bin/env python3
"""Kernel timeline of ONE call of da_dev_similarity_mh (100k h3n2-like) from a rocprofv3 --kernel-trace run: which kernels overlap.
   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/pipeline_timeline.py run
   python3 tools/pipeline_timeline.py report <dir>   ->  a table of the last call's kernels (start / end in ms after the call's first kernel)"""
import csv, glob, os, re, sys


def run():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import torch
    import dynaalign_amd as da
    from dynaalign_amd import device, synth
    n = 100000
    res, off = synth.h3n2_like(n, 20)
    ds = device.DeviceSequences(res, off)
    seeds = da.hash_family_seeds(7, 500)
    out = torch.empty((n, n), dtype=torch.float64, device="cuda")
    for _ in range(4):
        device.similarity_mh(ds, 4, 500, seeds, out=out)
        torch.cuda.synchronize()
    print(device.mh_last_route())


def report(d):
    rows = []
    for f in glob.glob(os.path.join(d, "*", "*kernel_trace.csv")) + glob.glob(os.path.join(d, "*kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", r["Kernel_Name"])
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:40], r.get("Queue_Id", "?")))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith("k_dd_insert")]      # the plan's first kernel opens a call
    last = rows[starts[-1]:]
    t0 = last[0][0]
    print("%-34s %6s %9s %9s %8s" % ("kernel", "queue", "start ms", "end ms", "ms"))
    for s, e, k, q in last:
        print("%-34s %6s %9.3f %9.3f %8.3f" % (k, q, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
    k2 = [(s, e) for s, e, k, q in last if k.startswith("k_mh_compare_p12")]
    ex = [(s, e) for s, e, k, q in last if k.startswith("k_expand_stream")]
    if k2 and ex:
        both = sum(max(0, min(e1, e2) - max(s1, s2)) for s1, e1 in k2 for s2, e2 in ex)
        print("\ncall: %.3f ms; K2 band launches busy %.3f ms, k_expand_stream launches busy %.3f ms (sum), both at once %.3f ms"
              % ((max(e for s, e, k, q in last) - t0) / 1e6, sum(e - s for s, e in k2) / 1e6, sum(e - s for s, e in ex) / 1e6, both / 1e6))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
