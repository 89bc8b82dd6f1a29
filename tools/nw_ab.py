#!/usr/bin/env python3
"""A / B of the NW duplicate route's ordered DP on a device-resident set: with prefix sharing (default) and without
(DYNAALIGN_NW_NO_PREFIX_SHARE=1); the direct sweep beside it; results compared bit for bit.  usage: nw_ab.py [n] [gen] [len]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from dynaalign_amd import device, synth, _capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
gen = sys.argv[2] if len(sys.argv) > 2 else "h3n2_like"
L = int(sys.argv[3]) if len(sys.argv) > 3 else 20
res, off = getattr(synth, gen)(n, L)
ds = device.DeviceSequences(res, off)
assert int(device.nw_encode(ds).item()) == 0
out = torch.empty((n, n), dtype=torch.float64, device="cuda")
r = {"n": n, "workload": gen, "len": L}
ref = None
for form, noasm in (("prefix_share", None), ("no_prefix_share", "1"), ("prefix_share_again", None)):
    for tag, nodedup in (("dedup", None), ("direct", "1")):
        for k, v in (("DYNAALIGN_NW_NO_PREFIX_SHARE", noasm), ("DYNAALIGN_NW_NO_DEDUP", nodedup)):
            if v: os.environ[k] = v
            else: os.environ.pop(k, None)
        device.nw(ds, out=out); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t = time.perf_counter(); device.nw(ds, out=out); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        rt = device.nw_last_route()
        r["%s_%s_ms" % (form, tag)] = [round(x * 1e3, 2) for x in ts]
        r["%s_%s_dp_ms" % (form, tag)] = round(rt["dp_ms"], 2)
        chk = out[: min(n, 3000)].view(torch.int64).clone() if n > 40000 else out.view(torch.int64).clone()
        if ref is None: ref = chk
        else: r["%s_%s_equal" % (form, tag)] = bool(torch.equal(chk, ref))
print(json.dumps(r))
