#!/usr/bin/env python3
"""Per-tile stamps of the PERSISTENT compare kernel (timing build, tools/build_timing_lib.sh): for the first 64 tiles of
every workgroup: decode of the next tile | the asm block | epilogue stores.  usage: k2p_timeline.py [f64|u16]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import dynaalign_amd as da
from dynaalign_amd import device, synth, _capi

n, n_hash = 100000, 500
kind = _capi.DA_OUT_COMPACT if (len(sys.argv) > 1 and sys.argv[1] == "u16") else _capi.DA_OUT_F64
lib = _capi.load()
res, off = synth.h3n2_like(n, 20)
ds = device.DeviceSequences(res, off)
sig, planes = device.minhash_signatures(ds, 4, n_hash, da.hash_family_seeds(12345, n_hash))
out = torch.empty((n, n), dtype=torch.float64 if kind == _capi.DA_OUT_F64 else torch.int16, device="cuda")
device.mh_compare(planes, n, n_hash, 0, n, True, kind, out=out)
buf = torch.zeros((1024 * 64 + 4096, 4), dtype=torch.int64, device="cuda")
fn = lib.da_debug_set_k2_timing
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
assert fn(buf.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
device.mh_compare(planes, n, n_hash, 0, n, True, kind, out=out)
e1.record()
torch.cuda.synchronize()
assert fn(None) == 0
t = buf.cpu().numpy()[:1024 * 64].reshape(1024, 64, 4)
t = t[:, 2:60]                       # skip the first tiles of a workgroup (cold start) -- and the last stamped ones
ok = (t > 0).all(axis=2)
tick = 0.01                          # 100 MHz -> us
dec = (t[..., 1] - t[..., 0])[ok] * tick
blk = (t[..., 2] - t[..., 1])[ok] * tick
epi = (t[..., 3] - t[..., 2])[ok] * tick
per = (t[:, 1:, 0] - t[:, :-1, 0])[ok[:, 1:] & ok[:, :-1]] * tick
print(json.dumps({"kind": "f64" if kind == _capi.DA_OUT_F64 else "u16", "kernel_ms": e0.elapsed_time(e1),
                  "decode_next_us": [float(dec.mean()), float(np.percentile(dec, 50)), float(np.percentile(dec, 95))],
                  "block_us": [float(blk.mean()), float(np.percentile(blk, 50)), float(np.percentile(blk, 95))],
                  "epilogue_us": [float(epi.mean()), float(np.percentile(epi, 50)), float(np.percentile(epi, 95))],
                  "tile_period_us": [float(per.mean()), float(np.percentile(per, 50)), float(np.percentile(per, 95))]}))
