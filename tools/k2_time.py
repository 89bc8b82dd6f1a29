#!/usr/bin/env python3
"""Times K2 (the library in DYNAALIGN_LIB) on the headline operand and checks the hand-scheduled kernel against the
compiled one (DYNAALIGN_K2_NO_ASM=1) bit for bit.  usage: k2_time.py [n] [workload] [reps] [min_plane_bits]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import dynaalign_amd as da
from dynaalign_amd import device, synth, _capi


def t_ms(f, reps):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    gen = sys.argv[2] if len(sys.argv) > 2 else "h3n2_like"
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    min_bits = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    n_hash = 500
    res, off = getattr(synth, gen)(n, 20)
    ds = device.DeviceSequences(res, off)
    sig, planes = device.minhash_signatures(ds, 4, n_hash, da.hash_family_seeds(12345, n_hash), min_plane_bits=min_bits)
    out = torch.empty((n, n), dtype=torch.float64, device="cuda")
    o16 = torch.empty((n, n), dtype=torch.int16, device="cuda")
    r = {"lib": os.path.basename(os.environ.get("DYNAALIGN_LIB", "default")), "n": n, "workload": gen, "plane_bits": planes.bits}
    os.environ.pop("DYNAALIGN_K2_NO_ASM", None)
    r["f64_ms"] = t_ms(lambda: device.mh_compare(planes, n, n_hash, 0, n, True, _capi.DA_OUT_F64, out=out), reps)
    import ctypes
    lib = _capi.load()
    if hasattr(lib, "da_debug_k2_roles_stats"):               # role-split kernel: tiles taken / stored / stored by compute workgroups / mailboxes
        st = (ctypes.c_uint * 4)()
        lib.da_debug_k2_roles_stats(st)
        r["roles"] = {"taken": st[0], "stored": st[1], "stored_by_compute_wg": st[2], "mailboxes": st[3]}
    r["u16_ms"] = t_ms(lambda: device.mh_compare(planes, n, n_hash, 0, n, True, _capi.DA_OUT_COMPACT, out=o16), reps)
    if "--check" in sys.argv:
        os.environ["DYNAALIGN_K2_NO_ASM"] = "1"
        ref16 = torch.empty_like(o16)
        device.mh_compare(planes, n, n_hash, 0, n, True, _capi.DA_OUT_COMPACT, out=ref16)
        r["u16_equal_compiled"] = bool(torch.equal(ref16, o16))
        del ref16
        ref = torch.empty_like(out)
        device.mh_compare(planes, n, n_hash, 0, n, True, _capi.DA_OUT_F64, out=ref)
        r["f64_equal_compiled"] = bool(torch.equal(ref.view(torch.int64), out.view(torch.int64)))
        r["no_asm_f64_ms"] = t_ms(lambda: device.mh_compare(planes, n, n_hash, 0, n, True, _capi.DA_OUT_F64, out=ref), 3)
        os.environ.pop("DYNAALIGN_K2_NO_ASM", None)
    print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
