#!/usr/bin/env python3
"""Per-workgroup phase stamps of k_mh_compare (needs the -DDA_K2_TIMING build of the library:
tools/build_timing_lib.sh, then DYNAALIGN_LIB=dynaalign_amd/lib/libdynaalign_hip_timing.so).
Prints how long a workgroup spends before / inside / after its plane loop and, per CU, how much
of the kernel's time 0/1/2/3 resident workgroups were inside the loop."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import dynaalign_amd as da
from dynaalign_amd import device, synth, _capi


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    n_hash = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    kind = _capi.DA_OUT_F64 if (len(sys.argv) <= 3 or sys.argv[3] == "f64") else _capi.DA_OUT_COMPACT
    raw = len(sys.argv) > 4 and sys.argv[4] == "32"
    lib = _capi.load()
    res, off = synth.h3n2_like(n, 20)
    ds = device.DeviceSequences(res, off)
    sig, planes = device.minhash_signatures(ds, 4, n_hash, da.hash_family_seeds(12345, n_hash), raw_planes=raw)
    out = torch.empty((n, n), dtype=torch.float64 if kind == _capi.DA_OUT_F64 else torch.int16, device="cuda")
    T = (n + 127) // 128
    nblocks = ((T * (T + 1) // 2 + 7) // 8) * 8
    device.mh_compare(planes, n, n_hash, 0, n, True, kind, out=out)       # warm
    buf = torch.zeros((nblocks, 8), dtype=torch.int64, device="cuda")
    fn = lib.da_debug_set_k2_timing
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
    assert fn(buf.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    device.mh_compare(planes, n, n_hash, 0, n, True, kind, out=out)
    e1.record()
    torch.cuda.synchronize()
    assert fn(None) == 0
    t = buf.cpu().numpy()
    t = t[t[:, 3] > 0]
    tick = 10e-9 * 1e6                      # wall_clock64: 100 MHz -> us
    pro = (t[:, 1] - t[:, 0]) * tick
    loop = (t[:, 2] - t[:, 1]) * tick
    epi = (t[:, 3] - t[:, 2]) * tick
    span = (t[:, 3].max() - t[:, 0].min()) * tick
    r = {"n": n, "n_hash": n_hash, "plane_bits": planes.bits, "kind": "f64" if kind == _capi.DA_OUT_F64 else "u16",
         "kernel_ms_events": e0.elapsed_time(e1), "span_ms_stamps": span / 1e3, "workgroups": int(len(t)),
         "prologue_us": [float(np.mean(pro)), float(np.percentile(pro, 50)), float(np.percentile(pro, 95))],
         "loop_us": [float(np.mean(loop)), float(np.percentile(loop, 50)), float(np.percentile(loop, 95))],
         "epilogue_issue_us": [float(np.mean(epi)), float(np.percentile(epi, 50)), float(np.percentile(epi, 95))]}
    if kind != _capi.DA_OUT_F64 and (t[:, 6] > 0).all() and (t[:, 7] > 0).all():   # uint16 kind, a12 kernel: prologue split
        r["prologue_parts_us"] = {"entry_to_tile_decoded": float(np.mean((t[:, 6] - t[:, 0]) * tick)),
                                  "tile_decoded_to_dma_source": float(np.mean((t[:, 7] - t[:, 6]) * tick)),
                                  "dma_source_to_loop": float(np.mean((t[:, 1] - t[:, 7]) * tick))}
    elif (t[:, 6] > 0).all() and (t[:, 7] > 0).all():      # f64 kind: table built / direct stores issued
        r["epilogue_parts_us"] = {"ratio_table": float(np.mean((t[:, 6] - t[:, 2]) * tick)),
                                  "direct_stores": float(np.mean((t[:, 7] - t[:, 6]) * tick)),
                                  "mirrored_stores": float(np.mean((t[:, 3] - t[:, 7]) * tick))}
    # residency: per CU (xcc, se, sh, cu) the workgroups in start order; slot reuse gap = next start - previous exit
    hw, xcc = t[:, 4], t[:, 5] & 0xF
    cu = (xcc << 16) | (hw & 0xFF00) | ((hw >> 13) & 0x7) << 4 | ((hw >> 12) & 1)   # cu_id bits 11:8, sh 12, se 15:13
    order = np.lexsort((t[:, 0], cu))
    ts, cus = t[order], cu[order]
    ncu = len(np.unique(cus))
    r["cus_seen"] = int(ncu)
    # time-weighted number of workgroups inside the loop, sampled over the kernel
    t0 = t[:, 0].min()
    grid = np.linspace(t0, t[:, 3].max(), 4000)
    frac = np.zeros(6)
    sample_cus = np.unique(cus)[:: max(1, ncu // 32)]
    for c in sample_cus:
        m = ts[cus == c]
        inloop = ((m[:, 1][None, :] <= grid[:, None]) & (grid[:, None] < m[:, 2][None, :])).sum(1)
        resident = ((m[:, 0][None, :] <= grid[:, None]) & (grid[:, None] < m[:, 3][None, :])).sum(1)
        for k in range(5):
            frac[k] += np.mean(inloop == k)
        frac[5] += np.mean(resident)
    r["time_fraction_with_k_workgroups_in_loop"] = [float(x / len(sample_cus)) for x in frac[:5]]
    r["mean_resident_workgroups_by_stamps"] = float(frac[5] / len(sample_cus))
    r["tiles_per_cu"] = float(len(t) / ncu)
    # relaunch gap: on a CU with S resident workgroups, the (k+S)-th start follows the k-th exit (both sorted)
    gaps = []
    for c in sample_cus:
        m = ts[cus == c]
        S = int(round(r["mean_resident_workgroups_by_stamps"] + 0.49))
        st, ex = np.sort(m[:, 0]), np.sort(m[:, 3])
        if len(st) > 2 * S:
            gaps.append(np.median((st[S:] - ex[:len(st) - S]) * tick))
    r["relaunch_gap_us_median"] = float(np.median(gaps)) if gaps else None
    print(json.dumps(r))


if __name__ == "__main__":
    main()
