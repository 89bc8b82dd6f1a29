#!/usr/bin/env python3
"""bench.py -- sequence-pairs/s of the all-pairs similarity hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full pass of similarityMH(k=4, n_hash=500) over the workload with the
packed residues already resident in HBM: signature build (K1) + all-pairs compare (K2)
producing the dense float64 N x N matrix in HBM (what the reference returns to R).  With
N > 1 ranks the pair space is row-sharded, each rank's compact block is exchanged with ONE
RCCL all-gather and widened/mirrored to the full float64 matrix on every rank.

Headline workload (BASELINE.json configs[3], the one the metric is quoted on):
100 000 h3n2-like 20-mers, MinHash k=4 n_hash=500, hash seed 12345.  The same JSON line also
carries the NW BLOSUM62 figure (`nw`), the dominant kernel's roofline position (`roofline`)
and the CPU oracle timed on this box's host cores (`cpu_baseline`, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_LANEOPS = 256 * 4 * 32 * 2.4e9   # CUs x SIMDs x lanes/clk x clock = 7.86e13 lane-ops/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=100000, help="number of peptides (default: the 100k headline set)")
    ap.add_argument("--workload", default="h3n2like", choices=["h3n2like", "uniform"])
    ap.add_argument("--no-nw", action="store_true", help="skip the similarityNW measurement")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU oracle baseline")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time per baseline leg")
    return ap.parse_args()


def cpu_baseline(res, off, seeds, target_s):
    """Time the CPU oracle (kind 'port': the repo's C restatement with the reference's loop
    structure and its two OpenMP sites) on a bounded sample of the same workload."""
    import numpy as np
    import oracle_lib as O
    O.build()
    L = O.lib()
    threads = O.num_threads()

    def run_mh(ns):
        r, o = res[:off[ns]], off[:ns + 1]
        out = np.empty((ns, ns), np.float64)
        t = time.perf_counter()
        rc = L.orc_similarity_mh(np.ascontiguousarray(r), np.ascontiguousarray(o), ns, 4, 500, seeds, out)
        dt = time.perf_counter() - t
        assert rc == 0
        return ns * (ns - 1) / 2 / dt, dt

    rate, _ = run_mh(1500)
    ns = int(min(len(off) - 1, 24000, max(2000, (2 * rate * target_s) ** 0.5)))
    mh_rate, mh_dt = run_mh(ns)

    def run_nw(rows, ns):
        import ctypes as C
        r, o = np.ascontiguousarray(res[:off[ns]]), np.ascontiguousarray(off[:ns + 1])
        mt = np.empty((rows, ns), np.int32)
        ln = np.empty((rows, ns), np.int32)
        buf = C.create_string_buffer(256)
        os.environ["OMP_NUM_THREADS"] = "1"
        t = time.perf_counter()
        rc = L.orc_nw_rows(r, o, ns, 0, rows, b"BLOSUM62", 10, 4, mt.ctypes.data, ln.ctypes.data, None, buf, 256)
        dt = time.perf_counter() - t
        assert rc == 0
        return rows * ns / dt, dt

    # NW is single-threaded in the reference (src/pairwiseSeqAlign.cpp:340-352): time it on 1 core
    import subprocess
    code = ("import sys,os,time,json;os.environ['OMP_NUM_THREADS']='1';sys.path.insert(0,%r);sys.path.insert(0,%r);"
            "import numpy as np,oracle_lib as O;from dynaalign_amd import synth;"
            "res,off=synth.%s(4000,20);seqs=synth.to_strings(res,off);"
            "t=time.perf_counter();rc,mt,ln,sc,_=O.nw_rows(seqs,0,%d);dt=time.perf_counter()-t;"
            "print(json.dumps({'pairs':%d*4000,'dt':dt}))")
    return {"mh_rate": mh_rate, "mh_n": ns, "mh_dt": mh_dt, "threads": threads, "nw_code": code}


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if world > 1:
        dist.barrier()
    import dynaalign_amd as da
    from dynaalign_amd import device, synth, _capi
    if world > 1:
        from dynaalign_amd import sharding

    n, k, n_hash, L = a.n, 4, 500, 20
    gen = synth.h3n2_like if a.workload == "h3n2like" else synth.uniform_peptides
    res, off = gen(n, L)
    seeds = da.hash_family_seeds(12345, n_hash)
    ds = device.DeviceSequences(res, off, "cuda")
    d_seeds = torch.from_numpy(seeds.view(np.int32).copy()).cuda()
    sig = torch.empty((n, device.sig_ld(n_hash)), dtype=torch.int32, device="cuda")
    planes = torch.empty((n, device.planes_ld(n_hash)), dtype=torch.int32, device="cuda")

    pairs_mh = n * (n - 1) // 2            # unordered pairs, diagonal excluded (src/minHash.cpp:164)
    pairs_nw = n * (n + 1) // 2            # the reference computes the NW diagonal (src/pairwiseSeqAlign.cpp:342)

    ev = lambda: torch.cuda.Event(enable_timing=True)
    k1_ms, k2_ms = [], []

    if world == 1:
        out = torch.empty((n, n), dtype=torch.float64, device="cuda")

        def step(record=False):
            e0, e1, e2 = ev(), ev(), ev()
            e0.record()
            device.minhash_signatures(ds, k, n_hash, d_seeds, out=sig, planes=planes)
            e1.record()
            device.mh_compare(planes, n, n_hash, 0, n, True, _capi.DA_OUT_F64, out=out)
            e2.record()
            if record:
                return e0, e1, e2
    else:
        plan = sharding.Plan(n, rank, world)
        out = torch.empty((n, n), dtype=torch.float64, device="cuda")
        work = sharding.MHWorkspace(plan, "cuda")

        def step(record=False):
            e0, e1, e2 = ev(), ev(), ev()
            e0.record()
            device.minhash_signatures(ds, k, n_hash, d_seeds, out=sig, planes=planes)   # every rank rebuilds all signatures (2 MB in)
            e1.record()
            sharding.mh_sharded_step(plan, work, planes, n_hash, out)
            e2.record()
            if record:
                return e0, e1, e2

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    evs = [step(True) for _ in range(a.steps)]
    sync()
    dt = time.perf_counter() - t0
    for e0, e1, e2 in evs:
        k1_ms.append(e0.elapsed_time(e1))
        k2_ms.append(e1.elapsed_time(e2))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / a.steps * 1e3
    value = pairs_mh / (dt / a.steps)

    # ---- roofline of the dominant kernel (K2 compare), HIP events on the launch stream
    k2 = float(np.mean(k2_ms)) * 1e-3
    bytes_k2 = n * n_hash * 4 + n * n * 8          # read the signatures once + write the f64 N x N
    roof = {"kernel": "k_mh_compare", "bound": "hbm", "achieved": bytes_k2 / k2 / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": bytes_k2 / k2 / 1e9 / HBM_PEAK_GBS, "traffic": None,
            "avg_launch_ms": k2 * 1e3, "algorithmic_bytes_per_launch": bytes_k2,
            "valu": {"note": "the binding unit: 2 integer lane-ops per u32 compare; upper-triangle tiles only",
                     "lane_ops_per_launch": 2 * 500 * 128 * 128 * ((n + 127) // 128) * ((n + 127) // 128 + 1) // 2,
                     "peak_lane_ops_per_s": VALU_PEAK_LANEOPS}}
    if world > 1:
        roof["note"] = "k2 interval holds compare + all-gather + widen on this rank"
    roof["valu"]["achieved_lane_ops_per_s"] = roof["valu"]["lane_ops_per_launch"] / k2
    roof["valu"]["frac"] = roof["valu"]["achieved_lane_ops_per_s"] / VALU_PEAK_LANEOPS

    line = {
        "metric": "sequence-pairs/sec (MinHash k=4 n_hash=500; NW BLOSUM62) at 1/2/4/8 MI355X",
        "value": value, "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": "similarityMH k=4 n_hash=500 on %d %s 20-mers (seed 12345), dense f64 NxN in HBM"
                               % (n, "h3n2-like" if a.workload == "h3n2like" else "uniform"),
                   "n": n, "k": k, "n_hash": n_hash, "pairs": pairs_mh,
                   "sharding": "1 GPU, upper-triangle tiles + mirrored store" if world == 1
                   else "row-tiles cyclic over %d ranks, one RCCL all-gather of uint16 counts" % world},
        "roofline": roof,
        "phases_ms": {"k1_signatures": float(np.mean(k1_ms)), "k2_compare": float(np.mean(k2_ms))},
    }

    # ---- similarityNW on the same set (second half of the metric)
    if not a.no_nw and world == 1:
        bad = device.nw_encode(ds)
        assert int(bad.item()) == 0
        device.nw(ds, "BLOSUM62", 10, 4, 0, n, True, _capi.DA_OUT_F64, out=out)
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        e0.record()
        device.nw(ds, "BLOSUM62", 10, 4, 0, n, True, _capi.DA_OUT_F64, out=out)
        e1.record()
        torch.cuda.synchronize()
        t_nw = e0.elapsed_time(e1) * 1e-3
        cells = pairs_nw * L * L
        line["nw"] = {"workload": "similarityNW BLOSUM62 go=10 ge=4, same %d 20-mers, dense f64 NxN in HBM" % n,
                      "value": pairs_nw / t_nw, "unit": "pairs/s", "ms": t_nw * 1e3, "gcups": cells / t_nw / 1e9,
                      "hbm_GBs": (n * L + n * n * 8) / t_nw / 1e9, "hbm_frac": (n * L + n * n * 8) / t_nw / 1e9 / HBM_PEAK_GBS}

    # ---- CPU oracle on this box's host cores (baseline only)
    if rank == 0 and world == 1 and not a.no_cpu:
        import subprocess
        cb = cpu_baseline(res, off, seeds, a.cpu_seconds)
        line["cpu_baseline"] = {
            "value": cb["mh_rate"], "unit": "pairs/s", "cores": cb["threads"], "kind": "port",
            "sample": "CPU oracle (C restatement, reference loop nest + its 2 OpenMP sites) similarityMH k=4 n_hash=500 "
                      "on the first %d peptides of the same set: %.1f s" % (cb["mh_n"], cb["mh_dt"])}
        rows = 100
        code = cb["nw_code"] % (ROOT, os.path.join(ROOT, "tests"), "h3n2_like" if a.workload == "h3n2like" else "uniform_peptides", rows, rows)
        # calibrate rows for ~cpu_seconds of single-thread work (about 16 us per pair)
        rows = max(20, int(a.cpu_seconds / (4000 * 16e-6)))
        code = cb["nw_code"] % (ROOT, os.path.join(ROOT, "tests"), "h3n2_like" if a.workload == "h3n2like" else "uniform_peptides", rows, rows)
        r = json.loads(subprocess.check_output([sys.executable, "-c", code]).decode().strip().splitlines()[-1])
        line["cpu_baseline"]["nw"] = {"value": r["pairs"] / r["dt"], "unit": "pairs/s", "cores": 1,
                                      "sample": "CPU oracle similarityNW rows 0..%d x 4000 peptides, 1 thread "
                                                "(the reference's NW loop is serial): %.1f s" % (rows, r["dt"])}
        line["speedup_vs_cpu"] = {"mh": value / cb["mh_rate"]}
        if "nw" in line:
            line["speedup_vs_cpu"]["nw"] = line["nw"]["value"] / line["cpu_baseline"]["nw"]["value"]

    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
