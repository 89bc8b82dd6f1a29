#!/usr/bin/env python3
"""bench.py -- sequence-pairs/s of the all-pairs similarity hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full pass of similarityMH(k=4, n_hash=500) over the workload with the
packed residues already resident in HBM: signature build (K1) + all-pairs compare (K2)
producing the dense float64 N x N matrix in HBM (what the reference returns to R).  With
N > 1 ranks the pair space is row-sharded (cyclic tile rows), each rank's compact block is
exchanged with ONE RCCL all-gather and mirrored/widened to the full float64 matrix on every
rank; total work is fixed, so `scaling` is "strong".

Headline workload (BASELINE.json configs[3], the one the metric is quoted on):
100 000 h3n2-like 20-mers, MinHash k=4 n_hash=500, hash seed 12345.  The same JSON line also
carries the NW BLOSUM62 figure (`nw`), the dominant kernel's roofline position (`roofline`)
and the CPU oracle timed on this box's host cores (`cpu_baseline`, rank 0, N=1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
BITOP3_PEAK = 62.0e12       # lane-v_bitop3/s this chip sustains in isolation (tools/ubench/inst_rate, profiles/)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=100000, help="number of peptides (default: the 100k headline set)")
    ap.add_argument("--workload", default="h3n2like", choices=["h3n2like", "uniform"])
    ap.add_argument("--no-nw", action="store_true", help="skip the similarityNW measurement")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU oracle baseline")
    ap.add_argument("--no-edges", action="store_true", help="skip the threshold + edge-list measurement")
    ap.add_argument("--plane-bits", type=int, default=0, choices=[0, 12, 16, 32],
                    help="0: compare the signatures' exact dictionary codes with as few bit planes as the data needs "
                         "(default); 12 / 16: at least that many code planes; 32: raw signature bits")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time per baseline leg")
    return ap.parse_args()


_CPU_LEG = r"""
import json, os, sys, time
kind, n, rows, gen, root = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
if kind == "nw":
    os.environ["OMP_NUM_THREADS"] = "1"   # the reference's NW loop is serial (src/pairwiseSeqAlign.cpp:340-352)
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, oracle_lib as O
from dynaalign_amd import synth
res, off = getattr(synth, gen)(n, 20)
seqs = synth.to_strings(res, off)
t = time.perf_counter()
if kind == "mh":
    rc, M = O.similarity_mh(seqs, 4, 500, O.seeds(12345, 500))
    pairs = n * (n - 1) // 2
else:
    rc, mt, ln, sc, _ = O.nw_rows(seqs, 0, rows)
    pairs = rows * n
dt = time.perf_counter() - t
assert rc == 0
print(json.dumps({"pairs": pairs, "dt": dt, "threads": O.num_threads()}))
"""


def cpu_threads():
    """Host threads for the CPU baseline: the box's CPU share (gpurun grants 16 per GPU), never more
    than the cores we may run on -- 128 OpenMP threads on a 16-CPU quota thrash instead of compute."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, int(os.environ.get("DYNAALIGN_CPU_THREADS", "16"))))


def cpu_leg(kind, n, rows, gen):
    t = time.perf_counter()
    env = dict(os.environ, OMP_NUM_THREADS=str(cpu_threads()), OMP_WAIT_POLICY="passive")
    out = subprocess.check_output([sys.executable, "-c", _CPU_LEG, kind, str(n), str(rows), gen, ROOT], timeout=240,
                                  env=env)
    r = json.loads(out.decode().strip().splitlines()[-1])
    print("[bench] cpu leg %s n=%d rows=%d: %.1f s in the oracle, %.1f s wall" % (kind, n, rows, r["dt"],
                                                                               time.perf_counter() - t), file=sys.stderr)
    return r


def cpu_baseline(gen, target_s):
    """The CPU oracle (kind 'port': the repo's C restatement -- reference loop nest, its two OpenMP
    sites for MH, serial NW) on a bounded sample of the same workload, in a child process."""
    probe = cpu_leg("mh", 1500, 0, gen)
    rate = probe["pairs"] / probe["dt"]
    ns = int(min(32000, max(2000, (2 * rate * target_s) ** 0.5)))
    mh = cpu_leg("mh", ns, 0, gen)
    rows = max(20, int(target_s / (4000 * 2.5e-6)))      # ~2.5 us per 20-mer pair on one core
    nw = cpu_leg("nw", 4000, min(rows, 4000), gen)
    return {
        "value": mh["pairs"] / mh["dt"], "unit": "pairs/s", "cores": mh["threads"], "kind": "port",
        "sample": "CPU oracle (C restatement: reference loop nest + its 2 OpenMP sites) similarityMH k=4 n_hash=500 on the "
                  "first %d peptides of the same generator: %.1f s" % (ns, mh["dt"]),
        "nw": {"value": nw["pairs"] / nw["dt"], "unit": "pairs/s", "cores": 1,
               "sample": "CPU oracle similarityNW BLOSUM62/10/4, %d rows x 4000 peptides, 1 thread (the reference's NW loop "
                         "is serial): %.1f s" % (min(rows, 4000), nw["dt"])},
    }


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*pmc*.json):
    WRITE_SIZE + 2 x FETCH_SIZE KiB (the gfx950 FETCH_SIZE half-count correction of MI355X_MICROARCH.md)."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if f.endswith(".json") and "pmc" in f:
            try:
                d = json.load(open(os.path.join(pdir, f)))
            except Exception:
                continue
            k = d.get("kernels", {}).get(kernel)
            if k and "WRITE_SIZE" in k and "FETCH_SIZE" in k:
                best = {"bytes": (k["WRITE_SIZE"] + 2.0 * k["FETCH_SIZE"]) * 1024.0, "source": "profiles/" + f,
                        "n": d.get("n")}
    return best


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and a.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if world > 1:
        dist.barrier()
    import dynaalign_amd as da
    from dynaalign_amd import _capi, device, sharding, synth

    n, k, n_hash, L = a.n, 4, 500, 20
    gen_name = "h3n2_like" if a.workload == "h3n2like" else "uniform_peptides"
    res, off = getattr(synth, gen_name)(n, L)
    seeds = da.hash_family_seeds(12345, n_hash)
    ds = device.DeviceSequences(res, off, "cuda")
    d_seeds = torch.from_numpy(seeds.view(np.int32).copy()).cuda()
    sig = torch.empty((n, device.sig_ld(n_hash)), dtype=torch.int32, device="cuda")
    planes = torch.empty(device.planes_words(n, n_hash), dtype=torch.int32, device="cuda")
    pwork = torch.empty(device.planes_workspace_bytes(n, n_hash), dtype=torch.uint8, device="cuda")
    state = {"bits": 32}

    def signatures_and_planes(e=None):
        """K1 (signatures) + K1b (exact dictionary codes -> 8 / 12 / 16 bit planes per 32 hash functions);
        --plane-bits 32 keeps the raw signature bits, 12 / 16 set a lower bound on the code planes."""
        device.minhash_signatures(ds, k, n_hash, d_seeds, out=sig, want_planes=False)
        if e is not None:
            e.record()
        pl = device.mh_planes(sig, n, n_hash, planes, pwork, a.plane_bits)
        state["bits"] = pl.bits
        return pl
    out = torch.empty((n, n), dtype=torch.float64, device="cuda")

    pairs_mh = n * (n - 1) // 2            # unordered pairs, diagonal excluded (src/minHash.cpp:164)
    pairs_nw = n * (n + 1) // 2            # the reference computes the NW diagonal (src/pairwiseSeqAlign.cpp:342)

    ev = lambda: torch.cuda.Event(enable_timing=True)

    if world == 1:
        def step():
            e = [ev() for _ in range(4)]
            e[0].record()
            pl = signatures_and_planes(e[1])
            e[2].record()
            device.mh_compare(pl, n, n_hash, 0, n, True, _capi.DA_OUT_F64, out=out)
            e[3].record()
            return e
        phase_names = ["k1_signatures", "k1b_codes_to_planes", "k2_compare"]
    else:
        plan = sharding.Plan(n, rank, world, sharding.MH_TILE)
        work = sharding.PackedWorkspace(plan, n_hash, "cuda")   # counts travel in bits(n_hash) = 9 bits, not 16

        def step():
            e = [ev() for _ in range(6)]
            e[0].record()
            pl = signatures_and_planes(e[1])                      # every rank: all signatures (2 MB in)
            e[2].record()
            sharding.mh_local_block(plan, work, pl, n_hash)
            sharding.pack_local_block(plan, work)
            e[3].record()
            sharding.all_pairs_sharded(plan, work.packed, work.gathered, lambda gathered: gathered)
            e[4].record()
            sharding.finalize_shards_packed(plan, work, work.gathered, n_hash, out)
            e[5].record()
            return e
        phase_names = ["k1_signatures", "k1b_codes_to_planes", "k2_compare_shard", "all_gather", "finalize"]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    evs = [step() for _ in range(a.steps)]
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    phases = {nm: float(np.mean([e[i].elapsed_time(e[i + 1]) for e in evs])) for i, nm in enumerate(phase_names)}
    ms_per_step = dt / a.steps * 1e3
    value = pairs_mh / (dt / a.steps)

    # ---- roofline of the dominant kernel (K2 compare); duration from HIP events on the launch stream
    k2_key = "k2_compare" if world == 1 else "k2_compare_shard"
    k2 = phases[k2_key] * 1e-3
    plane_bits = state["bits"]
    planes_row_bytes = 2 * 16 * plane_bits * 4        # two copies x 16 groups x planes x 4 B
    T = (n + 127) // 128
    if world == 1:
        bytes_k2 = n * planes_row_bytes + n * n * 8   # read the bit planes once + write the f64 N x N
        tiles = T * (T + 1) // 2
    else:
        tiles = sum(T - t for t in range(rank, T, world))
        bytes_k2 = n * planes_row_bytes + tiles * 128 * 128 * 2   # this rank's uint16 tiles
    lane_ops = tiles * 128 * 128 * 16 * plane_bits    # one v_bitop3 per pair and bit plane (16 groups x 8..32 planes)
    # 12 code planes, symmetric mode: the hand-scheduled kernel does all but the diagonal / border tiles
    k2_name = "k_mh_compare_a12<true>" if (world == 1 and plane_bits == 12) else "k_mh_compare<true, true, %d>" % plane_bits
    traffic = pmc_traffic(k2_name) if world == 1 else None
    roof = {"kernel": k2_name if world == 1 else "k_mh_compare", "bound": "hbm", "achieved": bytes_k2 / k2 / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": bytes_k2 / k2 / 1e9 / HBM_PEAK_GBS,
            "traffic": traffic["bytes"] if traffic and traffic.get("n") == n else None,
            "traffic_source": traffic["source"] if traffic and traffic.get("n") == n else None,
            "avg_launch_ms": k2 * 1e3, "algorithmic_bytes_per_launch": bytes_k2, "plane_bits": plane_bits,
            "valu": {"note": "the unit that actually binds: bit-sliced compare = 1 v_bitop3 per pair per bit plane; "
                             "peak = isolated v_bitop3 issue rate measured on this chip",
                     "lane_ops_per_launch": lane_ops, "achieved_lane_ops_per_s": lane_ops / k2,
                     "peak_lane_ops_per_s": BITOP3_PEAK, "frac": lane_ops / k2 / BITOP3_PEAK}}

    line = {
        "metric": "sequence-pairs/sec (MinHash k=4 n_hash=500; NW BLOSUM62) at 1/2/4/8 MI355X",
        "value": value, "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": "similarityMH k=4 n_hash=500 on %d %s 20-mers (hash seed 12345), dense f64 NxN in HBM"
                               % (n, "h3n2-like" if a.workload == "h3n2like" else "uniform"),
                   "n": n, "k": k, "n_hash": n_hash, "pairs": pairs_mh,
                   "sharding": "1 GPU: upper-triangle tiles + mirrored store" if world == 1
                   else "cyclic tile rows over %d ranks, one RCCL all-gather of the 9-bit packed counts, mirror+widen on every rank" % world},
        "roofline": roof,
        "phases_ms": phases,
    }

    # ---- similarityNW on the same set (second half of the metric); one timed launch
    if not a.no_nw:
        bad = device.nw_encode(ds)
        assert int(bad.item()) == 0
        if world == 1:
            run_nw = lambda: device.nw(ds, "BLOSUM62", 10, 4, 0, n, True, _capi.DA_OUT_F64, out=out)
        else:
            nplan = sharding.Plan(n, rank, world, sharding.NW_TILE)
            nwork = sharding.Workspace(nplan, "cuda")
            run_nw = lambda: sharding.nw_sharded_step(nplan, nwork, ds, out)
        run_nw()
        sync()
        t0 = time.perf_counter()
        run_nw()
        sync()
        t_nw = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([t_nw], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t_nw = float(t.item())
        cells = pairs_nw * L * L
        line["nw"] = {"workload": "similarityNW BLOSUM62 go=10 ge=4, same %d 20-mers, dense f64 NxN in HBM" % n,
                      "value": pairs_nw / t_nw, "unit": "pairs/s", "ms": t_nw * 1e3, "gcups": cells / t_nw / 1e9,
                      "hbm_GBs": (n * L + n * n * 8) / t_nw / 1e9, "hbm_frac": (n * L + n * n * 8) / t_nw / 1e9 / HBM_PEAK_GBS}

    # ---- similarityMH + clusterbreak's quantile threshold as a distributed edge list:
    # shard compare -> histogram -> ONE all-reduce of n_hash+1 words -> exact type-7 quantile -> local edges.
    # No N x N exchange, so this is the variant of the path whose whole-job time scales with the rank count.
    if not a.no_edges:
        if world == 1:
            # one GPU: nothing to shard -- symmetric uint16 compare, histogram of the upper triangle, exact quantile, edges
            cnt16 = out.view(-1).view(torch.int16)[:n * n].view(n, n)          # the f64 result buffer is free here: reuse its first 20 GB
            values = np.arange(n_hash + 1, dtype=np.float64) / n_hash

            def run_edges():
                pl = signatures_and_planes()
                device.mh_compare(pl, n, n_hash, 0, n, True, _capi.DA_OUT_COMPACT, out=cnt16)
                h = device.upper_histogram(cnt16, n, n_hash + 1).cpu().numpy().astype(np.uint64)
                thr = da.quantile_type7(h, values, 0.8)
                keep = (~(values < thr)) & (np.arange(n_hash + 1) != 0)
                m = int(h[keep].sum()) + n
                ei, ej, evv, c = device.extract_edges(cnt16, n, keep, m)
                return thr, ei, ej, evv, c
        else:
            eplan = sharding.Plan(n, rank, world, sharding.MH_TILE)

            def run_edges():
                return sharding.mh_edges_sharded(eplan, work, signatures_and_planes(), n_hash, 0.8)
        run_edges()
        sync()
        t0 = time.perf_counter()
        thr, ei, ej, evv, cnt = run_edges()
        sync()
        t_e = time.perf_counter() - t0
        tot = cnt.clone()
        if world > 1:
            t = torch.tensor([t_e], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t_e = float(t.item())
            dist.all_reduce(tot)
        line["edges"] = {"workload": "similarityMH k=4 n_hash=500 + quantile(S[upper.tri(S)], 0.8) threshold -> edge list "
                                     "(R/clusterbreak.R:219-221), same %d peptides; %s" % (n, "one GPU: symmetric uint16 compare -> histogram -> "
                                     "quantile -> edges" if world == 1 else "edges stay distributed over the ranks"),
                         "value": pairs_mh / t_e, "unit": "pairs/s", "ms": t_e * 1e3, "threshold": thr,
                         "edges_total": int(tot.item()), "edge_list_bytes": int(tot.item()) * 10}
        del ei, ej, evv

    # ---- CPU oracle on this box's host cores (baseline only; rank 0, N = 1)
    if rank == 0 and world == 1 and not a.no_cpu:
        cb = cpu_baseline(gen_name, a.cpu_seconds)
        line["cpu_baseline"] = cb
        line["speedup_vs_cpu"] = {"mh": value / cb["value"]}
        if "nw" in line:
            line["speedup_vs_cpu"]["nw"] = line["nw"]["value"] / cb["nw"]["value"]

    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
