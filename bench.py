#!/usr/bin/env python3
"""bench.py -- sequence-pairs/s of the all-pairs similarity hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full pass of similarityMH(k=4, n_hash=500) over the workload with the packed residues already
resident in HBM -- ONE C call, da_dev_similarity_mh: duplicate plan + signature build (K1) + exact re-coding (K1b) + all-pairs
compare (K2) [+ index expansion when byte-identical sequences were collapsed] producing the dense float64
N x N matrix in HBM (what the reference returns to R).  `value` is on that boundary -- T_k of SURVEY 8(d): kernels
only, inputs and result resident in HBM.  With N > 1 ranks the pair space is row-sharded (cyclic tile rows), each
rank's compact block is exchanged with ONE RCCL all-gather and mirrored / widened to the full float64 matrix on every
rank (T_g); total work is fixed, so `scaling` is "strong".

Headline workload (BASELINE.json configs[3], the one the metric is quoted on): 100 000 h3n2-like 20-mers, MinHash
k=4 n_hash=500, hash seed 12345.  The same JSON line also carries
    roofline      the dominant kernel of the timed step: algorithmic bytes / HIP-event duration vs the 8 TB/s HBM peak
                  (k_expand_rows when the duplicate-collapsing route runs, with K2 on the unique rows beside it; K2 + its VALU bound otherwise)
    direct        the same input with the route off (K2 on all N rows) and K2's roofline object
    nw            similarityNW BLOSUM62/10/4 on the same set (second half of the metric) with its own roofline object
    uniform       the other SURVEY 8(d) workload, S100k uniform (16 code planes instead of 12: k_mh_compare_a16)
    t_h           the host-pointer boundary (what R sees): da_similarity_mh / _nw into a pageable host matrix, PCIe-inclusive
    edges         similarityMH + clusterbreak's quantile threshold as an edge list
    clusterbreak  BASELINE configs[4]: clusterbreak(size_max=800, thresh_p=.8) end to end on the device edge path
    cpu_baseline  the CPU oracle timed on this box's host cores (rank 0, N=1 only) and speedups on the T_h boundary
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
VALU_PEAK = 256 * 4 * 32 * 2.4e9   # lane-ops/s: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (MI355X_MICROARCH.md chip table)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=int(os.environ.get("DYNAALIGN_BENCH_N", "100000")),
                    help="number of peptides (default: the 100k headline set; DYNAALIGN_BENCH_N sets it where a launcher eats --n)")
    ap.add_argument("--workload", default="h3n2like", choices=["h3n2like", "uniform"])
    ap.add_argument("--no-nw", action="store_true", help="skip the similarityNW measurement")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU oracle baseline")
    ap.add_argument("--no-edges", action="store_true", help="skip the threshold + edge-list measurement")
    ap.add_argument("--no-uniform", action="store_true", help="skip the uniform-peptide leg")
    ap.add_argument("--no-host", action="store_true", help="skip the host-pointer (T_h, PCIe-inclusive) measurement")
    ap.add_argument("--no-clusterbreak", action="store_true", help="skip the clusterbreak end-to-end run (BASELINE config 5)")
    ap.add_argument("--plane-bits", type=int, default=0, choices=[0, 12, 14, 15, 16, 32],
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
if kind in ("mh", "mh_rowptr"):
    rc, M = O.similarity_mh(seqs, 4, 500, O.seeds(12345, 500), rowptr=(kind == "mh_rowptr"))
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
    sites for MH, serial NW) on a bounded sample of the same workload, in a child process.  `value` is the variant SURVEY 8(d)
    specifies -- the reference's data structures (row-pointer signature storage, copied k-mers, column-major element stores:
    orc_similarity_mh_rowptr); the flat-array port is reported beside it."""
    probe = cpu_leg("mh", 1500, 0, gen)
    rate = probe["pairs"] / probe["dt"]
    ns = int(min(32000, max(2000, (rate * target_s) ** 0.5)))   # two MinHash legs share the budget
    mh_flat = cpu_leg("mh", ns, 0, gen)
    mh = cpu_leg("mh_rowptr", ns, 0, gen)
    rows = max(20, int(target_s / (4000 * 2.5e-6)))      # ~2.5 us per 20-mer pair on one core
    nw = cpu_leg("nw", 4000, min(rows, 4000), gen)
    return {
        "value": mh["pairs"] / mh["dt"], "unit": "pairs/s", "cores": mh["threads"], "kind": "port",
        "variant": "reference-structured (SURVEY 8(d)): vector<vector>-style row pointers, copied k-mers, column-major M(i,j) stores, -O2 -fopenmp",
        "sample": "CPU oracle (C restatement: reference loop nest + its 2 OpenMP sites, reference data structures) similarityMH k=4 n_hash=500 on the "
                  "first %d peptides of the same generator, whole call incl. the f64 matrix fill: %.1f s" % (ns, mh["dt"]),
        "flat_port": {"value": mh_flat["pairs"] / mh_flat["dt"], "unit": "pairs/s", "cores": mh_flat["threads"],
                      "sample": "the same call on flat arrays (orc_similarity_mh), same %d peptides: %.1f s" % (ns, mh_flat["dt"])},
        "nw": {"value": nw["pairs"] / nw["dt"], "unit": "pairs/s", "cores": 1,
               "sample": "CPU oracle similarityNW BLOSUM62/10/4, %d rows x 4000 peptides, 1 thread (the reference's NW loop "
                         "is serial): %.1f s" % (min(rows, 4000), nw["dt"])},
    }


def kernel_source_hash():
    """identifies the build a profile was taken on: sha256 over the kernel sources (tools/source_hash.py; the GPU box has no .git)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import source_hash
    return source_hash.source_hash()


def pmc_kernel(kernel, n):
    """per-launch counter averages of `kernel` from the newest committed rocprofv3 PMC summary (profiles/*pmc*.json)
    taken at the same N, + the file they come from and the source hash of the build they were measured on"""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if f.endswith(".json") and "pmc" in f:
            try:
                d = json.load(open(os.path.join(pdir, f)))
            except Exception:
                continue
            ks = d.get("kernels", {})
            k = ks.get(kernel)
            if k is None:                        # template instances are profiled under their full name ("k_expand_stream<true, 6>")
                inst = sorted(nm for nm in ks if nm.startswith(kernel + "<"))
                k = ks[inst[0]] if len(inst) == 1 else None
            if k and d.get("n") == n:
                best = dict(k, source="profiles/" + f, source_hash=d.get("source_hash"))
    return best


def pmc_traffic(kernel, n):
    """HBM bytes per launch: WRITE_SIZE + 2 x FETCH_SIZE KiB (the gfx950 FETCH_SIZE half-count correction of
    MI355X_MICROARCH.md)"""
    k = pmc_kernel(kernel, n)
    if k and "WRITE_SIZE" in k and "FETCH_SIZE" in k:
        cur = kernel_source_hash()
        src = "%s (kernel sources %s)" % (k["source"], k.get("source_hash") or "unrecorded")
        if k.get("source_hash") != cur:      # measured on other kernels than the ones running now: not this build's traffic
            return {"bytes": None, "source": src + " -- STALE: the running sources are %s, traffic not reported" % cur}
        return {"bytes": (k["WRITE_SIZE"] + 2.0 * k["FETCH_SIZE"]) * 1024.0, "source": src}
    return None


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_devices():
    """number of GPUs this process could open; torch.cuda.device_count() does not initialise the GPU on this image, so the
    launching parent may ask"""
    import torch
    return torch.cuda.device_count()


def launch_ranks(n_ranks, argv, child_cmd=None, device_count=visible_devices, timeout_s=None, out=None, err=None):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start the N ranks ourselves.

    The parent never touches a GPU (build() is CPU-only; counting devices does not initialise HIP here) and never replaces
    itself: it starts N fresh child processes -- one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 /
    MASTER_PORT set, the same arguments -- relays rank 0's stdout (its last JSON line is the parent's last stdout line) and the
    other ranks' stderr, and returns non-zero when ANY child does, when a child outlives `timeout_s`, or when rank 0 printed no
    JSON line.  With the nccl (= RCCL) backend fewer than N visible devices is an error: never a silent fall-back to fewer GPUs.
    Returns the exit code."""
    import threading
    out = out or sys.stdout
    err = err or sys.stderr
    backend = os.environ.get("DYNAALIGN_BENCH_BACKEND", "nccl")
    if n_ranks < 2:
        raise ValueError("launch_ranks is for N > 1")
    if backend == "nccl":
        ndev = device_count()
        if ndev < n_ranks:
            print("[bench] --gpus %d needs %d visible GPUs with the RCCL backend, this node shows %d: not starting "
                  "(DYNAALIGN_BENCH_BACKEND=gloo rehearses the code path with ranks sharing the visible devices; it is not a measurement)"
                  % (n_ranks, n_ranks, ndev), file=err)
            return 3
    if timeout_s is None:
        timeout_s = float(os.environ.get("DYNAALIGN_BENCH_LAUNCH_TIMEOUT", "2400"))
    cmd = list(child_cmd) if child_cmd else [sys.executable, os.path.abspath(__file__)]
    port = free_port()
    procs, pumps, rank0_lines = [], [], []

    def pump(stream, sink, keep=None, prefix=""):
        for raw in iter(stream.readline, b""):
            text = raw.decode("utf-8", "replace")
            if keep is not None:
                keep.append(text)
            else:
                sink.write(prefix + text)
                sink.flush()
        stream.close()

    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DYNAALIGN_BENCH_BUILT="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        p = subprocess.Popen(cmd + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                             stderr=subprocess.PIPE)
        procs.append(p)
        if r == 0:
            pumps.append(threading.Thread(target=pump, args=(p.stdout, None, rank0_lines), daemon=True))
        pumps.append(threading.Thread(target=pump, args=(p.stderr, err, None, "" if r == 0 else "[rank %d] " % r), daemon=True))
    for t in pumps:
        t.start()

    def stop_all():
        for p in procs:                      # exactly the processes started above, by handle
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()

    deadline = time.monotonic() + timeout_s
    rc = 0
    pending = set(range(n_ranks))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if 0 < code < 256 else 1
                    print("[bench] rank %d exited with %d: stopping the other ranks" % (r, code), file=err)
                    stop_all()
        if pending and time.monotonic() > deadline:
            print("[bench] ranks %s still running after %.0f s: stopping them" % (sorted(pending), timeout_s), file=err)
            stop_all()
            rc = rc or 124
            break
        if pending:
            time.sleep(0.05)
    for p in procs:
        p.wait()
    for t in pumps:
        t.join(5)
    json_line = None
    for text in rank0_lines:
        s = text.strip()
        if s.startswith("{") and s.endswith("}"):
            json_line = s
        else:
            out.write(text)
    if rc == 0 and json_line is None:
        print("[bench] rank 0 printed no result line", file=err)
        rc = 1
    if json_line is not None and rc == 0:
        out.write(json_line + "\n")
    elif json_line is not None:
        print("[bench] a rank failed; rank 0's line is NOT a result: " + json_line[:400], file=err)
    out.flush()
    return rc


def main():
    a = parse()
    import __graft_entry__ as g
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # no launcher around us: become one (never a silent 1-GPU run under `--gpus N`)
        g.build()
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if local_rank == 0 and not os.environ.get("DYNAALIGN_BENCH_BUILT"):
        g.build()                       # a no-op when the in-tree .so files are current; before anything touches the GPU
    import datetime
    import numpy as np
    import torch
    import torch.distributed as dist

    # DYNAALIGN_BENCH_BACKEND=gloo: rehearsal of the N > 1 code path on a box with fewer GPUs than ranks (the ranks share
    # the visible devices, the exchange goes through the host) -- never a measurement
    backend = os.environ.get("DYNAALIGN_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < 1 or (backend == "nccl" and world > 1 and local_rank >= ndev):
        raise SystemExit("[bench] rank %d (local %d of %d): %d visible GPUs -- a rank per GPU is required with the RCCL backend"
                         % (rank, local_rank, world, ndev))
    torch.cuda.set_device(local_rank % ndev if backend != "nccl" else local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a finite rendezvous / collective timeout: a rank that never arrives or an RCCL error ends the run with a non-zero exit
        # (torch's watchdog aborts the process on a timed-out or failed collective), not with a hang
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        tmo = datetime.timedelta(seconds=float(os.environ.get("DYNAALIGN_BENCH_DIST_TIMEOUT", "300")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
        if dist.get_world_size() != a.gpus:
            raise SystemExit("--gpus %d but the process group has %d ranks" % (a.gpus, dist.get_world_size()))
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

    def signatures_and_planes(e=None, seqs=ds, min_bits=a.plane_bits):
        """K1 (signatures) + K1b (exact dictionary codes -> 8 / 12 / 16 bit planes per 32 hash functions);
        --plane-bits 32 keeps the raw signature bits, 12 / 16 set a lower bound on the code planes."""
        device.minhash_signatures(seqs, k, n_hash, d_seeds, out=sig, want_planes=False)
        if e is not None:
            e.record()
        pl = device.mh_planes(sig, n, n_hash, planes, pwork, min_bits)
        state["bits"] = pl.bits
        return pl
    out = torch.empty((n, n), dtype=torch.float64, device="cuda")
    cnt16 = None

    pairs_mh = n * (n - 1) // 2            # unordered pairs, diagonal excluded (src/minHash.cpp:164)
    pairs_nw = n * (n + 1) // 2            # the reference computes the NW diagonal (src/pairwiseSeqAlign.cpp:342)

    ev = lambda: torch.cuda.Event(enable_timing=True)

    if world == 1:
        if a.plane_bits:
            os.environ["DYNAALIGN_PLANE_BITS"] = str(a.plane_bits)      # the one-call entry point reads the lower bound from here

        def step(seqs=ds):
            """ONE C call (da_dev_similarity_mh): plan (duplicate collapse) + K1 + K1b + K2 [+ column gather + expansion];
            it synchronises the stream, so the route's HIP-event phase times are final when it returns"""
            device.similarity_mh(seqs, k, n_hash, d_seeds, out=out)
            r = device.mh_last_route()
            state["bits"] = r["plane_bits"]
            return r
        phase_names = ["plan_ms", "codes_ms", "k2_ms", "gather_ms", "expand_ms", "border_ms"]
    else:
        # every rank builds the same duplicate plan from the input it holds anyway (deterministic: no exchange); when collapsing pays,
        # the ranks shard the count table of the UNIQUE strings and the one all-gather moves (U/n)^2 of the bytes
        probe = device.UniquePlan(ds.residues, ds.offsets, n, ds.total)
        state["dedup"] = sharding.dedup_worth(n, probe.unique, False, n_hash)
        state["unique"] = probe.unique
        del probe
        plan = sharding.Plan(n, rank, world, sharding.MH_TILE)
        work = None if state["dedup"] else sharding.PackedWorkspace(plan, n_hash, "cuda")   # counts travel in bits(n_hash) = 9 bits, not 16

        def step_dedup(seqs=ds):
            e = [ev() for _ in range(6)]
            e[0].record()
            up = device.UniquePlan(seqs.residues, seqs.offsets, seqs.n, seqs.total)
            e[1].record()
            uplan_, uwork = sharding.mh_unique_local(up, seqs, k, n_hash, d_seeds, rank, world)   # K1 + K1b + this rank's K2 tiles + pack, unique strings
            e[2].record()
            sharding.gather_blocks(uwork.gathered, uwork.packed)
            e[3].record()
            table = device.shards_to_table(uwork.gathered, 0, up.unique, world, uwork.bits)
            e[4].record()
            device.expand_unique(table, up, False, n_hash, 0, out)
            e[5].record()
            state["block_bytes"] = uwork.block_bytes
            return e

        def step_direct(seqs=ds):
            e = [ev() for _ in range(6)]
            e[0].record()
            pl = signatures_and_planes(e[1], seqs)                # every rank: all signatures (2 MB in)
            e[2].record()
            sharding.mh_local_block(plan, work, pl, n_hash)
            sharding.pack_local_block(plan, work)
            e[3].record()
            sharding.all_pairs_sharded(plan, work.packed, work.gathered, lambda gathered: gathered)
            e[4].record()
            sharding.finalize_shards_packed(plan, work, work.gathered, n_hash, out)
            e[5].record()
            state["block_bytes"] = work.block_bytes
            return e
        if state["dedup"]:
            step = step_dedup
            phase_names = ["plan", "codes_and_k2_shard_on_unique", "all_gather", "shards_to_table", "expand"]
        else:
            step = step_direct
            phase_names = ["k1_signatures", "k1b_codes_to_planes", "k2_compare_shard", "all_gather", "finalize"]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    def timed_steps(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        sync()
        t0 = time.perf_counter()
        evs = [fn() for _ in range(steps)]
        sync()
        return max_over_ranks(time.perf_counter() - t0), evs

    dt, evs = timed_steps(step, a.steps, a.warmup)
    def phase_means(evs):
        if world == 1:
            return {nm: float(np.mean([e[nm] for e in evs])) for nm in phase_names}
        return {nm: float(np.mean([e[i].elapsed_time(e[i + 1]) for e in evs])) for i, nm in enumerate(phase_names)}
    phases = phase_means(evs)
    route = evs[-1] if world == 1 else None
    ms_per_step = dt / a.steps * 1e3
    value = pairs_mh / (dt / a.steps)

    def expand_roofline(rows_ms, wl_n=n):
        """roofline object of k_expand_rows (the dominant kernel when the duplicate-collapsing route runs): per interior
        off-diagonal 128 x 128 tile it reads 128 x 128 uint16 counts once and writes the tile twice as float64 (direct + mirrored)"""
        Tf = wl_n // 128
        tiles = Tf * (Tf - 1) // 2
        pairs_x = tiles * 128 * 128                      # unordered pairs this launch finishes (interior off-diagonal tiles)
        bytes_alg = pairs_x * survey_bytes_per_pair(wl_n)     # 16.08 B per unordered pair at N = 100k
        bytes_x = pairs_x * (2 + 16)                     # what the kernel itself moves: 2 B of the gathered table read + 16 B written
        t = rows_ms * 1e-3
        traffic = pmc_traffic("k_expand_rows<false>", wl_n)
        return {"kernel": "k_expand_rows<false>", "bound": "hbm", "achieved": bytes_alg / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": bytes_alg / t / 1e9 / HBM_PEAK_GBS, "frac_kernel_bytes": bytes_x / t / 1e9 / HBM_PEAK_GBS,
                "traffic": traffic["bytes"] if traffic else None,
                "traffic_source": traffic["source"] if traffic else None, "avg_launch_ms": rows_ms,
                "algorithmic_bytes_per_launch": bytes_alg, "kernel_bytes_per_launch": bytes_x, "pairs_per_launch": pairs_x,
                "note": "index expansion of the U x U count table to the dense f64 N x N; frac = SURVEY 8(d)'s algorithmic bytes per unordered pair "
                        "x the pairs of one launch / its HIP-event duration; frac_kernel_bytes counts the kernel's own 2 B read + 16 B written per pair"}

    def stream_roofline(rows_ms, unique, wl_n=n):
        """roofline object of k_expand_stream (ROW expansion of the unique strings' count table: the dominant kernel of the duplicate route): every
        element of the dense f64 matrix written once -- 16-byte streaming stores from a table row held in LDS -- no gathered copy of the table"""
        pairs_x = wl_n * (wl_n - 1) // 2                 # every unordered pair (both halves and the diagonal are written by this kernel)
        bytes_alg = pairs_x * survey_bytes_per_pair(wl_n)
        bytes_x = wl_n * wl_n * 8 + unique * ((unique + 7) // 8 * 8) * 2     # the result + the uint16 table read once (rows of strings with > 4 copies: once per 4)
        t = rows_ms * 1e-3
        traffic = pmc_traffic("k_expand_stream", wl_n)
        return {"kernel": "k_expand_stream", "bound": "hbm", "achieved": bytes_alg / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": bytes_alg / t / 1e9 / HBM_PEAK_GBS, "frac_kernel_bytes": bytes_x / t / 1e9 / HBM_PEAK_GBS,
                "traffic": traffic["bytes"] if traffic else None, "traffic_source": traffic["source"] if traffic else None,
                "avg_launch_ms": rows_ms, "algorithmic_bytes_per_launch": bytes_alg, "kernel_bytes_per_launch": bytes_x, "pairs_per_launch": pairs_x,
                "note": "row expansion of the U x U count table to the dense f64 N x N (every element written once); frac = SURVEY 8(d)'s algorithmic bytes "
                        "per unordered pair x all pairs / the kernel's HIP-event time; frac_kernel_bytes counts the result + the table read"}

    def sparse_roofline(tile_ms, incidences, wl_n=n):
        """roofline object of k_sp_tiles (the dominant kernel of the SPARSE route: inputs whose signatures rarely agree): it writes every
        element of the dense f64 matrix once from a 128 x 128 count image built in LDS out of the tile's bucket of matching incidences"""
        pairs_x = wl_n * (wl_n - 1) // 2
        bytes_alg = pairs_x * survey_bytes_per_pair(wl_n)
        bytes_x = wl_n * wl_n * 8 + incidences * 2
        t = tile_ms * 1e-3
        traffic = pmc_traffic("k_sp_tiles", wl_n)
        return {"kernel": "k_sp_tiles", "bound": "hbm", "achieved": bytes_alg / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": bytes_alg / t / 1e9 / HBM_PEAK_GBS, "frac_kernel_bytes": bytes_x / t / 1e9 / HBM_PEAK_GBS,
                "traffic": traffic["bytes"] if traffic else None, "traffic_source": traffic["source"] if traffic else None,
                "avg_launch_ms": tile_ms, "algorithmic_bytes_per_launch": bytes_alg, "kernel_bytes_per_launch": bytes_x,
                "matching_incidences": incidences,
                "note": "sparse route: the dictionary codes of K1b give the matching (pair, hash function) incidences; they are bucketed per output tile "
                        "(phases_ms.k2_ms = that bucket phase) and every tile of the f64 matrix is written once (phases_ms.expand_ms = this kernel)"}

    def k2_roofline(k2_ms, plane_bits, wl_n=n, out_elem=8):
        """roofline object of the compare kernel: algorithmic bytes (with the plane words actually read) /
        HIP-event duration, against the 8 TB/s HBM peak; + the VALU bound that actually binds"""
        k2 = k2_ms * 1e-3
        planes_row_bytes = 2 * 16 * (16 if plane_bits in (14, 15) else plane_bits) * 4   # two copies x 16 groups x planes x 4 B
        T = (wl_n + 127) // 128
        if world == 1:
            bytes_k2 = wl_n * planes_row_bytes + wl_n * wl_n * out_elem   # read the bit planes once + write the N x N (f64, or uint16 counts)
            tiles = T * (T + 1) // 2
        else:
            tiles = sum(T - t for t in range(rank, T, world))
            bytes_k2 = wl_n * planes_row_bytes + tiles * 128 * 128 * 2   # this rank's uint16 tiles
        lane_ops = tiles * 128 * 128 * 16 * plane_bits    # one v_bitop3 per pair and bit plane (16 groups x 8..32 planes)
        # 12 / 16 code planes, symmetric mode: the hand-scheduled kernels do all but the diagonal / border tiles
        f64s = "true" if out_elem == 8 else "false"
        if world == 1 and plane_bits == 8 and 32 < n_hash < 2048:   # the dense half of the heavy / rare split: <float64, planes, one tile per workgroup>
            k2_name = "k_mh_compare_p12<%s, 8, true>" % f64s
        elif world == 1 and plane_bits == 12:
            k2_name = "k_mh_compare_a12<%s>" % f64s
        elif world == 1 and plane_bits in (14, 15, 16):      # <float64 output, code bits>: 14 / 15 skip the top planes' step / half step
            k2_name = "k_mh_compare_a16<%s, %d>" % (f64s, plane_bits)
        else:
            k2_name = "k_mh_compare<%s, true, %d>" % (f64s, 16 if plane_bits in (14, 15) else plane_bits)
        traffic = pmc_traffic(k2_name, wl_n) if world == 1 else None
        pmk = pmc_kernel(k2_name, wl_n) if world == 1 else None
        # SURVEY 8(d)'s figure for one launch: all unordered pairs x 16.08 B (float64 result); for the uint16 kinds 8(d) says to
        # count N^2 x 2 for the result -- that is the kernel's own figure, used for both
        bytes_alg = (wl_n * (wl_n - 1) / 2) * survey_bytes_per_pair(wl_n) if (world == 1 and out_elem == 8) else bytes_k2
        return {"kernel": k2_name if world == 1 else "k_mh_compare", "bound": "hbm", "achieved": bytes_alg / k2 / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_alg / k2 / 1e9 / HBM_PEAK_GBS,
                "frac_kernel_bytes": bytes_k2 / k2 / 1e9 / HBM_PEAK_GBS,
                "traffic": traffic["bytes"] if traffic else None, "traffic_source": traffic["source"] if traffic else None,
                "avg_launch_ms": k2 * 1e3, "algorithmic_bytes_per_launch": bytes_alg, "kernel_bytes_per_launch": bytes_k2, "plane_bits": plane_bits,
                "valu": {"note": "the unit that actually binds: bit-sliced compare = 1 v_bitop3 per pair per bit plane; "
                                 "peak = isolated v_bitop3 issue rate measured on this chip",
                         "lane_ops_per_launch": lane_ops, "achieved_lane_ops_per_s": lane_ops / k2,
                         "peak_lane_ops_per_s": BITOP3_PEAK, "frac": lane_ops / k2 / BITOP3_PEAK,
                         # the compare kernels are POWER-limited: with the float64 stores the shader clock drops to ~1.8 GHz (2.2 without;
                         # profiles/r03_i_k2_effective_clock.txt).  GRBM_GUI_ACTIVE of the committed PMC profile / 8 XCDs / that profile's launch time:
                         **({"effective_clock_hz_in_profile": pmk["GRBM_GUI_ACTIVE"] / 8.0 / (pmk["duration_ms"] * 1e-3),
                             "clock_source": pmk["source"]} if pmk and "GRBM_GUI_ACTIVE" in pmk and pmk.get("duration_ms") else {})}}

    def survey_bytes_per_pair(wl_n):
        """(N L + 2 x N n_hash 4 + N^2 8) bytes per call / N (N - 1) / 2 unordered pairs"""
        return (wl_n * L + 2 * wl_n * n_hash * 4 + wl_n * wl_n * 8) / (wl_n * (wl_n - 1) / 2)

    k2_key = "k2_ms" if world == 1 else "k2_compare_shard"
    if world > 1 and state.get("dedup"):
        main_roof = expand_roofline(phases["expand"])
        main_roof["note"] += "; here avg_launch_ms is the whole expansion call (column gather + k_expand_rows + border tiles)"
    elif world == 1 and route.get("sparse"):
        main_roof = sparse_roofline(phases["expand_ms"], route["sparse_pairs"])
    elif world == 1 and route["dedup"]:
        # the timed step ran on the table of unique strings: its dominant kernel is the expansion; K2 on U rows rides along
        rows_form = route.get("expansion", "").startswith("rows")
        main_roof = stream_roofline(phases["expand_ms"], route["unique"]) if rows_form else expand_roofline(phases["expand_ms"])
        if route.get("pipelined"):
            # the table's compare runs on a side stream under the expansion (api.cpp "PIPELINED form"): k_expand_rows is launched once per chunk,
            # on two alternating streams; expand_ms is the time SOME launch of it was running (union of the launches' HIP-event intervals)
            el = max(1, route.get("expand_launches", 1))
            main_roof["launches_per_step"] = el
            main_roof["time_in_kernel_ms"] = main_roof["avg_launch_ms"]
            main_roof["avg_launch_ms"] = main_roof["time_in_kernel_ms"] / el
            for key in ("algorithmic_bytes_per_launch", "kernel_bytes_per_launch", "pairs_per_launch"):
                main_roof[key.replace("_per_launch", "_per_step")] = main_roof[key]
                main_roof[key] = main_roof[key] / el
            if main_roof["traffic"] is not None:         # the PMC passes run the one-stream form (tools/prof_pmc.sh): one launch there = all launches here
                main_roof["traffic_per_step"] = main_roof["traffic"]
                main_roof["traffic"] = main_roof["traffic"] / el
            main_roof["note"] += ("; PIPELINED: %d chunk launches per step (mean size reported per launch; frac = bytes of all launches / the time some launch was "
                                  "running), co-running with the compare of the unique table (phases_ms.k2_ms = its span on the side stream: the first bands at "
                                  "full occupancy, then one persistent workgroup per CU)%s: they overlap; "
                                  "`one_stream` below is the same input with DYNAALIGN_MH_NO_PIPE=1 (each kernel alone)"
                                  % (el, "" if rows_form else " and the column gathers (phases_ms.gather_ms = sum of the launches incl. their wait for LDS)"))
        else:
            main_roof["k2_on_unique"] = k2_roofline(phases["k2_ms"], state["bits"], route["unique"], 2)
    else:
        main_roof = k2_roofline(phases[k2_key], state["bits"])
    # the whole step against the roofline: SURVEY 8(d)'s bytes of ONE call (8.04e10 at N = 100k) / the step's wall time
    step_bytes = n * L + 2 * n * n_hash * 4 + n * n * 8
    main_roof["step_frac"] = step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS
    main_roof["step_algorithmic_bytes"] = step_bytes
    line = {
        "metric": "sequence-pairs/sec (MinHash k=4 n_hash=500; NW BLOSUM62) at 1/2/4/8 MI355X",
        "value": value, "unit": "pairs/s", "n_gpus": dist.get_world_size() if world > 1 else 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic" if os.environ.get("DYNAALIGN_BENCH_BACKEND", "nccl") == "nccl" else "synthetic; REHEARSAL over %s, not a measurement" % os.environ["DYNAALIGN_BENCH_BACKEND"],
        "boundary": "T_k: kernels only, packed residues and the dense f64 result resident in HBM" if world == 1 else
                    "T_g: kernels + one RCCL all-gather + finalize, dense f64 result resident in every rank's HBM",
        "config": {"workload": "similarityMH k=4 n_hash=500 on %d %s 20-mers (hash seed 12345), dense f64 NxN in HBM"
                               % (n, "h3n2-like" if a.workload == "h3n2like" else "uniform"),
                   "n": n, "k": k, "n_hash": n_hash, "pairs": pairs_mh,
                   "sharding": "1 GPU: upper-triangle tiles + mirrored store" if world == 1
                   else "cyclic tile rows (of the unique strings' table when duplicates are collapsed) over %d ranks, one RCCL all-gather of the "
                        "9-bit packed counts, table rebuild + index expansion / mirror+widen on every rank" % world},
        "roofline": main_roof,
        "phases_ms": phases,
    }
    if world == 1:
        line["route"] = {"n": route["n"], "unique": route["unique"], "dedup": route["dedup"], "pipelined": route.get("pipelined", False),
                         "expansion": route.get("expansion", ""), "chunks": route.get("chunks", 0), "sparse": route.get("sparse", False), "plane_bits": route["plane_bits"],
                         "note": "dedup: byte-identical sequences collapsed (exact) -- K1 / K1b / K2 on the unique strings, then the index expansion to "
                                 "the dense N x N (rows: k_expand_stream, one pass; tiles: column gather + k_expand_rows); direct: the three kernels on all N rows"}
        if route.get("pipelined"):
            os.environ["DYNAALIGN_MH_NO_PIPE"] = "1"             # the same route with the table finished before the expansion starts: every kernel alone
            try:
                odt, oevs = timed_steps(step, max(2, min(a.steps, 3)), 1)
            finally:
                del os.environ["DYNAALIGN_MH_NO_PIPE"]
            osteps = max(2, min(a.steps, 3))
            oph = phase_means(oevs)
            oroof = stream_roofline(oph["expand_ms"], route["unique"]) if rows_form else expand_roofline(oph["expand_ms"])
            oroof["k2_on_unique"] = k2_roofline(oph["k2_ms"], state["bits"], route["unique"], 2)
            oroof["step_frac"] = step_bytes / (odt / osteps) / 1e9 / HBM_PEAK_GBS
            main_roof["frac_one_stream"] = oroof["frac"]      # the same kernel by itself (one launch, nothing co-running), same process
            main_roof["avg_launch_ms_one_stream"] = oph["expand_ms"]
            line["one_stream"] = {"ms_per_step": odt / osteps * 1e3, "value": pairs_mh / (odt / osteps), "unit": "pairs/s", "steps": osteps,
                                  "phases_ms": oph, "roofline": oroof,
                                  "note": "DYNAALIGN_MH_NO_PIPE=1: K2 on the unique strings, then the expansion, one after the other on one stream"}
        if route["dedup"]:
            os.environ["DYNAALIGN_MH_NO_DEDUP"] = "1"            # the same input with the routes off: K2 on all N rows
            os.environ["DYNAALIGN_MH_NO_SPARSE"] = "1"
            try:
                ddt, devs = timed_steps(step, max(2, min(a.steps, 3)), 1)
            finally:
                del os.environ["DYNAALIGN_MH_NO_DEDUP"]
                del os.environ["DYNAALIGN_MH_NO_SPARSE"]
            dsteps = max(2, min(a.steps, 3))
            dph = phase_means(devs)
            droof = k2_roofline(dph["k2_ms"], state["bits"])
            droof["step_frac"] = step_bytes / (ddt / dsteps) / 1e9 / HBM_PEAK_GBS
            dlast = devs[-1]
            line["direct"] = {"ms_per_step": ddt / dsteps * 1e3, "value": pairs_mh / (ddt / dsteps), "unit": "pairs/s", "steps": dsteps,
                              "phases_ms": dph, "roofline": droof,
                              "split": {"taken": dlast.get("split", False), "rare_incidences": dlast.get("rare_pairs", 0), "plane_bits": dlast["plane_bits"],
                                        "plane_bits_without": dlast.get("plane_bits_without", 0),
                                        "note": "heavy / rare split of the column dictionaries: the compare on 8 planes of dense codes for each column's 254 most "
                                                "frequent values + the rare values' incidences added from lists (k2_ms = compare + fix-up; the list kernels run beside it)"},
                              "note": "DYNAALIGN_MH_NO_DEDUP=1: every row goes through K1 / K1b / K2"}
            state["bits"] = route["plane_bits"]
    def same_as_single_gpu(compute_ref):
        """N > 1 only, outside every timed region: this rank recomputes the whole matrix by itself (single-GPU call) and compares
        it with what the sharded step left in `out`, bit for bit; the ranks' verdicts are AND-ed.  Makes the first run of the RCCL
        path on real hardware self-checking."""
        try:
            torch.cuda.empty_cache()
            ref = torch.empty((n, n), dtype=torch.float64, device="cuda")
            compute_ref(ref)
            torch.cuda.synchronize()
            ok = 1 if all(torch.equal(out[r0:r0 + 5000].view(torch.int64), ref[r0:r0 + 5000].view(torch.int64))
                          for r0 in range(0, n, 5000)) else 0
            del ref
        except Exception as e:                                  # (e.g. no room for a second matrix): never fail the measurement over the check
            print("[bench] rank %d: self-check skipped: %s" % (rank, e), file=sys.stderr)
            ok = 2
        t = torch.tensor([ok], dtype=torch.int32, device="cuda")
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return None if int(hi.item()) == 2 else bool(lo.item())

    if world > 1:
        line["verified_against_single_gpu"] = same_as_single_gpu(lambda ref: device.similarity_mh(ds, k, n_hash, d_seeds, out=ref))
        line["rccl"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                        "all_gather_bytes_per_rank": int(state["block_bytes"]), "all_gather_ms": phases["all_gather"],
                        "finalize_ms": phases["shards_to_table"] + phases["expand"] if state["dedup"] else phases["finalize"]}
        # the two T_g boundaries SURVEY 8(d) / 8(e) distinguish: (a) the COMPACT matrix complete on every rank (all-gather done [+ the
        # unique strings' table rebuilt]: what an edge-list / compact consumer needs), (b) + finalize to the dense float64 matrix
        compact_keys = [k_ for k_ in phase_names if k_ not in ("expand", "finalize")]
        t_compact = sum(phases[k_] for k_ in compact_keys)
        t_full = sum(phases[k_] for k_ in phase_names)
        line["boundaries"] = {
            "t_g_compact_ms": t_compact, "t_g_compact_pairs_per_s": pairs_mh / (t_compact * 1e-3),
            "t_g_f64_ms": t_full, "t_g_f64_pairs_per_s": pairs_mh / (t_full * 1e-3),
            "note": "per-step HIP-event sums on this rank (rank 0); `value` is T_g with the float64 finalize, from the barrier-bracketed "
                    "wall clock (max over ranks); compact = " + " + ".join(compact_keys)}
        line["route"] = {"n": n, "unique": state["unique"], "dedup": state["dedup"],
                         "note": "dedup: every rank builds the same duplicate plan, the ranks shard the count table of the unique strings "
                                 "(one all-gather of (U/n)^2 of the bytes) and expand it locally; direct: the n x n pair space is sharded"}
        # beside the sharded step: every rank computing the whole matrix by itself (the single-GPU call, no collective).  The dense result is
        # replicated on every rank either way, so when the compare is cheap next to the N x N stores (duplicates collapsed: 5 of 17 ms) sharding
        # it cannot pay for its all-gather -- this leg says by how much.  Not `value`: the north-star's N > 1 configuration is the sharded one.
        rsteps = max(2, min(a.steps, 3))
        rdt, revs = timed_steps(lambda: (device.similarity_mh(ds, k, n_hash, d_seeds, out=out), device.mh_last_route())[1], rsteps, 1)
        line["replicas_no_collective"] = {"ms_per_step": rdt / rsteps * 1e3, "value": pairs_mh / (rdt / rsteps), "unit": "pairs/s", "steps": rsteps,
                                          "route": {k_: revs[-1][k_] for k_ in ("dedup", "pipelined", "expansion", "unique", "plane_bits")},
                                          "note": "each rank runs da_dev_similarity_mh on the whole set (same result on every rank, no exchange); "
                                                  "barrier-bracketed, max over ranks"}

    # ---- similarityNW on the same set (second half of the metric), >= 3 timed launches
    if not a.no_nw:
        bad = device.nw_encode(ds)
        assert int(bad.item()) == 0
        if world == 1:
            run_nw = lambda: device.nw(ds, "BLOSUM62", 10, 4, 0, n, True, _capi.DA_OUT_F64, out=out)
        else:
            nprobe = device.UniquePlan(ds.codes, ds.offsets, n, ds.total)
            nw_dedup = sharding.dedup_worth(n, nprobe.unique, True, 0, ds.max_len)
            nw_unique = nprobe.unique
            del nprobe
            if nw_dedup:      # ordered table of the unique strings as cyclic row blocks, one all-gather, local expansion
                run_nw = lambda: sharding.nw_sharded_step_dedup(device.UniquePlan(ds.codes, ds.offsets, n, ds.total), ds.max_len, rank, world, out)
            else:
                nplan = sharding.Plan(n, rank, world, sharding.NW_TILE)
                nwork = sharding.Workspace(nplan, "cuda")
                run_nw = lambda: sharding.nw_sharded_step(nplan, nwork, ds, out)
        nw_launches = 3
        t_nw, _ = timed_steps(run_nw, nw_launches, 1)
        t_nw /= nw_launches
        cells = pairs_nw * L * L
        nw_obj = {"workload": "similarityNW BLOSUM62 go=10 ge=4, same %d 20-mers, dense f64 NxN in HBM" % n,
                  "value": pairs_nw / t_nw, "unit": "pairs/s", "ms": t_nw * 1e3, "launches_timed": nw_launches,
                  "gcups": cells / t_nw / 1e9}
        if world > 1:
            nw_obj["route"] = {"n": n, "unique": nw_unique, "dedup": nw_dedup}
            nw_obj["verified_against_single_gpu"] = same_as_single_gpu(
                lambda ref: device.nw(ds, "BLOSUM62", 10, 4, 0, n, True, _capi.DA_OUT_F64, out=ref))
        if world == 1:
            # the call collapses byte-identical sequences first (exact): the DP runs on the table of unique strings as an
            # ordered square and the N x N result is an index expansion (da_nw_last_route: unique count + phase times)
            route = device.nw_last_route()
            nw_obj["route"] = dict(route, note="dedup: DP on the unique strings (ordered square) + expansion; direct: one lane per pair of the input")
            # <NMAX, combined key, ordered mode, generated rows (experiment library only), prefix sharing (ordered mode, <= 20 residues)>
            pfx = route["dedup"] and ds.max_len <= 20 and not os.environ.get("DYNAALIGN_NW_NO_PREFIX_SHARE")
            nw_kernel = "k_nw_short<20, true, %s, false, %s>" % ("true" if route["dedup"] else "false", "true" if pfx else "false")
            pm = pmc_kernel(nw_kernel, n)
            bytes_nw = n * L + n * n * 8
            t_dp = route["dp_ms"] * 1e-3 if route["dp_ms"] > 0 else t_nw
            roof = {"kernel": nw_kernel, "bound": "valu", "avg_launch_ms": t_dp * 1e3,
                    "hbm": {"achieved": bytes_nw / t_nw / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_nw / t_nw / 1e9 / HBM_PEAK_GBS,
                            "algorithmic_bytes_per_call": bytes_nw, "note": "whole call (plan + DP + expansion)"},
                    "peak": VALU_PEAK, "unit": "lane-ops/s",
                    "peak_note": "256 CU x 4 SIMD x 32 lanes x 2.4 GHz; the kernel is integer-VALU-bound by construction (no MFMA: a DP recurrence is not a contraction)"}
            if pm and "SQ_INSTS_VALU" in pm:
                lane_ops = pm["SQ_INSTS_VALU"] * 64.0                       # wave instructions x 64 lanes, per launch of the DP kernel
                roof.update({"lane_ops_per_launch": lane_ops, "achieved": lane_ops / t_dp, "frac": lane_ops / t_dp / VALU_PEAK,
                             "counter_source": pm["source"]})
                if "GRBM_GUI_ACTIVE" in pm:                                  # sum over the 8 XCDs (MI355X_MICROARCH.md, DVFS)
                    clk = pm["GRBM_GUI_ACTIVE"] / 8.0 / t_dp
                    roof.update({"effective_clock_hz": clk, "peak_at_effective_clock": VALU_PEAK * clk / 2.4e9,
                                 "frac_at_effective_clock": lane_ops / t_dp / (VALU_PEAK * clk / 2.4e9)})
            nw_obj["roofline"] = roof
            os.environ["DYNAALIGN_NW_NO_DEDUP"] = "1"                        # the direct kernel on the same input, for reference
            try:
                t_dir, _ = timed_steps(run_nw, 1, 0)
            finally:
                del os.environ["DYNAALIGN_NW_NO_DEDUP"]
            nw_obj["direct"] = {"ms": t_dir * 1e3, "value": pairs_nw / t_dir, "gcups": cells / t_dir / 1e9,
                                "note": "DYNAALIGN_NW_NO_DEDUP=1: every pair of the input goes through the DP (what uniform peptides get)"}
        line["nw"] = nw_obj

    # ---- similarityMH + clusterbreak's quantile threshold as an edge list:
    # shard compare -> histogram -> ONE all-reduce of n_hash+1 words -> exact type-7 quantile -> local edges.
    # No N x N exchange, so this is the variant of the path whose whole-job time scales with the rank count.
    if not a.no_edges:
        if world == 1:
            # one GPU: nothing to shard -- symmetric uint16 compare, histogram of the upper triangle, exact quantile, edges
            cnt16 = out.view(-1).view(torch.int16)[:n * n].view(n, n)          # the f64 result buffer is free here: reuse its first 20 GB
            values = np.arange(n_hash + 1, dtype=np.float64) / n_hash

            eprobe = device.UniquePlan(ds.residues, ds.offsets, n, ds.total)
            edges_dedup = sharding.dedup_worth(n, eprobe.unique, False, n_hash) and eprobe.unique <= 65536
            del eprobe

            def run_edges_direct():
                pl = signatures_and_planes()
                device.mh_compare(pl, n, n_hash, 0, n, True, _capi.DA_OUT_COMPACT, out=cnt16)
                h = device.upper_histogram(cnt16, n, n_hash + 1).cpu().numpy().astype(np.uint64)
                thr = da.quantile_type7(h, values, 0.8)
                keep = (~(values < thr)) & (np.arange(n_hash + 1) != 0)
                m = int(h[keep].sum()) + n
                ei, ej, evv, c = device.extract_edges(cnt16, n, keep, m)
                return thr, ei, ej, evv, c

            def run_edges_dedup():
                # duplicates collapsed: K1 / K1b / K2 on the unique strings, then histogram + extraction read the n x n count matrix
                # through the plan's row map (da_dev_unique_rows) -- no n x n matrix at all
                up = device.UniquePlan(ds.residues, ds.offsets, n, ds.total)
                _, upl = device.minhash_signatures(sharding.UniqueSequences(up, ds.total, ds.max_len), k, n_hash, d_seeds)
                table = device.unique_table(upl, up.unique, n_hash)
                rows = device.unique_rows(table, up)
                h = device.upper_histogram_rows(rows, up, n_hash + 1).cpu().numpy().astype(np.uint64)
                thr = da.quantile_type7(h, values, 0.8)
                keep = (~(values < thr)) & (np.arange(n_hash + 1) != 0)
                m = int(h[keep].sum()) + n
                ei, ej, evv, c = device.extract_edges_rows(rows, up, keep, m)
                return thr, ei, ej, evv, c
            run_edges = run_edges_dedup if edges_dedup else run_edges_direct
        else:
            eplan = sharding.Plan(n, rank, world, sharding.MH_TILE)

            ework = work if work is not None else sharding.PackedWorkspace(eplan, n_hash, "cuda")

            def run_edges():
                return sharding.mh_edges_sharded(eplan, ework, signatures_and_planes(), n_hash, 0.8)
        run_edges()
        sync()
        t0 = time.perf_counter()
        thr, ei, ej, evv, cnt = run_edges()
        sync()
        t_e = max_over_ranks(time.perf_counter() - t0)
        tot = cnt.clone()
        if world > 1:
            dist.all_reduce(tot)
        line["edges"] = {"workload": "similarityMH k=4 n_hash=500 + quantile(S[upper.tri(S)], 0.8) threshold -> edge list "
                                     "(R/clusterbreak.R:219-221), same %d peptides; %s" % (n, "one GPU: symmetric uint16 compare (on the unique strings when "
                                     "duplicates are collapsed, read back through the plan's row map) -> histogram -> "
                                     "quantile -> edges" if world == 1 else "edges stay distributed over the ranks"),
                         "dedup": bool(world == 1 and edges_dedup),
                         "value": pairs_mh / t_e, "unit": "pairs/s", "ms": t_e * 1e3, "threshold": thr,
                         "edges_total": int(tot.item()), "edge_list_bytes": int(tot.item()) * 10}
        del ei, ej, evv

    # ---- the other SURVEY 8(d) workload: S100k uniform.  Its column dictionaries are larger (D ~ 15 000 -> 16 code planes)
    if world == 1 and not a.no_uniform and a.workload == "h3n2like":
        ures, uoff = synth.uniform_peptides(n, L)
        uds = device.DeviceSequences(ures, uoff, "cuda")
        usteps = max(2, min(a.steps, 3))
        udt, uevs = timed_steps(lambda: step(uds), usteps, 1)
        uph = phase_means(uevs)
        ulast = uevs[-1]
        uroof = sparse_roofline(uph["expand_ms"], ulast["sparse_pairs"]) if ulast.get("sparse") else k2_roofline(uph["k2_ms"], state["bits"])
        uroof["step_frac"] = step_bytes / (udt / usteps) / 1e9 / HBM_PEAK_GBS
        line["uniform"] = {"workload": "similarityMH k=4 n_hash=500 on %d uniform 20-mers (SURVEY 8(d) S100k), dense f64 NxN in HBM" % n,
                           "value": pairs_mh / (udt / usteps), "unit": "pairs/s", "ms_per_step": udt / usteps * 1e3, "steps": usteps,
                           "plane_bits": state["bits"], "phases_ms": uph,
                           "route": {"unique": ulast["unique"], "dedup": ulast["dedup"], "sparse": ulast.get("sparse", False),
                                     "matching_incidences": ulast.get("sparse_pairs", 0)},
                           "roofline": uroof}
        if ulast.get("sparse"):                                  # the dense kernels on the same input, for reference (what round 2 reported here)
            os.environ["DYNAALIGN_MH_NO_SPARSE"] = "1"
            try:
                ddt2, devs2 = timed_steps(lambda: step(uds), 2, 1)
            finally:
                del os.environ["DYNAALIGN_MH_NO_SPARSE"]
            dph2 = phase_means(devs2)
            line["uniform"]["dense_kernels"] = {"ms_per_step": ddt2 / 2 * 1e3, "phases_ms": dph2, "plane_bits": state["bits"],
                                                "roofline": k2_roofline(dph2["k2_ms"], state["bits"]),
                                                "note": "DYNAALIGN_MH_NO_SPARSE=1: K1 / K1b / K2 (bit-sliced compare of every pair)"}
        del uds

    # ---- BASELINE configs[4]: clusterbreak(size_max=800, thresh_p=.8) end to end, GPU similarityMH backend on the
    # device edge path (signatures resident; per recursion level codes + compare + histogram + quantile + edges), host Louvain
    if world == 1 and not a.no_clusterbreak:
        from dynaalign_amd.session import MinHashSession
        out = cnt16 = None                  # the 80 GB result buffer is not needed any more (closures above are done)
        torch.cuda.empty_cache()
        seqs = synth.to_strings(res, off)

        def run_clusterbreak():
            t0 = time.perf_counter()
            sess = MinHashSession(seqs, k, n_hash, seed=12345)
            torch.cuda.synchronize()
            t_sess = time.perf_counter() - t0
            r = da.clusterbreak(seqs, thresh_p=0.8, size_max=800, size_min=3, session=sess, cluster_seed=1)
            return time.perf_counter() - t0, t_sess, r
        # two complete runs: the first right after empty_cache() pays the device allocations of the first level (a fresh 20 GB count
        # matrix: hipMalloc takes 0.3 ... 2.3 s depending on the box); the second finds them in the process's allocator cache, like the
        # timed similarity steps above do.  `wall_s` is the second; the first is reported beside it.
        t_cold, _, r_cold = run_clusterbreak()
        labels_cold = r_cold["clustered_seq"]
        del r_cold
        t_cb, t_sess, r = run_clusterbreak()
        assert np.array_equal(labels_cold, r["clustered_seq"])
        del labels_cold
        sizes = np.unique(r["clustered_seq"][:, 1], return_counts=True)[1] if len(r["clustered_seq"]) else np.zeros(1, int)
        line["clusterbreak"] = {
            "workload": "clusterbreak(size_max=800, thresh_p=.8, size_min=3) on the same %d peptides, sim = similarityMH(k=4, n_hash=500) "
                        "through MinHashSession.edges (device edge path), cluster_fn = da_louvain(resolution 1.05)" % n,
            "wall_s": t_cb, "wall_s_first_run_cold_allocations": t_cold, "session_setup_s": t_sess, "calls": r.calls,
            "convergence": r.convergence,
            "similarity_s": float(sum(l["similarity_s"] for l in r.levels)), "louvain_s": float(sum(l["cluster_s"] for l in r.levels)),
            "edges_first_level": r.levels[0]["edges"], "threshold_first_level": r.levels[0]["threshold"],
            "clusters": int(len(sizes)), "largest_cluster": int(sizes.max()), "clustered": int(len(r["clustered_seq"])),
            "filtered": int(len(r["filtered_seq"])),
            "parity": "memberships identical to the oracle dense path at N = 12 000 (tests/test_gpu_clusterbreak.py); at this N the dense path "
                      "needs an 80 GB matrix per level and is not run"}
        del r
        torch.cuda.empty_cache()

    # ---- T_h: the host-pointer boundary (what R sees): da_similarity_mh / da_similarity_nw into a pageable host matrix,
    # upload + kernels + pipelined D2H.  PCIe-bound; never `value`.
    if world == 1 and not a.no_host:
        import psutil
        out = cnt16 = None
        torch.cuda.empty_cache()
        avail = psutil.virtual_memory().available
        nh = n
        while nh > 2000 and nh * nh * 8 * 1.5 + (8 << 30) > avail:
            nh = nh // 2
        hres, hoff = getattr(synth, gen_name)(nh, L)
        hout = np.empty((nh, nh), np.float64)
        lib = _capi.load()
        th = {"n": nh, "workload": "host-pointer entry points (the R glue's calls) on the first-generated %d peptides: upload + kernels + D2H "
                                   "into a pageable host f64 matrix (PCIe-inclusive)" % nh, "matrix_bytes": nh * nh * 8}
        t0 = time.perf_counter()
        _capi.check(lib.da_similarity_mh(hres.ctypes.data, hoff.ctypes.data, nh, k, n_hash, seeds.ctypes.data, hout.ctypes.data))
        t_first = time.perf_counter() - t0                                   # includes the first touch of the destination pages
        t0 = time.perf_counter()
        _capi.check(lib.da_similarity_mh(hres.ctypes.data, hoff.ctypes.data, nh, k, n_hash, seeds.ctypes.data, hout.ctypes.data))
        t_mh = time.perf_counter() - t0
        th["mh"] = {"s": t_mh, "s_first_call_cold_pages": t_first, "value": nh * (nh - 1) // 2 / t_mh, "unit": "pairs/s",
                    "effective_GBs": nh * nh * 8 / t_mh / 1e9}
        if not a.no_nw:
            t0 = time.perf_counter()
            _capi.check(lib.da_similarity_nw(hres.ctypes.data, hoff.ctypes.data, nh, b"BLOSUM62", 10, 4, hout.ctypes.data))
            t_nwh = time.perf_counter() - t0
            th["nw"] = {"s": t_nwh, "value": nh * (nh + 1) // 2 / t_nwh, "unit": "pairs/s", "effective_GBs": nh * nh * 8 / t_nwh / 1e9}
        # small calls: what clusterbreak's default sim_fn (and the R glue) pays per recursion level on a small subset -- the
        # fixed cost of one host-pointer call (upload, kernels, one plain copy of the codes, inline widening)
        small = {}
        for ns in (16, 256, 2000):
            sres, soff = getattr(synth, gen_name)(ns, L)
            sout = np.empty((ns, ns), np.float64)
            lat = {}
            for nm, call in (("mh", lambda: lib.da_similarity_mh(sres.ctypes.data, soff.ctypes.data, ns, k, n_hash, seeds.ctypes.data, sout.ctypes.data)),
                             ("nw", lambda: lib.da_similarity_nw(sres.ctypes.data, soff.ctypes.data, ns, b"BLOSUM62", 10, 4, sout.ctypes.data))):
                _capi.check(call())
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    _capi.check(call())
                    ts.append(time.perf_counter() - t0)
                lat[nm + "_ms"] = min(ts) * 1e3
            small[str(ns)] = lat
        th["small_calls"] = dict(small, note="best of 5 whole host-pointer calls (ms) at n = 16 / 256 / 2000")
        line["t_h"] = th
        del hout

    # ---- CPU oracle on this box's host cores (baseline only; rank 0, N = 1)
    if rank == 0 and world == 1 and not a.no_cpu:
        cb = cpu_baseline(gen_name, a.cpu_seconds)
        line["cpu_baseline"] = cb
        sp = {"boundary": "T_h (whole call, result in host memory) on both sides where t_h was measured, else T_k / whole-call CPU",
              "cpu_variant": "cpu_baseline.value = the reference-structured port (SURVEY 8(d)); cpu_baseline.flat_port is not used here"}
        mh_gpu = line.get("t_h", {}).get("mh", {}).get("value")
        nw_gpu = line.get("t_h", {}).get("nw", {}).get("value")
        sp["mh"] = (mh_gpu or value) / cb["value"]
        sp["mh_kernels_only"] = value / cb["value"]
        if "nw" in line:
            sp["nw"] = (nw_gpu or line["nw"]["value"]) / cb["nw"]["value"]
            sp["nw_kernels_only"] = line["nw"]["value"] / cb["nw"]["value"]
        line["speedup_vs_cpu"] = sp

    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
