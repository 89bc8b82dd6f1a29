#!/bin/bash
# PMC passes for the bench workload (run on the GPU box through gpurun).
# Counters go in separate passes (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2), never
# combined with trace domains other than --kernel-trace.
#   tools/prof_pmc.sh <tag> [bench args...]
set -e
TAG=${1:-pmc}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu $*"
# counters are per dispatch and the profiler serialises kernels: profile the one-stream form of the duplicate route (one launch of
# k_expand_rows / k_gather_columns / the compare per step; the pipelined form splits the same work over its chunk launches)
export DYNAALIGN_MH_NO_PIPE=1
export BENCH_ARGS="$ARGS"
run() { # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        import re
        m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
# launch durations of the GRBM pass (same run as its counter): GRBM_GUI_ACTIVE / 8 XCDs / duration = the effective shader clock
for f in glob.glob(os.path.join(out, "grbm", "*", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        import re
        m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:60]
        agg[k]["duration_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
import json
n = 100000
argv = os.environ.get("BENCH_ARGS", "").split()
if "--n" in argv: n = int(argv[argv.index("--n") + 1])
# the hash ties this profile to the build it was taken on (bench.py reports PMC traffic only from a profile of the running sources):
# a profile without it is useless, so fail loudly rather than record null
sys.path.insert(0, os.path.join(os.environ["ROOT"], "tools"))
import source_hash
src_hash = source_hash.source_hash()
assert src_hash, "source hash not computed"
json.dump({"n": n, "source_hash": src_hash, "bench_args": os.environ.get("BENCH_ARGS", ""), "note": "rocprofv3 --pmc, one pass per counter group; "
           "values are per-dispatch averages; FETCH_SIZE/WRITE_SIZE in KiB (FETCH_SIZE under-counts wide reads 2x on gfx950)",
           "kernels": {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}},
          open(os.path.join(out, "summary.json"), "w"), indent=1)
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for k, cs in agg.items():
        fo.write(k + "\n")
        for c, v in sorted(cs.items()):
            fo.write("   %-26s n=%d avg=%.6g\n" % (c, len(v), sum(v) / len(v)))
print(open(os.path.join(out, "summary.txt")).read())
PY
