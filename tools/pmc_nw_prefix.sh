#!/bin/bash
# SQ_INSTS_VALU / SQ_WAVE_CYCLES / SQ_BUSY_CYCLES of the NW kernels with and without prefix sharing
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in share noshare; do
  if [ $v = noshare ]; then export DYNAALIGN_NW_NO_PREFIX_SHARE=1; else unset DYNAALIGN_NW_NO_PREFIX_SHARE; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/$v -- python3 $ROOT/tools/nw_time.py > $OUT/$v.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections, re
out=sys.argv[1]
for v in ('share','noshare'):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out,v,'*','*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            m=re.search(r"(k_nw_short<[^>]*>)", r["Kernel_Name"])
            if m: agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,cs in agg.items():
        print(v,k,{c: "%.4g" % (sum(x)/len(x)) for c,x in cs.items()})
PY
