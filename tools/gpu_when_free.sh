#!/bin/bash
# gpurun, waiting for a free slot: retries ONLY while gpurun reports "no box or slot free" (exit code 3: nothing ran, nothing
# was charged).  Any other outcome -- including a failed or timed-out command -- is final.
#   tools/gpu_when_free.sh <timeout_s> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
