#!/bin/bash
# one script, several builds of the library, ONE box, one process each:   tools/libs_ab.sh out.txt "<python script + args>" lib1.so lib2.so ...
OUT=$1; CMD=$2; shift 2
for lib in "$@"; do
  echo "== $lib" >> "$OUT"
  DYNAALIGN_LIB=$lib timeout -k 10 300 python3 $CMD >> "$OUT" 2>/dev/null || echo failed >> "$OUT"
done
cat "$OUT"
