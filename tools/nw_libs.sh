#!/bin/bash
# times similarityNW (100k h3n2-like, duplicate route + direct sweep) with several builds of the library on ONE box, one process each
#   tools/nw_libs.sh out.txt libA.so libB.so ...
OUT=$1; shift
for lib in "$@"; do
  echo "== $lib" >> "$OUT"
  DYNAALIGN_LIB=$lib timeout -k 10 200 python3 tools/nw_time.py >> "$OUT" 2>/dev/null || echo failed >> "$OUT"
done
cat "$OUT"
