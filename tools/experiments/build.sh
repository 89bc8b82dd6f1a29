#!/bin/bash
# Builds the EXPERIMENT twin of the library (not the product): libdynaalign_hip.so + round 3's two float64-store compare kernels
# (k_mh_compare_r12, k_mh_compare_q12; tools/experiments/k2_store_*.inc) behind DYNAALIGN_K2_ROLES=1 / DYNAALIGN_K2_INLOOP=1, and round 4's
# generated NW rows (tools/gen_nw_asm.py -> nw_rows_p12 / p20.inc) behind DYNAALIGN_NW_ASM=1.
#   tools/experiments/build.sh          -> tools/experiments/lib/libdynaalign_hip.so
#   DYNAALIGN_LIB=tools/experiments/lib/libdynaalign_hip.so python tools/k2_time.py ...      (dynaalign_amd/_capi.py honours DYNAALIGN_LIB)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
K2ASM_INLOOP=1 python3 "$ROOT/tools/gen_k2_asm.py" "$ROOT/tools/experiments/k2_loop_p12q.inc" > /dev/null
for n in 12 20; do NWASM_NMAX=$n python3 "$ROOT/tools/gen_nw_asm.py" "$ROOT/tools/experiments/nw_rows_p$n.inc" > /dev/null; done
make -s -C "$ROOT/dynaalign_amd/csrc" BUILD=build_experiments OUT="$ROOT/tools/experiments/lib/libdynaalign_hip.so" \
  CXXFLAGS="-O3 -std=c++17 -fPIC -fwrapv --offload-arch=gfx950 -Wall -Wno-unused-function -DDA_K2_EXPERIMENTS -I. -I$ROOT/tools/experiments"
echo "built $ROOT/tools/experiments/lib/libdynaalign_hip.so"
