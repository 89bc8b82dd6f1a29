#!/bin/bash
# Builds variants of the library that differ in the generated K2 stage loop / compile-time knobs:
#   tools/k2_variants.sh name "K2ASM_X=.. K2ASM_Y=.." "-DK2_PRO_PRIO=3 ..."   ->  dynaalign_amd/lib/libdynaalign_hip_<name>.so
set -e
NAME=$1; GENENV=$2; DEFS=$3
cd "$(dirname "$0")/../dynaalign_amd/csrc"
env $GENENV python3 ../../tools/gen_k2_asm.py k2_loop_$NAME.inc > /dev/null
make -s BUILD=build_$NAME OUT=../lib/libdynaalign_hip_$NAME.so \
  CXXFLAGS="-O3 -std=c++17 -fPIC -fwrapv --offload-arch=gfx950 -Wno-unused-function $DEFS -DK2_LOOP_INC=\\\"k2_loop_$NAME.inc\\\""
echo built libdynaalign_hip_$NAME.so
