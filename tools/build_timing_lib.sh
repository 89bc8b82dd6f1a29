#!/bin/bash
# Debug build of the library with per-workgroup phase stamps in k_mh_compare (tools/k2_timeline.py).
set -e
cd "$(dirname "$0")/../dynaalign_amd/csrc"
mkdir -p build_timing
FLAGS="-O3 -std=c++17 -fPIC -fwrapv --offload-arch=gfx950 -DDA_K2_TIMING -Wno-inline-asm"
for f in api.cpp louvain.cpp minhash_kernels.hip dict_kernels.hip nw_kernels.hip graph_kernels.hip; do
  /opt/rocm/bin/hipcc $FLAGS -x hip -c $f -o build_timing/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libdynaalign_hip_timing.so build_timing/*.o -Wl,-rpath,/opt/rocm/lib -lpthread -ldl
echo built ../lib/libdynaalign_hip_timing.so
