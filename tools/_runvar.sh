mkdir -p gpurun_out/r2e
timeout -k 10 200 python tools/k2_time.py 100000 h3n2_like 5 0 --check > gpurun_out/r2e/default.json 2> gpurun_out/r2e/default.err
for v in p3e0 p3e3 p3e1 p2e2; do
  DYNAALIGN_LIB=$PWD/dynaalign_amd/lib/libdynaalign_hip_$v.so timeout -k 10 200 python tools/k2_time.py 100000 h3n2_like 5 0 > gpurun_out/r2e/$v.json 2> gpurun_out/r2e/$v.err
done
cat gpurun_out/r2e/*.json
