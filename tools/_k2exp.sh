set -e
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/dynaalign_amd/lib
for wg in 4 3 2; do
  for v in dyn dynns; do
    echo "== persistent $v wg_per_cu=$wg"
    DYNAALIGN_LIB=$L/libdynaalign_hip_$v.so DYNAALIGN_K2_PERSIST=1 DYNAALIGN_K2_WG_PER_CU=$wg timeout -k 10 120 python tools/k2_time.py 100000 h3n2_like 5
  done
done
echo "== one tile per workgroup (default lib)"
timeout -k 10 120 python tools/k2_time.py 100000 h3n2_like 5
