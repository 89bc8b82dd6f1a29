cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/dynaalign_amd/lib
echo "== q12 full"; timeout -k 10 120 python tools/k2_time.py 100000 h3n2_like 5 2>/dev/null
echo "== q12 without the global stores"; DYNAALIGN_LIB=$L/libdynaalign_hip_qx1.so timeout -k 10 120 python tools/k2_time.py 100000 h3n2_like 5 2>/dev/null
echo "== q12 without the table reads"; DYNAALIGN_LIB=$L/libdynaalign_hip_qx2.so timeout -k 10 120 python tools/k2_time.py 100000 h3n2_like 5 2>/dev/null
echo "== baseline one tile per WG"; DYNAALIGN_K2_NO_INLOOP=1 timeout -k 10 120 python tools/k2_time.py 100000 h3n2_like 5 2>/dev/null
