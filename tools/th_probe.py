import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import dynaalign_amd as da
from dynaalign_amd import _capi, synth
n = int(sys.argv[1])
res, off = synth.h3n2_like(n, 20)
seeds = da.hash_family_seeds(12345, 500)
lib = _capi.load()
out = np.empty((n, n), np.float64)
for rep in range(3):
    t = time.perf_counter()
    _capi.check(lib.da_similarity_mh(res.ctypes.data, off.ctypes.data, n, 4, 500, seeds.ctypes.data, out.ctypes.data))
    print("MH call %d: %.3f s" % (rep, time.perf_counter() - t), flush=True)
for rep in range(2):
    t = time.perf_counter()
    _capi.check(lib.da_similarity_nw(res.ctypes.data, off.ctypes.data, n, b"BLOSUM62", 10, 4, out.ctypes.data))
    print("NW call %d: %.3f s" % (rep, time.perf_counter() - t), flush=True)
