cd $GRAFT_REPO_ROOT
python - <<'PY' 2>/dev/null
import os, sys, json, time
sys.path.insert(0, os.getcwd())
import torch
from dynaalign_amd import device, synth
n = 100000
res, off = synth.h3n2_like(n, 20)
ds = device.DeviceSequences(res, off)
device.nw_encode(ds)
out = torch.empty((n, n), dtype=torch.float64, device="cuda")
for k in range(3):
    device.nw(ds, out=out); torch.cuda.synchronize()
    print(json.dumps(device.nw_last_route()))
ures, uoff = synth.uniform_peptides(n, 20)
uds = device.DeviceSequences(ures, uoff); device.nw_encode(uds)
for k in range(2):
    t = time.perf_counter(); device.nw(uds, out=out); torch.cuda.synchronize(); print("uniform direct ms", (time.perf_counter() - t) * 1e3, json.dumps(device.nw_last_route()))
PY
