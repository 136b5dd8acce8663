import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ubdvss_amd import NetConfig, Model
torch.cuda.set_device(0)
m = Model(NetConfig(grey=True), seed=1)
for side in (512, 1024):
    x = torch.zeros((1, side, side, 1), device="cuda")
    for _ in range(50): m.predict_on_device(x)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); y = m.predict_on_device(x); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(side, "on-device median ms", np.median(ts), flush=True)
import time
for side in (512, 1024):
    x = torch.zeros((1, side, side, 1), device="cuda")
    gf = m.graphed_forward(1, side, side)
    for _ in range(50): gf(x)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.time(); gf(x); torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3)
    print(side, "graph wall-clock median ms", np.median(ts), "equal", bool(torch.equal(gf(x), m.predict_on_device(x))), flush=True)
    ts = []
    for _ in range(20):
        t0 = time.time(); m.predict_on_device(x); torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3)
    print(side, "launches wall-clock median ms", np.median(ts), flush=True)
