"""Times the train step (fwd + loss + bwd + Adam) at batch 64, 512x512x3: tools/bench_train.py [batch] [dtype] [n_classes]."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic
torch.cuda.set_device(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ncls = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cfg = NetConfig(grey=False, class_names=[f"c{i}" for i in range(ncls)] if ncls else None)
dtype = sys.argv[2] if len(sys.argv) > 2 else "float32"
m = Model(cfg, dtype=dtype, seed=1)
tr = Trainer(m, Adam())
lab = synthetic.rectangle_maps(30, n, 128, 128, n_classes=ncls)
x = torch.from_numpy(synthetic.textured_images(31, lab, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
y = torch.from_numpy(lab).cuda()
for _ in range(200): tr.train_step_on_device(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): tr.train_step_on_device(x, y)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
print(f"{dtype}: train step {dt*1e3:.3f} ms  ({n/dt:.0f} img/s)")
