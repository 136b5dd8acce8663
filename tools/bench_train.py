"""bf16 train step at batch 64 (the bench's train leg alone): ms per step, three repeats."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic
torch.cuda.set_device(0)
dt = sys.argv[1] if len(sys.argv) > 1 else "bfloat16"
m = Model(NetConfig(grey=False), dtype=dt, seed=1)
tr = Trainer(m, Adam())
lab = synthetic.rectangle_maps(30, 64, 128, 128)
x = torch.from_numpy(synthetic.textured_images(31, lab, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
y = torch.from_numpy(lab).cuda()
for _ in range(60): tr.train_step_on_device(x, y)
torch.cuda.synchronize()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): tr.train_step_on_device(x, y)
    e1.record(); torch.cuda.synchronize()
    print(f"{dt} train step {e0.elapsed_time(e1) / 100:.4f} ms  ({64 / (e0.elapsed_time(e1) / 100) * 1e3:.0f} img/s)", flush=True)
