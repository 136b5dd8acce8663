"""Per-launch durations of selected kernels in one bf16 train step (rocprofv3 kernel trace): tools/gpu_trace_py.sh wraps it."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic
torch.cuda.set_device(0)
n = 64
m = Model(NetConfig(grey=False), dtype="bfloat16", seed=1)
tr = Trainer(m, Adam())
lab = synthetic.rectangle_maps(30, n, 128, 128)
x = torch.from_numpy(synthetic.textured_images(31, lab, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
y = torch.from_numpy(lab).cuda()
for _ in range(3): tr.train_step_on_device(x, y)
torch.cuda.synchronize()
