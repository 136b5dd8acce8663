"""cfg5 alone (8 x 1024 x 1024 x 3 fp16 forward), 300 passes: for a kernel trace (tools/gpu_prof_py.sh tools/bench_cfg5.py)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, synthetic
torch.cuda.set_device(0)
m = Model(NetConfig(grey=False), dtype="float16", seed=1)
x = torch.from_numpy(synthetic.noise_images(2, 8, 1024, 1024, 3)).cuda()
for _ in range(300): m.predict_on_device(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(300): m.predict_on_device(x)
e1.record(); torch.cuda.synchronize()
print(f"cfg5: {e0.elapsed_time(e1) / 300:.4f} ms/batch")
