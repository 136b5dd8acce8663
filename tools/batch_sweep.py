"""Forward pass at several batch sizes (run under rocprofv3 --kernel-trace by tools/gpu_batch_sweep.sh): is the per-image
time of the dilated (Winograd) layers lower when two layers' activations fit the 256 MiB Infinity Cache?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, synthetic
torch.cuda.set_device(0)
m = Model(NetConfig(grey=False), seed=1)
for n in (8, 16, 24, 32, 48, 64, 96, 128):
    x = torch.from_numpy(synthetic.noise_images(2, n, 512, 512, 3)).cuda()
    for _ in range(40): m.predict_on_device(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(60): m.predict_on_device(x)
    e1.record(); torch.cuda.synchronize()
    print(f"n={n:4d} net {e0.elapsed_time(e1) / 60 * 1e3:8.1f} us  {e0.elapsed_time(e1) / 60 * 1e3 / n:6.2f} us/image", flush=True)
