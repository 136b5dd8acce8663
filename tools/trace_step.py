"""Pipelined fwd+CCL steps for a kernel-trace timeline (run under rocprofv3 --kernel-trace by tools/gpu_timeline.sh)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, ModelRunner, synthetic
torch.cuda.set_device(0)
cfg = NetConfig(grey=False)
m = Model(cfg, seed=1)
runner = ModelRunner(cfg, pixel_threshold=0.5, max_objects_per_image=1024, pipelined=True)
labels = synthetic.rectangle_maps(3, 32, 128, 128)
x = torch.from_numpy(synthetic.textured_images(4, labels, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
for _ in range(int(os.environ.get("STEPS", 400))): runner.predict_on_device(m, x)
torch.cuda.synchronize()
