"""Postprocess alone on rectangle maps (32 x 128 x 128), for rocprofv3 kernel traces."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, ModelRunner, synthetic
torch.cuda.set_device(0)
cfg = NetConfig(grey=False)
m = Model(cfg, seed=1)
runner = ModelRunner(cfg, pixel_threshold=0.5, max_objects_per_image=1024)
labels = synthetic.rectangle_maps(3, 32, 128, 128)
logits = torch.from_numpy(synthetic.logits_from_maps(labels, 0, seed=5)).cuda()
for _ in range(5): m.postprocess_on_device(logits, runner.logit_threshold, 4, 5, cap=1024)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): out = m.postprocess_on_device(logits, runner.logit_threshold, 4, 5, cap=1024)
e1.record(); torch.cuda.synchronize()
print(f"postprocess alone: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per batch of 32 maps; objects/img {out[3].float().mean().item():.2f}")
