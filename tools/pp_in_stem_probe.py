"""Forward pass of 32 images with the postprocess of k earlier maps riding in the stem kernel: how the pass stretches with k."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, synthetic
torch.cuda.set_device(0)
cfg = NetConfig(grey=False)
m = Model(cfg, seed=1)
labs = synthetic.rectangle_maps(3, 32, 128, 128)
x = torch.from_numpy(synthetic.textured_images(4, labs, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
out = torch.empty((32, 128, 128, 1), device="cuda")
prev = m.predict_on_device(x).clone()
rect = torch.from_numpy(synthetic.logits_from_maps(labs, 0, seed=5)).cuda()
def timed(fn, reps=300):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("bare forward", round(timed(lambda: m.predict_on_device(x, out=out)), 4))
for name, src in (("net maps", prev), ("rectangle maps", rect)):
    for k in (1, 8, 16, 32):
        lg = src[:k].contiguous()
        outs = m.alloc_postprocess_outputs(k, 128, 128, 1024)
        job = {"logits": lg, "logit_threshold": 0.0, "scale": 4, "min_area": 5, "cap": 1024, "outputs": outs}
        print(name, "pp of", k, "maps in the stem:", round(timed(lambda: m.predict_on_device(x, out=out, postprocess=job)), 4), flush=True)
