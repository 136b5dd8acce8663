"""Forward-pass time with the fused L2->L3 stem kernel vs the separate kernels (UBD_STEM is read when the handle is created)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, synthetic
torch.cuda.set_device(0)
cfg = NetConfig(grey=False)
x = torch.from_numpy(synthetic.noise_images(2, 32, 512, 512, 3)).cuda()
models = {}
for mode in ("fused123", "fused", "unfused"):
    os.environ["UBD_STEM"] = mode
    models[mode] = Model(cfg, seed=1)
def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for _ in range(300): models["fused123"].predict_on_device(x)
for rep in range(3):
    for mode, m in models.items():
        print(f"{mode}: net {timed(lambda: m.predict_on_device(x), 300):.4f} ms", flush=True)
b = models["unfused"].predict_on_device(x).clone()
for mode in ("fused", "fused123"):
    print(f"max |{mode} - unfused| =", float((models[mode].predict_on_device(x) - b).abs().max()), "max |logit| =", float(b.abs().max()))
