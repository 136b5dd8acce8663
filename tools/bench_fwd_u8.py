"""fp32 / fp16 forward fed preprocessed fp32 images vs uint8 images (fused (x - 127.5) / 127.5), steady state; 32 x 512 x 512 and 8 x 1024 x 1024."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, synthetic, PreprocessingType
torch.cuda.set_device(0)
def timed(fn, per=500, blocks=3):
    for _ in range(300): fn()
    out = []
    for _ in range(blocks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(per): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(round(e0.elapsed_time(e1) / per, 4))
    return out
for dt, shape in (("float32", (32, 512, 512)), ("float16", (8, 1024, 1024))):
    n, h, w = shape
    x8 = np.random.default_rng(3).integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    xf = torch.from_numpy(x8.astype(np.float32) / 127.5 - 1.0).cuda()
    xu = torch.from_numpy(x8).cuda()
    mf = Model(NetConfig(grey=False), dtype=dt, seed=1)
    mu = Model(NetConfig(grey=False, preprocessing=PreprocessingType.MOBILENET_LIKE), dtype=dt, seed=1)
    print(dt, shape, "fp32-fed", timed(lambda: mf.predict_on_device(xf)), "uint8-fed", timed(lambda: mu.predict_on_device(xu)), flush=True)
