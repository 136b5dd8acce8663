"""cfg5 (8 x 1024 x 1024 x 3, fp16 activations) fed three ways: fp32 images as they are (LDS-DMA path), fp32 images with the
reference's preprocessing left to the device, uint8 raw pixels with the preprocessing fused (what the reference feeds: A0)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, synthetic, PreprocessingType
torch.cuda.set_device(0)
def timed(fn, reps=200):
    for _ in range(300): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
m = Model(NetConfig(grey=False, preprocessing=PreprocessingType.MOBILENET_LIKE), dtype="float16", seed=1)
xs = [torch.from_numpy(synthetic.noise_images(2 + k, 8, 1024, 1024, 3)).cuda() for k in range(4)]
us = [torch.from_numpy(synthetic.noise_images(2 + k, 8, 1024, 1024, 3, as_float=False)).cuda() for k in range(4)]
i = [0]
def f32(): i[0] += 1; m.predict_on_device(xs[i[0] & 3])
def u8(): i[0] += 1; m.predict_on_device(us[i[0] & 3])
for rep in range(2):
    print(f"cfg5 fp32 input: {timed(f32):.4f} ms   uint8 input (preprocessing fused): {timed(u8):.4f} ms", flush=True)
