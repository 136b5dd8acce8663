"""Forward-only timing of the 16-bit paths: cfg5 (8 x 1024x1024x3 fp16) and the cfg3 forward shape (64 x 512x512x3 bf16)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, synthetic
torch.cuda.set_device(0)
for name, dtype, n, side in (("cfg5 fp16 8x1024^2", "float16", 8, 1024), ("cfg3-fwd bf16 64x512^2", "bfloat16", 64, 512), ("cfg2-fwd f32 32x512^2", "float32", 32, 512)):
    m = Model(NetConfig(grey=False), dtype=dtype, seed=1)
    x = torch.from_numpy(synthetic.noise_images(2, n, side, side, 3)).cuda()
    for _ in range(3): m.predict_on_device(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): m.predict_on_device(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    elts = side * side * (3 + 45 + 1 / 16)
    bpe = 4 if dtype == "float32" else 2
    gb = n * elts * bpe / 1e9
    print(f"{name}: {ms:.3f} ms/batch  {n / ms * 1e3:.0f} img/s  algorithmic {gb / ms * 1e3:.0f} GB/s = {gb / ms * 1e3 / 8000 * 100:.1f}% of 8 TB/s")
