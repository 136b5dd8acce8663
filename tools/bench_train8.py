"""bf16 train step with 8 classes at batch 64 (cfg3's second run), steady state: blocks of 200 steps; LIB=<name under tools/_ab> selects another build."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ubdvss_amd import _lib
if os.environ.get("LIB"): _lib.LIB_PATH = os.path.join(ROOT, "tools", "_ab", os.environ["LIB"])
from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic
torch.cuda.set_device(0)
m = Model(NetConfig(class_names=[f"c{i}" for i in range(8)], grey=False), dtype="bfloat16", seed=1)
tr = Trainer(m, Adam())
lab = synthetic.rectangle_maps(30, 64, 128, 128, n_classes=8)
x = torch.from_numpy(synthetic.textured_images(31, lab, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
y = torch.from_numpy(lab).cuda()
for _ in range(50): tr.train_step_on_device(x, y)
out = []
for blk in range(int(os.environ.get("BLOCKS", "6"))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): tr.train_step_on_device(x, y)
    e1.record(); torch.cuda.synchronize()
    out.append(round(e0.elapsed_time(e1) / 200, 4))
print(os.environ.get("UBD_HEADBWD", "one pass"), out)
