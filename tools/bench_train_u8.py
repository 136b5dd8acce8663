import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic, PreprocessingType
torch.cuda.set_device(0)
lab = synthetic.rectangle_maps(30, 64, 128, 128)
img8 = synthetic.textured_images(31, lab, 4, 3)
y = torch.from_numpy(lab).cuda()
for name, cfg, x in (("fp32 preprocessed", NetConfig(grey=False), torch.from_numpy(img8.astype(np.float32) / 127.5 - 1.0).cuda()),
                     ("uint8 + fused preprocessing", NetConfig(grey=False, preprocessing=PreprocessingType.MOBILENET_LIKE), torch.from_numpy(img8).cuda())):
    tr = Trainer(Model(cfg, dtype="bfloat16", seed=1), Adam())
    for _ in range(300): tr.train_step_on_device(x, y)
    out = []
    for blk in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): tr.train_step_on_device(x, y)
        e1.record(); torch.cuda.synchronize()
        out.append(round(e0.elapsed_time(e1) / 200, 4))
    print(name, out)
