"""In-kernel phase timing of sep123_16_kernel (diagnostic build, tools/build_diag.sh): s_memtime stamps of every wave at the phase
boundaries of its first 16 tiles, cfg5 shape (8 x 1024 x 1024 fp16) or the bf16 train step (TRAIN=1).  Prints median ticks (100 MHz) per segment."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ubdvss_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_ab", os.environ.get("DIAG_LIB", "libubd_hip_diag.so"))
from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic
torch.cuda.set_device(0)
lib = _lib.load()
train = os.environ.get("TRAIN") == "1"
if train:
    m = Model(NetConfig(grey=False), dtype="bfloat16", seed=1)
    tr = Trainer(m, Adam())
    lab = synthetic.rectangle_maps(30, 64, 128, 128)
    x = torch.from_numpy(synthetic.textured_images(31, lab, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
    y = torch.from_numpy(lab).cuda()
    run = lambda: tr.train_step_on_device(x, y)
else:
    m = Model(NetConfig(grey=False), dtype="float16", seed=1)
    x = torch.from_numpy(synthetic.noise_images(2, 8, 1024, 1024, 3)).cuda()
    run = lambda: m.predict_on_device(x)
for _ in range(200): run()
st = torch.zeros((768, 4, 16, 8), dtype=torch.int64, device="cuda")
lib.ubd_debug_set_stamps_s123.argtypes = [ctypes.c_void_p]; lib.ubd_debug_set_stamps_s123.restype = None
lib.ubd_debug_set_stamps_s123(st.data_ptr())
for _ in range(3): run()
torch.cuda.synchronize()
lib.ubd_debug_set_stamps_s123(None)
s = st.cpu().numpy().astype(np.int64)
names = ["wait img+barrier", "L1", "frag req+barrier", "frag wait", "dma issue(+a1 copy) L2", "L3 frag req+barrier", "a2 copy + L3"]
names = ["wait img + barrier", "L1 units", "L2-frag request + barrier", "L2-frag wait", "dma issue, a1 copy, L2 units", "L3-frag request + barrier", "a2 copy, L3 unit"]
seg = np.diff(s, axis=-1)                        # (blk, wave, it, 7)
its = slice(2, 9)
print("ticks of the 100-MHz s_memtime clock (x ~21 = shader cycles), median over blocks x tiles 2..8, per wave:")
for w in range(4):
    print(f" wave {w}: " + "  ".join(f"{names[k]} {np.median(seg[:, w, its, k]):6.0f}" for k in range(7)))
per_tile = s[:, 0, 3:9, 0] - s[:, 0, 2:8, 0]
print("tile period (wave 0): median", np.median(per_tile), "p10", np.percentile(per_tile, 10), "p90", np.percentile(per_tile, 90))
