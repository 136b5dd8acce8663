"""In-kernel phase timing of the fp32 sep_bwd_kernel (diagnostic build, tools/build_diag.sh): s_memtime of every wave at the phase
boundaries of its first 8 tiles, fp32 train step at batch 64.  Usage: stamps_sepb32.py   (24 1 = L2, 3 2 = L1, 24 2 = L3)"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ubdvss_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_ab", "libubd_hip_diag.so")
from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic
torch.cuda.set_device(0)
lib = _lib.load()
m = Model(NetConfig(grey=False), dtype="float32", seed=1)
tr = Trainer(m, Adam())
lab = synthetic.rectangle_maps(30, 64, 128, 128)
x = torch.from_numpy(synthetic.textured_images(31, lab, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
y = torch.from_numpy(lab).cuda()
for _ in range(40): tr.train_step_on_device(x, y)
lib.ubd_debug_set_stamps_sepb.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]; lib.ubd_debug_set_stamps_sepb.restype = None
names = ["top barrier", "issue DMA (G, X, upper patch) / X regs", "wait + barrier (DMA landed)", "border fix", "-", "-", "G tile", "barrier", "row loop"]
for cin, stride in ((24, 1), (3, 2), (24, 2)):
    st = torch.zeros((1024, 4, 8, 12), dtype=torch.int64, device="cuda")
    lib.ubd_debug_set_stamps_sepb(st.data_ptr(), cin, stride)
    tr.train_step_on_device(x, y); torch.cuda.synchronize()
    lib.ubd_debug_set_stamps_sepb(None, 0, 0)
    s = st.cpu().numpy()
    used = s[:, 0, 2, 0] > 0
    s = s[used]
    for k in (5, 6): s[:, :, :, k] = s[:, :, :, 4]       # stamps 5, 6 (upper patch through registers) are no longer taken
    if not (s[0, 0, 2, 7] > 0): s[:, :, :, 7] = s[:, :, :, 4]   # no in-block G tile
    seg = np.diff(s[:, :, 1:7, :10], axis=-1)          # (blk, wave, tile 1..6, 9 segments between the 10 stamps)
    period = s[:, 0, 2:8, 0] - s[:, 0, 1:7, 0]
    print(f"sep_bwd<{cin},{stride}> fp32: blocks {used.sum()}, tile period median {np.median(period):.0f} cycles (s_memtime)")
    for w in range(4):
        print(f"  wave {w}: " + "  ".join(f"{names[k]}: {np.median(seg[:, w, :, k]):.0f}" for k in range(9)))
