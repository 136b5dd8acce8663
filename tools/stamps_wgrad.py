"""In-kernel phase timing of dil_wgrad16_kernel (diagnostic build): s_memtime of every wave at the phase boundaries of its
first 8 items, bf16 train step at batch 64, per dilation."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ubdvss_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_ab", os.environ.get("DIAG_LIB", "libubd_hip_diag.so"))
from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic
torch.cuda.set_device(0)
lib = _lib.load()
m = Model(NetConfig(grey=False), dtype="bfloat16", seed=1)
tr = Trainer(m, Adam())
lab = synthetic.rectangle_maps(30, 64, 128, 128)
x = torch.from_numpy(synthetic.textured_images(31, lab, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
y = torch.from_numpy(lab).cuda()
for _ in range(100): tr.train_step_on_device(x, y)
lib.ubd_debug_set_stamps_sepb.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]; lib.ubd_debug_set_stamps_sepb.restype = None
names = ["wait DMA", "barrier", "decode + issue next DMA", "weight-gradient MFMAs", "data-gradient phase"]
for d in (1, 2, 4, 8, 16):
    st = torch.zeros((1024, 4, 8, 8), dtype=torch.int64, device="cuda")
    lib.ubd_debug_set_stamps_sepb(st.data_ptr(), -1, d)
    tr.train_step_on_device(x, y); torch.cuda.synchronize()
    lib.ubd_debug_set_stamps_sepb(None, 0, 0)
    s = st.cpu().numpy()
    s = s[s[:, 0, 2, 0] > 0]
    seg = np.diff(s[:, :, 1:7, :6], axis=-1)
    nxt = s[:, :, 2:8, 0] - s[:, :, 1:7, 5]
    period = s[:, 0, 2:8, 0] - s[:, 0, 1:7, 0]
    print(f"dil_wgrad16 d={d}: blocks {len(s)}, item period median {np.median(period):.0f} cycles")
    for w in range(4):
        print(f"  wave {w}: " + "  ".join(f"{names[k]}: {np.median(seg[:, w, :, k]):.0f}" for k in range(5)) + f"  loop: {np.median(nxt[:, w]):.0f}")
