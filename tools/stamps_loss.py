"""In-kernel stamps of the one-launch loss (a -DLOSS1_STAMPS build: tools/build_variant.sh loss_stamps loss -DLOSS1_STAMPS): wall-clock stamps
(100 MHz) of every block at the stage boundaries, read back from the (otherwise unused) ce buffer of the workspace.
  UBD_LIB_PATH=tools/_ab/loss_stamps.so python tools/stamps_loss.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import _lib, synthetic
torch.cuda.set_device(0)
lib = _lib.load()
n, h, w, k = 64, 128, 128, 1
lab = torch.from_numpy(synthetic.rectangle_maps(30, n, h, w)).cuda().to(torch.int32)
g = torch.Generator(device="cuda"); g.manual_seed(1)
logits = (torch.randn((n, h, w, k), device="cuda", generator=g) * 2.0 - 3.0 + 5.0 * (lab > 0).float()[..., None]).contiguous()
cfg = _lib.UbdConfig(1, 0, 1, _lib.UBD_F32)
hd = ctypes.c_void_p(); _lib.check(lib.ubd_create(ctypes.byref(cfg), ctypes.byref(hd)), "create")
nb = int(lib.ubd_loss_workspace_bytes(hd, n, h, w))
ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
loss = torch.zeros(16, device="cuda"); grad = torch.empty_like(logits)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def call(): _lib.check(lib.ubd_loss(hd, logits.data_ptr(), lab.data_ptr(), n, h, w, loss.data_ptr(), grad.data_ptr(), ws.data_ptr(), ws.numel(), st), "loss")
for _ in range(50): call()
torch.cuda.synchronize()
ce_off = nb - ((n * h * w * 4 + 255) // 256 * 256)
names = ["start", "loads", "lds hist", "flush+records", "barrier 1 passed", "stage 1", "barrier 2 passed", "stage 2", "barrier 3 passed", "gradient", "end"]
acc = []
for rep in range(20):
    call(); torch.cuda.synchronize()
    s = ws[ce_off:ce_off + 256 * 16 * 8].view(torch.int64).cpu().numpy().reshape(256, 16).astype(np.float64)
    t0 = s[:, 0].min()
    acc.append((s - t0) / 100.0)       # us
a = np.median(np.stack(acc), axis=0)
print("stage                    min     median  max   (us since the first block's start; 256 blocks)")
for i, nm in enumerate(names):
    print(f"{nm:22s} {a[:, i].min():7.2f} {np.median(a[:, i]):7.2f} {a[:, i].max():7.2f}")
for i, nm in ((12, "last arriver 1 starts"), (13, "last arriver 2 starts"), (14, "last arriver 3 starts")):
    v = a[:, i][a[:, i] > 0]
    print(nm, v)
