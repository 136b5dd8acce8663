"""Time of the loss alone (ubd_loss) on the train step's shape, 64 x 128 x 128 x 1 logits (1 M pixels), rocprof-free: HIP events around 400
calls after 100.  Default: the one-launch kernel (+ the workspace memset); UBD_LOSS=chain: stats -> two histogram levels -> tie count ->
gradient as five launches.  UBD_LIB_PATH selects another build (A/B)."""
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
ws = torch.empty(int(lib.ubd_loss_workspace_bytes(hd, n, h, w)), dtype=torch.uint8, device="cuda")
loss = torch.zeros(16, device="cuda"); grad = torch.empty_like(logits)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def call(): _lib.check(lib.ubd_loss(hd, logits.data_ptr(), lab.data_ptr(), n, h, w, loss.data_ptr(), grad.data_ptr(), ws.data_ptr(), ws.numel(), st), "loss")
for _ in range(100): call()
out = []
for blk in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(400): call()
    e1.record(); torch.cuda.synchronize()
    out.append(round(e0.elapsed_time(e1) / 400 * 1e3, 2))
print(os.environ.get("UBD_LIB_PATH", "product"), "UBD_LOSS=" + os.environ.get("UBD_LOSS", "one-launch"), "us per call (incl. the workspace memset):", out, "loss", float(loss[0]), "k", float(loss[3]))
