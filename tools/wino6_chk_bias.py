import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ab", sys.argv[1])
from ubdvss_amd import NetConfig, Model
torch.cuda.set_device(0)
lib = _lib.load()
m = Model(NetConfig(grey=False), seed=1)
p = m.params
p.zero_()
off = 1755
p[off + 5184: off + 5208] = torch.arange(1, 25, dtype=torch.float32).cuda()
ws = torch.empty(int(lib.ubd_forward_workspace_bytes(m._h, 1, 4, 4)), dtype=torch.uint8, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.ubd_pack_weights(m._h, p.data_ptr(), ws.data_ptr(), ws.numel(), st), "pack")
for shape in ((1, 32, 32), (2, 72, 100), (32, 128, 128)):
    x = torch.rand((*shape, 24), device="cuda")
    y = torch.empty_like(x)
    _lib.check(lib.ubd_dilated_layer(m._h, p.data_ptr(), 0, x.data_ptr(), y.data_ptr(), *shape, ws.data_ptr(), st), "dil")
    torch.cuda.synchronize()
    yy = y.reshape(-1, 24)
    bad = (yy != torch.arange(1, 25, device="cuda", dtype=torch.float32)).any(dim=1)
    print(shape, "pixels wrong", int(bad.sum()), "of", yy.shape[0])
    if bad.any():
        idx = torch.nonzero(bad)[:3, 0]
        for k in idx: print("  pixel", int(k), yy[k].cpu().numpy())
