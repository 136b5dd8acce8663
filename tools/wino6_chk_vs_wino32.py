import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ab", sys.argv[1])
from ubdvss_amd import NetConfig, Model
torch.cuda.set_device(0)
lib = _lib.load()
def make(env):
    if env: os.environ["UBD_DILCONV"] = env
    else: os.environ.pop("UBD_DILCONV", None)
    m = Model(NetConfig(grey=False), seed=1)
    ws = torch.empty(int(lib.ubd_forward_workspace_bytes(m._h, 1, 4, 4)), dtype=torch.uint8, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.ubd_pack_weights(m._h, m.params.data_ptr(), ws.data_ptr(), ws.numel(), st), "pack")
    return m, ws, st
A, B = make("wino32"), make("")
x = torch.rand((2, 72, 100, 24), device="cuda") - 0.3
ya, yb = torch.empty_like(x), torch.empty_like(x)
for k in range(6):
    for (m, ws, st), y in ((A, ya), (B, yb)):
        _lib.check(lib.ubd_dilated_layer(m._h, m.params.data_ptr(), k, x.data_ptr(), y.data_ptr(), 2, 72, 100, ws.data_ptr(), st), "dil")
    torch.cuda.synchronize()
    e = (ya - yb).abs()
    print(sys.argv[1:] or "product", "layer", k, "max diff", e.max().item(), "per channel", e.amax(dim=(0, 1, 2)).cpu().numpy().round(2)[:24])
