"""In-kernel timing of dilconv_wino6_kernel<0> (diagnostic build): s_memtime per wave at entry (0), at the start of the wave's SECOND group (1), after its
transform rows a = 0..3 (2..5), after its epilogue (6), at exit (7).  Launch = one dilated layer on 32 x 128 x 128 x 24 (bench shape)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ubdvss_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_ab", os.environ.get("DIAG_LIB", "libubd_hip_diag.so"))
from ubdvss_amd import NetConfig, Model
torch.cuda.set_device(0)
lib = _lib.load()
m = Model(NetConfig(grey=False), seed=1)
n, mh, mw = int(os.environ.get("N", 32)), 128, 128
a = torch.rand((n, mh, mw, 24), device="cuda") - 0.3
b = torch.empty_like(a)
ws = torch.empty(int(lib.ubd_forward_workspace_bytes(m._h, 1, 4, 4)), dtype=torch.uint8, device="cuda")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.ubd_pack_weights(m._h, m.params.data_ptr(), ws.data_ptr(), ws.numel(), stream), "pack")
def run(layer):
    _lib.check(lib.ubd_dilated_layer(m._h, m.params.data_ptr(), layer, a.data_ptr(), b.data_ptr(), n, mh, mw, ws.data_ptr(), stream), "dil")
for _ in range(500): run(2)
lib.ubd_debug_set_stamps_wino6.argtypes = [ctypes.c_void_p]; lib.ubd_debug_set_stamps_wino6.restype = None
for layer in (0, 2, 4):
    st = torch.zeros((4096 * 2, 8), dtype=torch.int64, device="cuda")
    for _ in range(50): run(layer)
    lib.ubd_debug_set_stamps_wino6(st.data_ptr())
    run(layer); torch.cuda.synchronize()
    lib.ubd_debug_set_stamps_wino6(None)
    s = st.cpu().numpy().astype(np.int64)
    s = s[(s[:, 0] > 0) & (s[:, 6] > 0)]
    med = lambda v: int(np.median(v))
    print(f"layer {layer} (d={[1,2,4,8,16,1][layer]}): waves {len(s)}; entry -> start of group 2: {med(s[:,1]-s[:,0])}; rows a=0..3 of group 2: "
          f"{[med(s[:,2]-s[:,1]), med(s[:,3]-s[:,2]), med(s[:,4]-s[:,3]), med(s[:,5]-s[:,4])]}; epilogue {med(s[:,6]-s[:,5])}; group 2 total {med(s[:,6]-s[:,1])}; entry -> exit {med(s[:,7]-s[:,0])} ticks")
