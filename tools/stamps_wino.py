"""In-kernel timing of dilconv_wino_kernel<0> (diagnostic build): per wave s_memtime at kernel entry, after the U copy +
barrier, after each group, at exit.  Launch = one dilated layer on 32 x 128 x 128 x 24 (bench shape)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ubdvss_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_ab", "libubd_hip_diag.so")
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
lib.ubd_debug_set_stamps_wino.argtypes = [ctypes.c_void_p]; lib.ubd_debug_set_stamps_wino.restype = None
for layer in (0, 2, 4):
    st = torch.zeros((768 * 4, 8), dtype=torch.int64, device="cuda")
    for _ in range(50): run(layer)
    lib.ubd_debug_set_stamps_wino(st.data_ptr())
    run(layer); torch.cuda.synchronize()
    lib.ubd_debug_set_stamps_wino(None)
    s = st.cpu().numpy().astype(np.int64)
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    end = s[:, 7]
    ng = (s[:, 2:7] > 0).sum(1)
    print(f"layer {layer} (d={[1,2,4,8,16,1][layer]}): waves {len(s)}; kernel span {end.max() - t0} cycles; entry spread p50 {np.median(s[:,0]-t0):.0f} p95 {np.percentile(s[:,0]-t0,95):.0f}")
    print(f"   prologue (entry -> after U copy + barrier): p50 {np.median(s[:,1]-s[:,0]):.0f} p95 {np.percentile(s[:,1]-s[:,0],95):.0f}")
    for k in (2, 3):
        sel = ng == k
        if sel.any():
            per = [(np.median(s[sel, 2 + j] - s[sel, 1 + j])) for j in range(k)]
            print(f"   waves with {k} groups: {sel.sum()}; cycles per group {per}; exit at p50 {np.median(end[sel]-t0):.0f} p95 {np.percentile(end[sel]-t0,95):.0f}")
