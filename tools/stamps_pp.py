"""In-kernel phase timing of pp_front_lds_kernel (diagnostic build): s_memtime of thread 0 after every block barrier."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ubdvss_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_ab", os.environ.get("DIAG_LIB", "libubd_hip_diag.so"))
from ubdvss_amd import NetConfig, Model, synthetic
torch.cuda.set_device(0)
lib = _lib.load()
m = Model(NetConfig(grey=False), seed=1)
labels = synthetic.rectangle_maps(3, 32, 128, 128)
for name, lg in (("rectangle maps", synthetic.logits_from_maps(labels, 0, seed=5)), ("net-like maps (97 % foreground, hundreds of holes)", np.where(np.random.default_rng(2).random((32, 128, 128, 1)) < 0.02, -1.0, 1.0).astype(np.float32)), ("noise maps (p=0.5)", np.where(np.random.default_rng(1).random((32, 128, 128, 1)) < 0.5, 1.0, -1.0).astype(np.float32))):
    lt = torch.from_numpy(lg).cuda()
    for _ in range(200): m.postprocess_on_device(lt, 0.0, 4, 5, cap=1024)
    st = torch.zeros((32, 16), dtype=torch.int64, device="cuda")
    lib.ubd_debug_set_stamps_pp.argtypes = [ctypes.c_void_p]; lib.ubd_debug_set_stamps_pp.restype = None
    lib.ubd_debug_set_stamps_pp(st.data_ptr())
    m.postprocess_on_device(lt, 0.0, 4, 5, cap=1024); torch.cuda.synchronize()
    lib.ubd_debug_set_stamps_pp(None)
    s = st.cpu().numpy()
    k = int((s[0] > 0).sum())
    seg = np.diff(s[:, :k], axis=1)
    names = ["init", "merge", "flatten", "roots", "owner", "zero area", "area", "keep", "rows init", "extents", "boxes + vote", "emit"]
    print(name, "total cycles median", int(np.median(s[:, k - 1] - s[:, 0])))
    print("   " + "  ".join(f"{names[i] if i < len(names) else i}: {int(np.median(seg[:, i]))}" for i in range(k - 1)))
