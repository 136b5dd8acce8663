"""In-kernel phase timing of stem123w_kernel (diagnostic build, tools/build_diag.sh): s_memtime stamps of every wave at the
phase boundaries of its first 16 tiles.  Prints median cycles per segment."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ubdvss_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_ab", "libubd_hip_diag.so")
from ubdvss_amd import NetConfig, Model, synthetic
os.environ["UBD_STEM"] = "fused123w"
torch.cuda.set_device(0)
lib = _lib.load()
m = Model(NetConfig(grey=False), seed=1)
x = torch.from_numpy(synthetic.noise_images(2, 32, 512, 512, 3)).cuda()
for _ in range(300): m.predict_on_device(x)
nblk = 256
st = torch.zeros((nblk, 16, 16, 8), dtype=torch.int64, device="cuda")
lib.ubd_debug_set_stamps.argtypes = [ctypes.c_void_p]; lib.ubd_debug_set_stamps.restype = None
lib.ubd_debug_set_stamps(st.data_ptr())
for _ in range(5): m.predict_on_device(x)
torch.cuda.synchronize()
lib.ubd_debug_set_stamps(None)
s = st.cpu().numpy().astype(np.int64)
names = ["request", "L2 unit", "wait cnt0", "L1 early", "wait cnt1", "L1 late / role", "patch+barrier"]
s = np.maximum.accumulate(s, axis=-1)  # slots a wave did not write keep the previous time
seg = np.diff(s[..., :8], axis=-1)              # (blk, wave, it, 7)
its = slice(2, 14)
print("cycles (s_memtime ticks = shader cycles), median over blocks x tiles 2..13, per wave:")
for w in range(16):
    print(f" wave {w}: " + "  ".join(f"{names[k]} {np.median(seg[:, w, its, k]):7.0f}" for k in range(7)))
per_tile = s[:, 0, 3:14, 0] - s[:, 0, 2:13, 0]
print("tile period (wave 0): median", np.median(per_tile), "p10", np.percentile(per_tile, 10), "p90", np.percentile(per_tile, 90))
tot = s[:, :, 13, 6].max() - s[:, :, 2, 0].min()
print("all segments sum (median, wave 0):", np.median(seg[:, 0, its, :].sum(-1)))
# units of two tiles: even tiles start a unit, odd tiles end one
for par, name in ((0, "first tile of a unit (even)"), (1, "last tile of a unit (odd)")):
    sl = slice(2 + par, 14, 2)
    print(name)
    for w in (0, 3, 4, 13, 14, 15):
        print(f"  wave {w}: " + "  ".join(f"{names[k]} {np.median(seg[:, w, sl, k]):7.0f}" for k in range(7)))
    nxt = s[:, 0, 3 + par:15:2, 0] - s[:, 0, 2 + par:14:2, 0]
    print("  tile period (wave 0): median", np.median(nxt), " mean", nxt.mean())
