"""Fuzz of one dilated layer through ubd_dilated_layer: the split-product form (wino6.hip, default) against the fp32-MFMA form (UBD_DILCONV=wino32) on
random map sizes (1 x 1 .. 70 x 150, any residue), batch sizes and all six layers (dilations 1, 2, 4, 8, 16, 1).  Bound: 4e-6 of the layer's largest
value (the two forms differ by rounding only).  FUZZ_CASES scales it (default 300)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, _lib
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
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", 1)))
worst, bad = 0.0, 0
cases = int(os.environ.get("FUZZ_CASES", 300))
for c in range(cases):
    n = int(rng.integers(1, 5)); hh = int(rng.integers(1, 71)); ww = int(rng.integers(1, 151))
    if c % 7 == 0: hh, ww = int(rng.integers(1, 6)), int(rng.integers(1, 6))
    k = int(rng.integers(0, 6))
    x = (torch.rand((n, hh, ww, 24), device="cuda") - 0.3) * float(rng.choice([1.0, 100.0, 1e-3]))
    ys = []
    for m, ws, st in (A, B):
        y = torch.full_like(x, float("nan"))
        _lib.check(lib.ubd_dilated_layer(m._h, m.params.data_ptr(), k, x.data_ptr(), y.data_ptr(), n, hh, ww, ws.data_ptr(), st), "dil")
        ys.append(y)
    torch.cuda.synchronize()
    if torch.isnan(ys[1]).any() or torch.isnan(ys[0]).any():
        bad += 1; print("case", c, (n, hh, ww, k), "NaN / unwritten outputs:", int(torch.isnan(ys[0]).sum()), int(torch.isnan(ys[1]).sum())); continue
    e = float((ys[0] - ys[1]).abs().max()) / max(float(ys[0].abs().max()), 1e-30)
    worst = max(worst, e)
    if e > 4e-6:
        bad += 1; print("case", c, (n, hh, ww, k), "rel diff", e)
print(f"{cases} cases, worst relative difference {worst:.2e}, failures {bad}")
sys.exit(1 if bad else 0)
