"""A/B of the two forward forms of the Winograd dilated layer in ONE process: products on the fp32 MFMA (wino.hip, UBD_DILCONV=wino32)
against three-way bf16 split products on the bf16 MFMA (wino6.hip, the default).  UBD_DILCONV is read when a handle is created, so the
two models are created under different settings.  Prints, per layer, the error of each form against an fp64 convolution on the host
(relative to max |y|) and the time per launch at 32 x 128 x 128 x 24 (HIP events around 300 back-to-back launches)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, _lib          # noqa: E402

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


def layer(mws, k, x, y):
    m, ws, st = mws
    n, hh, ww, _ = x.shape
    _lib.check(lib.ubd_dilated_layer(m._h, m.params.data_ptr(), k, x.data_ptr(), y.data_ptr(), n, hh, ww, ws.data_ptr(), st), "dil")


A = make("wino32")
B = make("")
os.environ.pop("UBD_DILCONV", None)
dil = [1, 2, 4, 8, 16, 1]
# ---- accuracy against fp64 on the host (2 x 72 x 100: ragged against every tile size)
rng = np.random.default_rng(5)
for scale in (1.0, 255.0):
    xs = (rng.random((2, 72, 100, 24), dtype=np.float32) - 0.3) * scale
    x = torch.from_numpy(xs).cuda()
    p = A[0].params.cpu().numpy().astype(np.float64)
    off = 3 * 3 * 3 + 3 * 24 + 24 + 2 * (9 * 24 + 24 * 24 + 24)
    for k in range(6):
        wk = torch.from_numpy(p[off:off + 5184].reshape(3, 3, 24, 24)).permute(3, 2, 0, 1).contiguous()
        bk = torch.from_numpy(p[off + 5184:off + 5208])
        off += 5208
        ref = torch.relu(torch.nn.functional.conv2d(torch.from_numpy(xs.astype(np.float64)).permute(0, 3, 1, 2), wk, bk, padding=dil[k], dilation=dil[k])).permute(0, 2, 3, 1).numpy()
        out = []
        for mws in (A, B):
            y = torch.empty_like(x)
            layer(mws, k, x, y)
            torch.cuda.synchronize()
            out.append(np.abs(y.cpu().numpy().astype(np.float64) - ref).max() / np.abs(ref).max())
        print(f"scale {scale:5.0f} layer {k} d {dil[k]:2d}: rel err fp32-MFMA {out[0]:.2e}  bf16x6 {out[1]:.2e}", flush=True)
# bit-exact homogeneity: f(2x) = 2 f(x) with zero biases is not available here (biases are in the params); compare A and B directly instead
x = torch.rand((32, 128, 128, 24), device="cuda") - 0.3
ya, yb = torch.empty_like(x), torch.empty_like(x)
for k in range(6):
    layer(A, k, x, ya); layer(B, k, x, yb)
    torch.cuda.synchronize()
    print(f"layer {k}: max |fp32-MFMA - bf16x6| = {(ya - yb).abs().max().item():.3e} (max |y| {ya.abs().max().item():.3f})", flush=True)
# ---- timing
n = int(os.environ.get("N", 32))
x = torch.rand((n, 128, 128, 24), device="cuda") - 0.3
y = torch.empty_like(x)
for _ in range(600): layer(B, 2, x, y)
for rep in range(2):
    for name, mws in (("fp32-MFMA", A), ("bf16x6", B)):
        res = []
        for k in range(6):
            for _ in range(30): layer(mws, k, x, y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(300): layer(mws, k, x, y)
            e1.record(); torch.cuda.synchronize()
            res.append(round(e0.elapsed_time(e1) / 300 * 1e3, 2))
        print(f"{name:10s} us per layer {res}  mean {sum(res) / 6:.2f}", flush=True)
# ---- whole forward pass
xin = torch.rand((32, 512, 512, 3), device="cuda")
for name, m in (("fp32-MFMA", A[0]), ("bf16x6", B[0])):
    for _ in range(200): m.predict_on_device(xin)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(500): m.predict_on_device(xin)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:10s} net ms {e0.elapsed_time(e1) / 500:.4f}", flush=True)
