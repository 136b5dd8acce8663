"""Micro-benchmark of single kernels through the layer-level C-ABI entry points (used under rocprofv3)."""
import argparse
import ctypes
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--side", type=int, default=128)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--layers", type=str, default="0,1,2,3,4,5")
args = ap.parse_args()
torch.cuda.set_device(0)
model = Model(NetConfig(grey=False), seed=1)
lib = _lib.load()
dev = model.device
x = torch.rand((args.n, args.side, args.side, 24), device=dev) - 0.3
y = torch.empty_like(x)
ws = torch.empty(int(lib.ubd_forward_workspace_bytes(model._h, 1, 4, 4)), dtype=torch.uint8, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.ubd_pack_weights(model._h, model.params.data_ptr(), ws.data_ptr(), ws.numel(), st), "pack")
for layer in [int(v) for v in args.layers.split(",")]:
    f = lambda: _lib.check(lib.ubd_dilated_layer(model._h, model.params.data_ptr(), layer, x.data_ptr(), y.data_ptr(), args.n, args.side, args.side, ws.data_ptr(), st), "dil")
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / args.reps * 1e3
    fl = 2.0 * 216 * 24 * args.n * args.side * args.side
    print(f"layer {layer}: {us:.1f} us  {fl / us / 1e6:.1f} TFLOP/s useful ({fl / us / 1e6 / 157.3 * 100:.1f}% of fp32 MFMA peak)")
