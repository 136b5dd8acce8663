"""Pipelined step: is the overhead over the bare forward pass CU contention or queue plumbing?  Variants of the side-stream work."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
sys.path.insert(0, ROOT)
from ubdvss_amd import NetConfig, Model, synthetic
from ubdvss_amd.model_runner import _DeviceEvent, _TorchEvent
from ubdvss_amd import _lib
import ctypes
torch.cuda.set_device(0)
cfg = NetConfig(grey=False)
m = Model(cfg, seed=1)
labs = synthetic.rectangle_maps(3, 32, 128, 128)
x = torch.from_numpy(synthetic.textured_images(4, labs, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
main = torch.cuda.current_stream()
side = torch.cuda.Stream(priority=-1)
slots = [{"lg": torch.empty((32, 128, 128, 1), device="cuda"), "fwd": _DeviceEvent(), "done": _TorchEvent(), "done_nf": _DeviceEvent(), "used": False, "used_nf": False,
          "out": m.alloc_postprocess_outputs(32, 128, 128, 1024), "out1": m.alloc_postprocess_outputs(1, 128, 128, 1024),
          "out8": m.alloc_postprocess_outputs(8, 128, 128, 1024), "outt": m.alloc_postprocess_outputs(1, 8, 8, 16),
          "tiny": torch.zeros((1, 8, 8, 1), device="cuda")} for _ in range(2)]
step = [0]
def run(kind):
    s = slots[step[0] & 1]; step[0] += 1
    if kind == "full_nofence":
        if s["used_nf"] and not s["done_nf"].query(): s["done_nf"].wait(main)
        lg = m.predict_on_device(x, out=s["lg"])
        s["fwd"].record(main); s["fwd"].wait(side)
        with torch.cuda.stream(side):
            m.postprocess_on_device(lg, 0.0, 4, 5, cap=1024, outputs=s["out"])
            s["done_nf"].record(side)
        s["used_nf"] = True
        return
    if kind == "full_norecord":
        lg = m.predict_on_device(x, out=s["lg"])
        s["fwd"].record(main); s["fwd"].wait(side)
        with torch.cuda.stream(side):
            m.postprocess_on_device(lg, 0.0, 4, 5, cap=1024, outputs=s["out"])
        return
    if s["used"] and not s["done"].query(): s["done"].wait(main)
    lg = m.predict_on_device(x, out=s["lg"])
    if kind == "none": return
    s["fwd"].record(main); s["fwd"].wait(side)
    with torch.cuda.stream(side):
        if kind == "full": m.postprocess_on_device(lg, 0.0, 4, 5, cap=1024, outputs=s["out"])
        elif kind == "one": m.postprocess_on_device(lg[:1], 0.0, 4, 5, cap=1024, outputs=s["out1"])
        elif kind == "tiny": m.postprocess_on_device(s["tiny"], 0.0, 4, 5, cap=16, outputs=s["outt"])
        elif kind == "delay": _lib.check(_lib.load().ubd_stream_delay(ctypes.c_void_p(side.cuda_stream), 60), "delay")
        elif kind == "eight": m.postprocess_on_device(lg[:8], 0.0, 4, 5, cap=1024, outputs=s["out8"])
        s["done"].record(side)
    s["used"] = True
def timed(fn, reps=400):
    for _ in range(300): fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        t.append(round(e0.elapsed_time(e1) / reps, 4))
    return t
for kind in ("none", "events", "tiny", "delay", "one", "full"):
    print(kind, timed(lambda: run(kind)), flush=True)
