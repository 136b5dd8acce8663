"""Pipelined step: is the overhead over the bare forward pass CU contention or queue plumbing?  Variants of the side-stream work."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
sys.path.insert(0, ROOT)
from ubdvss_amd import NetConfig, Model, synthetic
from ubdvss_amd.model_runner import _DeviceEvent, _TorchEvent
torch.cuda.set_device(0)
cfg = NetConfig(grey=False)
m = Model(cfg, seed=1)
labs = synthetic.rectangle_maps(3, 32, 128, 128)
x = torch.from_numpy(synthetic.textured_images(4, labs, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
main = torch.cuda.current_stream()
side = torch.cuda.Stream(priority=-1)
slots = [{"lg": torch.empty((32, 128, 128, 1), device="cuda"), "fwd": _DeviceEvent(), "done": _TorchEvent(), "used": False,
          "out": m.alloc_postprocess_outputs(32, 128, 128, 1024), "out1": m.alloc_postprocess_outputs(1, 128, 128, 1024),
          "out8": m.alloc_postprocess_outputs(8, 128, 128, 1024)} for _ in range(2)]
step = [0]
def run(kind):
    s = slots[step[0] & 1]; step[0] += 1
    if s["used"] and not s["done"].query(): s["done"].wait(main)
    lg = m.predict_on_device(x, out=s["lg"])
    if kind == "none": return
    s["fwd"].record(main); s["fwd"].wait(side)
    with torch.cuda.stream(side):
        if kind == "full": m.postprocess_on_device(lg, 0.0, 4, 5, cap=1024, outputs=s["out"])
        elif kind == "one": m.postprocess_on_device(lg[:1], 0.0, 4, 5, cap=1024, outputs=s["out1"])
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
for kind in ("none", "events", "one", "eight", "full"):
    print(kind, timed(lambda: run(kind)), flush=True)
