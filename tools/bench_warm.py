"""How many back-to-back repetitions the chip needs before timings settle (clock ramp after an idle gap)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, ModelRunner, synthetic
torch.cuda.set_device(0)
cfg = NetConfig(grey=False)
m = Model(cfg, seed=1)
runner = ModelRunner(cfg, pixel_threshold=0.5, max_objects_per_image=1024, pipelined=True)
labels = synthetic.rectangle_maps(3, 32, 128, 128)
x = torch.from_numpy(synthetic.textured_images(4, labels, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
def timed(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for name, fn in (("net", lambda: m.predict_on_device(x)), ("net+ccl", lambda: runner.predict_on_device(m, x))):
    for reps in (20, 100, 500, 2000, 20, 100, 2000):
        time.sleep(0.5)
        print(f"{name}: reps {reps}: {timed(fn, reps):.4f} ms/step", flush=True)
