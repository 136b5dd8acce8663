"""One-kernel stem at two vs four waves per SIMD (UBD_STEM=fused123 / fused123w, read when the handle is created): bit equality on a
set of shapes (RGB / grey, fp32 / uint8, few CUs) and interleaved timings of the net alone and of the pipelined image -> quads step."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "_ab", "libubd_hip_diag.so")   # the experimental kernel only exists in the diagnostic build (tools/build_diag.sh)
from ubdvss_amd import NetConfig, Model, ModelRunner, PreprocessingType, synthetic
torch.cuda.set_device(0)
MODES = tuple((os.environ.get("AB_MODES") or "fused123,fused123w").split(","))

def model(mode, cfg, seed=1, few=False):
    os.environ["UBD_STEM"] = mode
    if few: os.environ["UBD_TEST_NUM_CUS"] = "2"
    else: os.environ.pop("UBD_TEST_NUM_CUS", None)
    return Model(cfg, seed=seed)

bad = 0
for grey, u8, n, hh, ww, few in ((False, False, 2, 128, 128, False), (True, False, 2, 72, 100, False), (False, True, 3, 64, 200, True), (False, False, 1, 8, 8, False),
                                 (True, True, 2, 4, 36, False), (False, False, 2, 96, 512, True), (False, False, 4, 512, 512, False), (True, False, 2, 136, 72, True),
                                 (False, True, 2, 256, 320, False), (False, False, 3, 260, 516, True)):
    cfg = NetConfig(grey=grey, preprocessing=PreprocessingType.MOBILENET_LIKE) if u8 else NetConfig(grey=grey)
    cin = 1 if grey else 3
    x = synthetic.noise_images(19, n, hh, ww, cin, as_float=not u8)
    outs = [model(m, cfg, few=few).predict(x) for m in MODES]
    same = all(np.array_equal(outs[0], o) for o in outs[1:])
    bad += not same
    print(f"grey {grey} uint8 {u8} {n} x {hh} x {ww} few_cus {few}: {'bit-identical' if same else 'DIFFERENT max ' + str(max(float(np.abs(outs[0] - o).max()) for o in outs[1:]))}", flush=True)
os.environ.pop("UBD_TEST_NUM_CUS", None)
print("BIT-EQUALITY", "OK" if not bad else f"FAILED ({bad})", flush=True)

def timed(fn, reps):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
cfg = NetConfig(grey=False)
x = torch.from_numpy(synthetic.noise_images(2, 32, 512, 512, 3)).cuda()
labs = synthetic.rectangle_maps(3, 32, 128, 128)
xs = torch.from_numpy(synthetic.textured_images(4, labs, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
ms = {m: model(m, cfg) for m in MODES}
runners = {m: ModelRunner(cfg, max_objects_per_image=1024, pipelined=True) for m in MODES}
for _ in range(300): ms[MODES[0]].predict_on_device(x)
for rep in range(int(os.environ.get("AB_ROUNDS", "3"))):
    for m in MODES:
        print(f"{m}: net {timed(lambda: ms[m].predict_on_device(x), 500):.4f} ms   pipelined step {timed(lambda: runners[m].predict_on_device(ms[m], xs), 500):.4f} ms", flush=True)
