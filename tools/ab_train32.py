"""Same-box A/B of the fp32 train step (64 x 512 x 512, the reference's training precision) between builds of the library under tools/_ab/
(and the product library: name 'product'), each in its own process, interleaved."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    from ubdvss_amd import _lib
    if sys.argv[2] != "product": _lib.LIB_PATH = os.path.join(ROOT, "tools", "_ab", sys.argv[2])
    from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic
    torch.cuda.set_device(0)
    dt = os.environ.get("DT", "float32")
    m = Model(NetConfig(grey=False), dtype=dt, seed=1)
    tr = Trainer(m, Adam())
    lab = synthetic.rectangle_maps(30, 64, 128, 128)
    x = torch.from_numpy(synthetic.textured_images(31, lab, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
    y = torch.from_numpy(lab).cuda()
    for _ in range(40): tr.train_step_on_device(x, y)
    out = []
    for blk in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(60): tr.train_step_on_device(x, y)
        e1.record(); torch.cuda.synchronize()
        out.append(round(e0.elapsed_time(e1) / 60, 4))
    print(json.dumps(out), "loss", float(tr.loss[0]))
else:
    for rep in range(2):
        for lib in sys.argv[1:]:
            r = subprocess.run([sys.executable, __file__, "child", lib], capture_output=True, text=True)
            print(lib, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
