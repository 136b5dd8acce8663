"""Steady-state A/B of the bf16 train step (MODE=cfg5: the 8 x 1024 x 1024 fp16 forward, blocks of 1000 passes): every build under tools/_ab/ named on the command line
runs BLOCKS blocks in its own process (interleaved, two rounds)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    from ubdvss_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "_ab", sys.argv[2])
    from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic
    torch.cuda.set_device(0)
    m = Model(NetConfig(grey=False), dtype="bfloat16", seed=1)
    tr = Trainer(m, Adam())
    lab = synthetic.rectangle_maps(30, 64, 128, 128)
    x = torch.from_numpy(synthetic.textured_images(31, lab, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
    y = torch.from_numpy(lab).cuda()
    step, per = (lambda: tr.train_step_on_device(x, y)), 200
    if os.environ.get("MODE") == "cfg5":
        m5 = Model(NetConfig(grey=False), dtype="float16", seed=1)
        x5 = torch.from_numpy(synthetic.noise_images(7, 8, 1024, 1024, 3)).cuda()
        step, per = (lambda: m5.predict_on_device(x5)), 1000
    if os.environ.get("MODE") == "fwd32":                        # the headline's net: 32 x 512 x 512 fp32 forward
        mf = Model(NetConfig(grey=False), seed=1)
        xf = torch.from_numpy(synthetic.noise_images(2, 32, 512, 512, 3)).cuda()
        step, per = (lambda: mf.predict_on_device(xf)), 500
    for _ in range(50): step()
    out = []
    for blk in range(int(os.environ.get("BLOCKS", "12"))):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(per): step()
        e1.record(); torch.cuda.synchronize()
        out.append(round(e0.elapsed_time(e1) / per, 4))
    print(json.dumps(out))
else:
    for rep in range(2):
        for lib in sys.argv[1:]:
            r = subprocess.run([sys.executable, __file__, "child", lib], capture_output=True, text=True)
            print(lib, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
