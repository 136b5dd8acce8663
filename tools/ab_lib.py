"""A/B of two builds of the library inside ONE process on one box (box-to-box timing differences reach 10 %): bf16 train step
and the cfg5 fp16 forward with two builds kept under tools/_ab/ (git-ignored scratch; cp ubdvss_amd/libubd_hip.so tools/_ab/new.so), interleaved."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    from ubdvss_amd import _lib
    _lib.LIB_PATH = sys.argv[2] if os.path.isabs(sys.argv[2]) else os.path.join(ROOT, "tools", "_ab", sys.argv[2])
    from ubdvss_amd import NetConfig, Model, ModelRunner, Trainer, Adam, synthetic
    torch.cuda.set_device(0)
    def timed(fn, reps):
        for _ in range(20): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    m = Model(NetConfig(grey=False), dtype="bfloat16", seed=1)
    tr = Trainer(m, Adam())
    lab = synthetic.rectangle_maps(30, 64, 128, 128)
    x = torch.from_numpy(synthetic.textured_images(31, lab, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
    y = torch.from_numpy(lab).cuda()
    m5 = Model(NetConfig(grey=False), dtype="float16", seed=1)
    x5 = torch.from_numpy(synthetic.noise_images(7, 8, 1024, 1024, 3)).cuda()
    mf = Model(NetConfig(grey=False), seed=1)
    xf = torch.from_numpy(synthetic.noise_images(2, 32, 512, 512, 3)).cuda()
    labs = synthetic.rectangle_maps(3, 32, 128, 128)
    xs = torch.from_numpy(synthetic.textured_images(4, labs, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
    runner = ModelRunner(NetConfig(grey=False), max_objects_per_image=1024, pipelined=True)
    out = {"train_ms": [], "cfg5_ms": [], "fwd32_ms": [], "step_ms": []}
    for _ in range(3):
        out["train_ms"].append(round(timed(lambda: tr.train_step_on_device(x, y), 100), 4))
        out["cfg5_ms"].append(round(timed(lambda: m5.predict_on_device(x5), 200), 4))
        out["fwd32_ms"].append(round(timed(lambda: mf.predict_on_device(xf), 200), 4))
        out["step_ms"].append(round(timed(lambda: runner.predict_on_device(mf, xs), 300), 4))
    print(json.dumps(out))
else:
    for rep in range(2):
        for lib in (sys.argv[1:] or ["old.so", "new.so"]):       # other builds: names inside tools/_ab/
            env = dict(os.environ)
            name = lib
            if "@" in lib:                                                    # libname@VAR=value: the child runs with that variable set
                name, kv = lib.split("@", 1)
                env[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
            r = subprocess.run([sys.executable, __file__, "child", name], capture_output=True, text=True, env=env)
            print(lib, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
