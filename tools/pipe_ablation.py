"""Where the pipelined step's overhead over the bare forward pass goes: event flavour, stagger delay, postprocess on/off.
Each variant is a child process (the knobs are read when the runner is created)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    from ubdvss_amd import NetConfig, Model, ModelRunner, synthetic
    torch.cuda.set_device(0)
    cfg = NetConfig(grey=False)
    m = Model(cfg, seed=1)
    labs = synthetic.rectangle_maps(3, 32, 128, 128)
    x = torch.from_numpy(synthetic.textured_images(4, labs, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
    runner = ModelRunner(cfg, max_objects_per_image=1024, pipelined=True)
    serial = ModelRunner(cfg, max_objects_per_image=1024, pipelined=False)
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
    print(json.dumps({"net": timed(lambda: m.predict_on_device(x)), "pipelined": timed(lambda: runner.predict_on_device(m, x)),
                      "serial": timed(lambda: serial.predict_on_device(m, x))}))
else:
    for name, env in (("hip events, stagger 3", {}), ("torch events, stagger 3", {"UBD_PIPE_EVENTS": "torch"}),
                      ("hip events, stagger 0", {"UBD_STAGGER_US": "0"}), ("hip events, stagger 1", {"UBD_STAGGER_US": "1"}),
                      ("hip events, stagger 6", {"UBD_STAGGER_US": "6"})):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=e)
        print(name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
