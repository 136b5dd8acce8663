"""Times the dilated layers (ubd_dilated_layer, 32 x 128 x 128 x 24, HIP events around 300 launches) for several builds of the library, each in a
child process, interleaved: python tools/ab_layer_libs.py product w6_X.so ...   (names inside tools/_ab/; 'product' = ubdvss_amd/libubd_hip.so;
name@VAR=value sets an environment variable for that child).  Timing only: experiment builds may compute wrong values."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "child":
    import ctypes, torch
    sys.path.insert(0, ROOT)
    from ubdvss_amd import _lib
    if sys.argv[2] != "product":
        _lib.LIB_PATH = os.path.join(ROOT, "tools", "_ab", sys.argv[2])
    from ubdvss_amd import NetConfig, Model
    torch.cuda.set_device(0)
    model = Model(NetConfig(grey=False), seed=1)
    lib = _lib.load()
    n, side = int(os.environ.get("N", 32)), int(os.environ.get("SIDE", 128))
    x = torch.rand((n, side, side, 24), device="cuda") - 0.3
    y = torch.empty_like(x)
    ws = torch.empty(int(lib.ubd_forward_workspace_bytes(model._h, 1, 4, 4)), dtype=torch.uint8, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.ubd_pack_weights(model._h, model.params.data_ptr(), ws.data_ptr(), ws.numel(), st), "pack")
    def f(layer): _lib.check(lib.ubd_dilated_layer(model._h, model.params.data_ptr(), layer, x.data_ptr(), y.data_ptr(), n, side, side, ws.data_ptr(), st), "dil")
    for _ in range(600): f(2)
    out = []
    for layer in range(6):
        for _ in range(30): f(layer)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): f(layer)
        e1.record(); torch.cuda.synchronize()
        out.append(round(e0.elapsed_time(e1) / 300 * 1e3, 2))
    print("layers us", out, "mean", round(sum(out) / 6, 2), flush=True)
else:
    for rep in range(int(os.environ.get("REPS", 2))):
        for v in sys.argv[1:]:
            env = dict(os.environ)
            name = v
            if "@" in v:
                name, kv = v.split("@", 1)
                env[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
            r = subprocess.run([sys.executable, __file__, "child", name], capture_output=True, text=True, env=env)
            print(f"{v:28s}", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
