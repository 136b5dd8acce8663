"""Same-process A/B of the Winograd dilated layer (ubd_dilated_layer, 32 x 128 x 128 x 24) between launcher variants selected by an
environment variable that the launcher reads ONCE per process -- so each variant runs in a child process, interleaved rounds."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import ctypes, torch
    sys.path.insert(0, ROOT)
    from ubdvss_amd import NetConfig, Model, _lib
    torch.cuda.set_device(0)
    model = Model(NetConfig(grey=False), seed=1)
    lib = _lib.load()
    n, side = int(os.environ.get("N", 32)), 128
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
    xin = torch.rand((32, 512, 512, 3), device="cuda")
    for _ in range(200): model.predict_on_device(xin)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(500): model.predict_on_device(xin)
    e1.record(); torch.cuda.synchronize()
    print("layers us", out, "mean", round(sum(out) / 6, 2), " net ms", round(e0.elapsed_time(e1) / 500, 4), flush=True)
else:
    variants = sys.argv[1:] or ["UBD_WINO_WPB=4", "UBD_WINO_WPB=12"]
    for rep in range(2):
        for v in variants:
            env = dict(os.environ)
            k, val = v.split("=", 1)
            env[k] = val
            r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=env)
            print(v, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
