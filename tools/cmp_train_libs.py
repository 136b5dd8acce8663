"""Bit-equality of the 16-bit train step between two builds of the library (a refactoring that must not change a single bit -- round 6: the
ReLU masks of the separable backward read as bits): each build runs in a child process on the same seeded batches (several shapes, ragged
widths, both 16-bit types, uint8 and fp32 input, 1 and 3 input channels), prints a digest of gradients, parameters and loss per step; the
parent compares.   python tools/cmp_train_libs.py tools/_ab/bwd_before.so product"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "child":
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic
    from ubdvss_amd.net import PreprocessingType
    MOBILENET_LIKE = PreprocessingType.MOBILENET_LIKE
    torch.cuda.set_device(0)
    shapes = [(2, 64, 64), (3, 72, 104), (1, 36, 52), (2, 128, 200), (5, 40, 296), (4, 256, 256), (1, 20, 28), (2, 516, 68)]
    for dtype in ("bfloat16", "float16"):
        for k, (n, H, W) in enumerate(shapes):
            grey, u8 = (k % 3 == 1), (k % 2 == 1)
            cfg = NetConfig(grey=grey, preprocessing=MOBILENET_LIKE) if u8 else NetConfig(grey=grey)
            m = Model(cfg, dtype=dtype, seed=11 + k)
            tr = Trainer(m, Adam(lr=1e-3))
            rng = np.random.default_rng(300 + k)
            labels = (rng.random((n, H // 4, W // 4)) < 0.3).astype(np.int32)
            img = rng.integers(0, 256, (n, H, W, 1 if grey else 3), dtype=np.uint8)
            x = torch.from_numpy(img if u8 else img.astype(np.float32) / 127.5 - 1.0).cuda()
            y = torch.from_numpy(labels).cuda()
            h = hashlib.sha256()
            for step in range(3):
                loss = tr.train_step_on_device(x, y)
                torch.cuda.synchronize()
                h.update(tr.grads.cpu().numpy().tobytes()); h.update(m.params.cpu().numpy().tobytes())
                h.update(np.asarray(loss.cpu() if hasattr(loss, "cpu") else loss, dtype=np.float32).tobytes())
            print("DIGEST", dtype, n, H, W, "grey" if grey else "rgb", "u8" if u8 else "f32", h.hexdigest()[:24], flush=True)
    sys.exit(0)

outs = []
for name in sys.argv[1:3]:
    env = dict(os.environ)
    if name != "product": env["UBD_LIB_PATH"] = os.path.join(ROOT, name)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", name], env=env, capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")]
    if r.returncode != 0 or not lines:
        print(r.stdout[-2000:], r.stderr[-3000:]); sys.exit(1)
    outs.append(lines)
bad = 0
for a, b in zip(*outs):
    same = a == b
    bad += not same
    print(("same   " if same else "DIFFER ") + a + ("" if same else "   |   " + b))
print("CMP_TRAIN_LIBS", "OK: bit-equal" if bad == 0 and len(outs[0]) == len(outs[1]) else f"FAILED ({bad})")
sys.exit(1 if bad else 0)
