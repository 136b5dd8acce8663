"""Shape sweep of the kernel variants that must agree BIT FOR BIT (run on the GPU box):
  * bf16 / fp16 forward: LDS-staged dilated kernel vs direct kernel (UBD_DILCONV16=direct)
  * bf16 train step: data gradient fused into the weight-gradient kernel (UBD_DILBWD=pair8: 8-wide tiles for narrow sub-grids) vs the two-kernel path
    (UBD_DILBWD=split); the default (pairs of 8-column sub-grids in 16-wide tiles) agrees with both to 1e-5 (another order of the weight-gradient sums)
  * fp32 inference: one-kernel stem vs three kernels (UBD_STEM=fused123 / unfused), fp32 and uint8 input
  * 16-bit passes: the stem in one kernel (L1 -> L2 -> L3, default) vs L1 -> L2 fused + L3 (UBD_STEM16=fused12) vs three kernels (UBD_STEM16=split),
    forward (fp32 and uint8 input) and bf16 train step
  * bf16 train step: chained partial-sum reduction (default) vs the two batched launches (UBD_REDUCE=batched)
  * bf16 train step: L1 backward with its fp32 input patch by LDS-DMA (default) vs staged through registers (UBD_SEPB16_X=regs)
(UBD_VARIANT_RANDOM_SHAPES=n adds n random shapes)
on ragged and non-square shapes (sides are multiples of 4, maps not multiples of 16, narrow sub-grids)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic
torch.cuda.set_device(0)
shapes = [(1, 68, 140), (2, 100, 76), (1, 132, 260), (1, 516, 36), (2, 36, 516), (3, 200, 328), (1, 1028, 68), (2, 260, 260), (1, 64, 64), (5, 128, 192)]
_rng = np.random.default_rng(int(os.environ.get("UBD_VARIANT_SEED", "3")))
for _ in range(int(os.environ.get("UBD_VARIANT_RANDOM_SHAPES", "0"))):
    shapes.append((int(_rng.integers(1, 7)), 4 * int(_rng.integers(4, 100)), 4 * int(_rng.integers(4, 100))))
bad = 0
def model(env, **kw):
    for k in ("UBD_DILCONV16", "UBD_DILBWD", "UBD_STEM", "UBD_STEM16", "UBD_REDUCE", "UBD_SEPB16_X"): os.environ.pop(k, None)
    os.environ.update(env)
    return Model(NetConfig(grey=False), seed=7, **kw)
for (n, h, w) in shapes:
    x = torch.from_numpy(synthetic.noise_images(n + h, n, h, w, 3)).cuda()
    for dt in ("bfloat16", "float16"):
        a = model({}, dtype=dt).predict_on_device(x).clone()
        b = model({"UBD_DILCONV16": "direct"}, dtype=dt).predict_on_device(x).clone()
        ok = torch.equal(a, b) and bool(torch.isfinite(a).all()); bad += not ok
        print(f"{n}x{h}x{w} {dt:9s} forward staged == direct: {ok}", flush=True)
    lab = synthetic.rectangle_maps(n + w, n, h // 4, w // 4)
    xt = torch.from_numpy(synthetic.textured_images(h, lab, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
    y = torch.from_numpy(lab).cuda()
    g = []
    for env in ({}, {"UBD_DILBWD": "split"}, {"UBD_DILBWD": "pair8"}):
        t = Trainer(model(env, dtype="bfloat16"), Adam()); t.backward_on_device(xt, y); g.append(t.grads.clone())
    ok = torch.equal(g[2], g[1]) and bool(torch.isfinite(g[0]).all()); bad += not ok
    print(f"{n}x{h}x{w} bf16 train fused (8-wide tiles for narrow sub-grids) == split: {ok}", flush=True)
    # default: sub-grids of exactly 8 columns go in pairs into 16-wide tiles -- the same products in another order of the weight-gradient sums
    err = float((g[0].double() - g[2].double()).abs().max()) / max(float(g[2].abs().max()), 1e-30)
    ok = err <= 1e-5; bad += not ok
    print(f"{n}x{h}x{w} bf16 train paired sub-grids vs 8-wide tiles: max deviation {err:.1e} of the largest gradient: {ok}", flush=True)
    for env, name in (({"UBD_STEM16": "split"}, "one-kernel stem == three kernels"), ({"UBD_STEM16": "fused12"}, "one-kernel stem == L1 -> L2 fused + L3"),
                      ({"UBD_REDUCE": "batched"}, "chained reduction == batched launches"),
                      ({"UBD_SEPB16_X": "regs"}, "L1 backward input by LDS-DMA == through registers")):
        t = Trainer(model(env, dtype="bfloat16"), Adam()); t.backward_on_device(xt, y)
        ok = torch.equal(g[0], t.grads); bad += not ok
        print(f"{n}x{h}x{w} bf16 train, {name}: {ok}", flush=True)
    x8 = torch.from_numpy(np.random.default_rng(n * h).integers(0, 256, (n, h, w, 3), dtype=np.uint8)).cuda()
    for name, inp in (("fp32", x), ("uint8", x8)):
        a = model({"UBD_STEM": "fused123"}).predict_on_device(inp).clone()
        b = model({"UBD_STEM": "unfused"}).predict_on_device(inp).clone()
        ok = torch.equal(a, b) and bool(torch.isfinite(a).all()); bad += not ok
        print(f"{n}x{h}x{w} fp32 net, {name} input, stem fused123 == unfused: {ok}", flush=True)
        for dt in ("bfloat16", "float16"):
            a = model({}, dtype=dt).predict_on_device(inp).clone()
            for mode in ("split", "fused12"):
                b = model({"UBD_STEM16": mode}, dtype=dt).predict_on_device(inp).clone()
                ok = torch.equal(a, b) and bool(torch.isfinite(a).all()); bad += not ok
                print(f"{n}x{h}x{w} {dt} net, {name} input, one-kernel stem == {mode}: {ok}", flush=True)
# ---- repeat runs at the headline sizes: every launch of the persistent kernels must reproduce the first one bit for bit
REPS = int(os.environ.get("UBD_VARIANT_REPEATS", "200"))
x = torch.from_numpy(synthetic.noise_images(5, 32, 512, 512, 3)).cuda()
for name, env, kw in (("fp32 one-kernel stem + Winograd", {"UBD_STEM": "fused123"}, {}), ("fp16 one-kernel stem + staged dilated forward", {}, {"dtype": "float16"})):
    m = model(env, **kw)
    first = m.predict_on_device(x).clone()
    diff = sum(0 if torch.equal(m.predict_on_device(x), first) else 1 for _ in range(REPS))
    bad += diff
    print(f"{name}: {REPS} launches at 32 x 512 x 512, {diff} differ from the first", flush=True)
lab = synthetic.rectangle_maps(9, 64, 128, 128)
xt = torch.from_numpy(synthetic.textured_images(10, lab, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
y = torch.from_numpy(lab).cuda()
t = Trainer(model({}, dtype="bfloat16"), Adam())
t.backward_on_device(xt, y); g0 = t.grads.clone()
diff = 0
for _ in range(REPS):
    t.backward_on_device(xt, y)
    diff += 0 if torch.equal(t.grads, g0) else 1
bad += diff
print(f"bf16 train step (one-kernel stem, fused dilated backward, chained reduction): {REPS} passes at 64 x 512 x 512, {diff} differ from the first", flush=True)
print("MISMATCHES:", bad)
sys.exit(1 if bad else 0)
