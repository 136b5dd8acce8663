"""GPU parity of the 16-bit activation forward paths (bf16 / fp16 storage + 16-bit MFMA in the dilated layers,
fp32 accumulation, fp32 logits) -- BASELINE.json configs[2] (bf16) and configs[4] (fp16, 1024x1024).
Two gates per case:
  * vs the oracle run with the SAME storage rounding (activations and dilated kernels rounded to the 16-bit
    type after every layer): max |diff| <= 4e-3 (fp16) / 2e-2 (bf16: one storage ulp is 2^-8 = 0.4 %, and a
    rounding tie that flips because fp32 and fp64 accumulate in different orders propagates through the
    remaining layers) * max|logit| + 1e-5,
  * vs the plain fp64 oracle: <= 5e-3 * max|logit| for fp16 and <= 2.5e-2 * max|logit| for bf16 (3 x the largest error
    observed on MI355X, see GATE_FP64),
  * the binary detection map equals the oracle's on every pixel whose oracle logit is further than the gate from the
    threshold; the number of pixels inside that margin is printed."""
import os

import numpy as np
import pytest
import torch

from oracle import net_numpy as onet
from ubdvss_amd import NetConfig, Model, ModelRunner, synthetic

pytestmark = pytest.mark.gpu


# Gates relative to max|logit| of the case, set from the errors observed on MI355X over the cases of this file plus
# 2 x 512 x 512 and 2 x 1024 x 1024 (tools/parity16_report.py -> profiles/r02_parity16.json): largest observed
# bf16 9.4e-3 (same-rounding oracle) / 7.9e-3 (fp64 oracle), fp16 1.8e-3 / 1.6e-3.  A bf16 storage rounding that flips
# between two correct evaluations (fp32 vs fp64 accumulation order) moves a value by 2^-8 relative and propagates, which is
# why the same-rounding oracle is no closer than the fp64 one at the large shapes; gates = 2-3 x the observed maxima.
GATE_SAME_ROUNDING = {"bfloat16": 2e-2, "float16": 4e-3}        # vs the oracle with the same storage roundings
GATE_FP64 = {"bfloat16": 2.5e-2, "float16": 5e-3}               # vs the plain fp64 oracle (round 1: 8e-2 / 2e-2)


def measure(dtype, cin, ncls, fml, n, hh, ww, seed, thr=0.0):
    """Runs one case; returns the observed errors and the binary-map agreement figures."""
    cfg = NetConfig(class_names=[f"c{i}" for i in range(ncls)] if ncls else None, grey=(cin == 1), fml_compatible=fml)
    w = onet.init_weights(seed, cin, ncls, bias_scale=0.2)
    m = Model(cfg, dtype=dtype)
    m.set_weights(w)
    x = synthetic.noise_images(seed + 1, n, hh, ww, cin)
    lg = m.predict(x)
    ref16 = onet.forward(x.astype(np.float64), w, fml, act_dtype=dtype)
    ref = onet.forward(x.astype(np.float64), w, fml)
    assert lg.dtype == np.float32 and lg.shape == ref.shape
    scale = float(np.abs(ref).max())
    out = {"scale": scale, "e16": float(np.abs(lg - ref16).max()) / scale, "e64": float(np.abs(lg - ref).max()) / scale}
    # binary detection map (model_runner.py:124) vs the map of the oracle logits, outside a margin around the threshold
    got = lg[..., 0] > thr
    for name, r, gate in (("same_rounding", ref16, GATE_SAME_ROUNDING[dtype]), ("fp64", ref, GATE_FP64[dtype])):
        margin = gate * scale
        decided = np.abs(r[..., 0] - thr) > margin
        out[f"map_{name}"] = {"margin": margin, "pixels": int(decided.size), "in_margin": int((~decided).sum()),
                              "mismatch_outside_margin": int((got != (r[..., 0] > thr))[decided].sum()),
                              "mismatch_anywhere": int((got != (r[..., 0] > thr)).sum())}
    return out


def _run(dtype, cin, ncls, fml, n, hh, ww, seed):
    r = measure(dtype, cin, ncls, fml, n, hh, ww, seed)
    print(f"{dtype} {cin}ch {ncls}cls {n}x{hh}x{ww}: e16 {r['e16']:.2e} e64 {r['e64']:.2e} of max|logit| {r['scale']:.3f}; "
          f"map vs same-rounding oracle: {r['map_same_rounding']['mismatch_anywhere']} of {r['map_same_rounding']['pixels']} pixels differ, "
          f"{r['map_same_rounding']['in_margin']} inside the margin; vs fp64: {r['map_fp64']['mismatch_anywhere']} differ, "
          f"{r['map_fp64']['in_margin']} inside the margin")
    assert r["e16"] <= GATE_SAME_ROUNDING[dtype] + 1e-5 / r["scale"], r
    assert r["e64"] <= GATE_FP64[dtype], r
    # the maps agree on every pixel whose oracle logit is further than the stated bound from the threshold
    assert r["map_same_rounding"]["mismatch_outside_margin"] == 0 and r["map_fp64"]["mismatch_outside_margin"] == 0, r
    return r


def test_golden_16bit_fixture(golden_dir):
    """Committed same-rounding oracle logits (tests/golden/net16_rgb_cls2.npz): catches a regression of the 16-bit
    kernels without running the oracle; gates as above."""
    d = np.load(os.path.join(golden_dir, "net16_rgb_cls2.npz"))
    cfg = NetConfig(class_names=["c0", "c1"], grey=False)
    for dtype in ("bfloat16", "float16"):
        m = Model(cfg, dtype=dtype)
        m.params.copy_(torch.from_numpy(d["params"]))
        lg = m.predict(d["x"])
        scale = float(np.abs(d["logits_f64"]).max())
        assert np.abs(lg - d[f"logits_{dtype}"]).max() <= GATE_SAME_ROUNDING[dtype] * scale + 1e-5
        assert np.abs(lg - d["logits_f64"]).max() <= GATE_FP64[dtype] * scale


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("cin,ncls,fml,n,hh,ww", [(3, 0, True, 2, 128, 128), (1, 3, True, 1, 64, 192), (3, 2, False, 2, 72, 100)])
def test_forward16_vs_oracle(dtype, cin, ncls, fml, n, hh, ww):
    _run(dtype, cin, ncls, fml, n, hh, ww, 50 + cin + ncls)


def test_forward16_random_shape_soak():
    """Random small shapes (sides multiples of 4 from 16 to 160, 1-3 images, grey / RGB, with / without classes, both padding rules,
    both 16-bit types) against the rounding-aware and the fp64 oracle, same gates as above.  UBD_FWD16_SOAK_CASES scales it."""
    rng = np.random.default_rng(99)
    for case in range(int(os.environ.get("UBD_FWD16_SOAK_CASES", "12"))):
        dtype = str(rng.choice(["bfloat16", "float16"]))
        cin, ncls, fml = int(rng.choice([1, 3])), int(rng.choice([0, 0, 2])), bool(rng.integers(0, 2))
        n, hh, ww = int(rng.integers(1, 4)), 4 * int(rng.integers(4, 41)), 4 * int(rng.integers(4, 41))
        _run(dtype, cin, ncls, fml, n, hh, ww, 400 + case)


def test_cfg5_shape_fp16_1024():
    """configs[4]: 1024x1024 fp16 forward with dilation rates 1,2,4,8,16(,1); batch reduced to 2 for the oracle."""
    _run("float16", 3, 0, True, 2, 1024, 1024, 77)


def test_bf16_full_batch_property_and_postprocess():
    """configs[2] forward shape (64 x 512 x 512 x 3, bf16): image independence of the batch (bit-exact: same
    kernels, same per-image data) and the device postprocess running on the bf16 model's fp32 logits."""
    cfg = NetConfig(grey=False)
    m = Model(cfg, dtype="bfloat16", seed=1)
    x = torch.from_numpy(synthetic.noise_images(3, 64, 512, 512, 3)).cuda()
    a = m.predict_on_device(x).clone()
    b = m.predict_on_device(x[9:10].contiguous())
    assert torch.equal(a[9], b[0])
    out = ModelRunner(cfg).predict_on_device(m, x[:4].contiguous())
    assert out[4].shape == (4,)


@pytest.mark.parametrize("dtype,tol64", [("bfloat16", 4e-2), ("float16", 3e-2)])
@pytest.mark.parametrize("cin,ncls,fml,n,hh,ww", [(3, 0, True, 2, 64, 64), (1, 2, True, 2, 64, 96), (3, 2, False, 1, 128, 64),
                                                   (3, 0, True, 3, 72, 104)])
def test_train_step_16bit(dtype, tol64, cin, ncls, fml, n, hh, ww):
    _train_step_16bit_case(dtype, tol64, cin, ncls, fml, n, hh, ww)


@pytest.mark.parametrize("cin,ncls,fml,n,hh,ww", [(3, 0, True, 2, 100, 140), (1, 0, False, 1, 132, 68)])
def test_train_step_bf16_ragged_maps(cin, ncls, fml, n, hh, ww):
    """The bf16 train step (configs[2]) on maps that are ragged at every resolution (half- and quarter-resolution sizes that are
    no multiples of the 16-pixel tiles, sub-grids narrower than a tile): same gates as test_train_step_16bit."""
    _train_step_16bit_case("bfloat16", 4e-2, cin, ncls, fml, n, hh, ww)


def test_train_step_bf16_random_shape_soak_with_every_negative_mined():
    """Random-shape soak of the bf16 train step AGAINST THE ORACLE (the 16-bit backward was soaked for self-consistency only,
    because on random shapes the 16-bit logits of kernel and oracle may rank two near-tied hard negatives differently and the top-k
    choice is discontinuous).  The discontinuity needs k = min(n_pos, n_neg) < n_neg; these label maps are mostly POSITIVE (ones with
    one to three rectangular holes of < 45 % of the area), so k = n_neg and the mining term takes EVERY negative whatever their order
    (losses.py:110-116) -- the comparison is continuous again and random shapes are fair: batch 1-3, sides 32..144 in steps of 4
    (ragged at every resolution; maps down to 8 pixels wide, i.e. NARROWER than the larger dilations), grey / RGB, both padding rules,
    detection only, bf16 and fp16 (UBD_TRAIN16_SOAK_CASES, default 8).  Gate against the same-rounding fp32-evaluated oracle: 2e-2 per
    tensor here (5e-3 on the fixed shapes): on maps of a few hundred pixels the nine-value depthwise gradient of a grey L1 moves by 1e-2
    with single 16-bit ulp flips further up (case 15: 3 x 60 x 80, l1.dw 1.0e-2) -- a defect is orders of magnitude beyond that.  Its first run found a real defect: column phases without a pixel
    (map narrower than the dilation) staged other columns' pixels in the bf16 dilated backward (bwd16.h, fixed)."""
    import soak_labels
    rng = np.random.default_rng(2026)
    for case in range(int(os.environ.get("UBD_TRAIN16_SOAK_CASES", "8"))):
        cin, fml = int(rng.choice([1, 3])), bool(rng.integers(0, 2))
        top = int(os.environ.get("UBD_TRAIN16_SOAK_MAXSIDE", "144")) // 4 + 1
        n, hh, ww = int(rng.integers(1, 4)), 4 * int(rng.integers(8, top)), 4 * int(rng.integers(8, top))
        if top > 64 and case % 4 == 3: ww = 512                       # big runs: the 128-wide maps of the BASELINE shapes (paired sub-grids at dilation 16)
        mh, mw = hh // 4, ww // 4
        ncls = 2 if case % 3 == 2 else 0                            # every third case with classes (labels 1..2 in vertical bands)
        labels = soak_labels.mostly_positive_maps(rng, n, mh, mw, ncls)
        for dtype, tol in (("bfloat16", 4e-2), ("float16", 3e-2)):
            try:
                _train_step_16bit_case(dtype, tol, cin, ncls, fml, n, hh, ww, labels=labels, seed=500 + case, tol32=2e-2, u8=(case % 4 == 1))
            except AssertionError as e:
                raise AssertionError(f"case {case} {dtype}: cin {cin} classes {ncls} fml {fml} {n} x {hh} x {ww}: {e}")


def _train_step_16bit_case(dtype, tol64, cin, ncls, fml, n, hh, ww, labels=None, seed=None, tol32=5e-3, u8=False):
    """configs[2] (bf16 train step) on small shapes: 16-bit activations, kernels and depthwise intermediates, 16-bit
    MFMA forward, fp32 accumulation / weight gradients / master weights; bf16 mode also keeps the gradient tensors
    between L3..L9 and the depthwise-output gradients of L2/L3 in bf16 (fp16 mode keeps them fp32).
    Oracle: torch autograd with the SAME storage roundings (activations and kernels straight-through, gradient tensors
    by backward hooks).  A storage rounding can flip by one 16-bit ulp between two correct evaluations that accumulate
    differently (fp32 vs fp64), and such flips compound down the layers, so there are two gates:
      * against the oracle evaluated in fp32, like the kernels: every weight-gradient tensor within 5e-3 (relative L2);
      * against the oracle evaluated in fp64: within `tol64` (the fp32-vs-fp64 oracle spread itself reaches 2e-2).
    The shapes of these tests are fixed on purpose: a random-shape soak of this comparison (round 3) tripped on small maps where the
    16-bit logits of the kernel and of the oracle rank two near-tied hard negatives differently -- the bias gradient of the head
    then still agrees to 1e-6 (the two candidates weigh the same) while every other tensor moves by 0.3-2 % (k = 318 of 828 pixels;
    tests/diag_train16_case.py prints the per-tensor figures, and the two ORACLE evaluations differ by as much on a neighbouring
    case).  The top-k choice is discontinuous; fp32 logits agree to 1e-6 and the fp32 soak (test_gpu_train.py) has no such cases."""
    from oracle import net_torch as otorch
    from ubdvss_amd import Trainer, Adam
    from ubdvss_amd.net import PreprocessingType
    cfg = NetConfig(class_names=[f"c{i}" for i in range(ncls)] if ncls else None, grey=(cin == 1), fml_compatible=fml,
                    preprocessing=PreprocessingType.MOBILENET_LIKE if u8 else PreprocessingType.NONE)
    model = Model(cfg, dtype=dtype, seed=0)
    w = onet.init_weights(90 + cin, cin, ncls, bias_scale=0.2)
    w[-2] = (w[-2] * 4).astype(np.float32)
    model.set_weights(w)
    if labels is None:
        labels = synthetic.rectangle_maps(91, n, hh // 4, ww // 4, n_classes=ncls)
    x8 = synthetic.textured_images(92 if seed is None else seed, labels, 4, cin)
    x = x8.astype(np.float32) / 127.5 - 1.0                     # what the oracle sees; u8: the device applies (x - 127.5) / 127.5 itself (net.py:217-218)
    tr = Trainer(model, Adam())
    tr.backward_on_device(torch.from_numpy(x8 if u8 else x).cuda(), torch.from_numpy(labels).cuda())
    l = tr.loss.cpu().numpy()
    g = tr.grads.cpu().numpy().astype(np.float64)
    gdt = "bfloat16" if dtype == "bfloat16" else None
    for odt, tol in ((torch.float32, tol32), (torch.float64, tol64)):
        loss_ref, _, _, grads_ref = otorch.loss_and_grads(x, labels[..., None], w, ncls > 0, fml, dtype=odt, act_dtype=dtype,
                                                          grad_dtype=gdt)
        assert abs(l[0] - loss_ref) <= tol * abs(loss_ref), (l[0], loss_ref)
        off = 0
        for (nm, _), gr in zip(onet.weight_shapes(cin, ncls), grads_ref):
            k = gr.size
            err = np.linalg.norm(g[off:off + k] - gr.reshape(-1)) / max(np.linalg.norm(gr), 1e-30)
            assert err <= tol, (nm, err, str(odt))
            off += k
    # and the optimiser step runs on the fp32 master weights
    tr.apply_gradients()
    assert torch.isfinite(model.params).all()


def test_bf16_train_uint8_input_and_loss_goes_down():
    """bf16 train step on a larger batch (several tiles per block in every backward kernel), fed with uint8 images
    (mobilenet preprocessing fused into L1 forward and backward): gradients equal those of the same step fed with the
    preprocessed fp32 images, and Adam steps on the fixed batch reduce the objective."""
    from ubdvss_amd import Trainer, Adam
    from ubdvss_amd.net import PreprocessingType
    cfg = NetConfig(grey=False, preprocessing=PreprocessingType.MOBILENET_LIKE)
    n, side = 6, 256
    labels = synthetic.rectangle_maps(7, n, side // 4, side // 4)
    img8 = synthetic.textured_images(8, labels, 4, 3).astype(np.uint8)
    y = torch.from_numpy(labels).cuda()
    ma, mb = Model(cfg, dtype="bfloat16", seed=2), Model(cfg, dtype="bfloat16", seed=2)
    ta, tb = Trainer(ma, Adam(lr=2e-3)), Trainer(mb, Adam(lr=2e-3))
    ta.backward_on_device(torch.from_numpy(img8).cuda(), y)
    tb.backward_on_device(torch.from_numpy((img8.astype(np.float32) - 127.5) / 127.5).cuda(), y)
    ga, gb = ta.grads.cpu().numpy(), tb.grads.cpu().numpy()
    assert np.linalg.norm(ga - gb) <= 1e-5 * np.linalg.norm(gb)
    first = float(ta.loss[0])
    for _ in range(25):
        ta.train_step_on_device(torch.from_numpy(img8).cuda(), y)
    assert np.isfinite(float(ta.loss[0])) and float(ta.loss[0]) < first


def test_bf16_train_full_size_batch_is_deterministic():
    """configs[2] at its full size (batch 64 of 512x512x3, bf16): every weight-gradient kernel reduces through per-block
    partial rows added in a fixed order, so two runs of the same step give bit-identical gradients; the bf16 gradients
    stay close to the fp32-activation train step of the same weights (cosine > 0.99 per layer group)."""
    from ubdvss_amd import Trainer, Adam
    cfg = NetConfig(grey=False)
    n, side = 64, 512
    labels = synthetic.rectangle_maps(17, n, side // 4, side // 4)
    x = torch.from_numpy(synthetic.textured_images(18, labels, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
    y = torch.from_numpy(labels).cuda()
    m16 = Model(cfg, dtype="bfloat16", seed=4)
    t16 = Trainer(m16, Adam())
    t16.backward_on_device(x, y)
    g_a = t16.grads.clone()
    t16.backward_on_device(x, y)
    assert torch.equal(g_a, t16.grads)
    assert torch.isfinite(g_a).all()
    m32 = Model(cfg, seed=4)
    t32 = Trainer(m32, Adam())
    t32.backward_on_device(x, y)
    a, b = g_a.double().cpu().numpy(), t32.grads.double().cpu().numpy()
    assert abs(float(t16.loss[0]) - float(t32.loss[0])) <= 2e-2 * abs(float(t32.loss[0]))
    cos = float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b)))
    assert cos > 0.99, cos


@pytest.mark.parametrize("n,hh,ww", [(2, 256, 256), (3, 72, 104)])
def test_bf16_dilated_backward_fused_equals_split(monkeypatch, n, hh, ww):
    """The bf16 dilated backward computes a layer's data gradient inside its weight-gradient kernel (same staged tiles, same
    MFMA order as the stand-alone data-gradient kernel): the gradients must be BIT-IDENTICAL to the two-kernel path
    (UBD_DILBWD=split, read when the handle is created) -- on maps with several 16 x 16 sub-grid tiles per phase and on ragged ones.
    (The fused side runs with UBD_DILBWD=pair8: sub-grids of exactly 8 columns -- dilation 8 on the 64-wide maps here -- otherwise go in pairs
    into 16-wide tiles, which adds the same weight-gradient products in another order: test_bf16_dilated_backward_paired_subgrids_... below.)"""
    from ubdvss_amd import Trainer, Adam
    cfg = NetConfig(grey=False)
    labels = synthetic.rectangle_maps(5, n, hh // 4, ww // 4)
    x = torch.from_numpy(synthetic.textured_images(6, labels, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
    y = torch.from_numpy(labels).cuda()
    grads = {}
    for mode in ("fused", "split"):
        monkeypatch.setenv("UBD_DILBWD", "split" if mode == "split" else "pair8")
        m = Model(cfg, dtype="bfloat16", seed=9)
        t = Trainer(m, Adam())
        t.backward_on_device(x, y)
        grads[mode] = t.grads.clone()
        assert torch.isfinite(grads[mode]).all()
    monkeypatch.delenv("UBD_DILBWD", raising=False)
    assert torch.equal(grads["fused"], grads["split"])


@pytest.mark.parametrize("n,hh,ww,ncls", [(2, 256, 256, 8), (3, 72, 104, 3), (1, 40, 36, 1), (2, 64, 96, 31)])
def test_bf16_head_backward_with_classes_in_one_pass_equals_the_two_kernels(monkeypatch, n, hh, ww, ncls):
    """bf16 train step with classes: the head's data gradient G9 is written by the head's weight-gradient kernel from the tile it stages anyway
    (backward.hip head_wgrad_kernel<TX, true>) instead of by a second pass over A9 (UBD_HEADBWD=split).  Same expression, same order: loss and
    every gradient bit-identical, up to the 31 classes the ABI allows, on ragged maps too."""
    from ubdvss_amd import Trainer, Adam
    cfg = NetConfig(class_names=[f"c{i}" for i in range(ncls)], grey=False)
    labels = synthetic.rectangle_maps(55, n, hh // 4, ww // 4, n_classes=ncls)
    x = torch.from_numpy(synthetic.textured_images(56, labels, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
    y = torch.from_numpy(labels).cuda()
    grads, loss = {}, {}
    for mode in ("one", "split"):
        if mode == "split":
            monkeypatch.setenv("UBD_HEADBWD", "split")
        else:
            monkeypatch.delenv("UBD_HEADBWD", raising=False)
        t = Trainer(Model(cfg, dtype="bfloat16", seed=9), Adam())
        t.backward_on_device(x, y)
        grads[mode], loss[mode] = t.grads.clone(), t.loss.clone()
        assert torch.isfinite(grads[mode]).all() and float(grads[mode].abs().max()) > 0
    monkeypatch.delenv("UBD_HEADBWD", raising=False)
    assert torch.equal(loss["one"], loss["split"])
    assert torch.equal(grads["one"], grads["split"]), float((grads["one"] - grads["split"]).abs().max())


@pytest.mark.parametrize("n,hh,ww,layers", [(2, 512, 512, ("l8",)), (1, 64, 512, ("l8",)), (3, 260, 512, ("l8",)), (1, 36, 512, ("l8",)), (2, 256, 256, ("l7",)),
                                               (2, 128, 128, ("l6",))])
def test_bf16_dilated_backward_paired_subgrids_equal_the_8_wide_tiles(monkeypatch, n, hh, ww, layers):
    """Dilation 16 on 128-wide maps (dilation 8 on 64-wide, 4 on 32-wide ones): sub-grids of exactly 8 columns.  The fused backward kernel takes TWO of them (phases rx, rx + 1) side by
    side in one 16-wide tile with a zero column between them (bwd16.h PAIR) instead of one per 8-wide item (UBD_DILBWD=pair8).  The data
    gradient is the same arithmetic per pixel: every tensor that flows on is bit-identical, so the gradients of all OTHER layers are too up
    to their own sums; the layer's own weight gradient adds the same products in a different order (other items, other blocks): 1e-5.
    Also against the two-kernel path (UBD_DILBWD=split).  Full, single-row (64 x 512: one sub-grid row), ragged (260: 5 or 4 rows per
    sub-grid) and short (36 x 512) maps."""
    from ubdvss_amd import Trainer, Adam
    cfg = NetConfig(grey=False)
    labels = synthetic.rectangle_maps(15, n, hh // 4, ww // 4)
    x = torch.from_numpy(synthetic.textured_images(16, labels, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
    y = torch.from_numpy(labels).cuda()
    grads = {}
    for mode in ("pair", "pair8", "split"):
        if mode == "pair":
            monkeypatch.delenv("UBD_DILBWD", raising=False)
        else:
            monkeypatch.setenv("UBD_DILBWD", mode)
        t = Trainer(Model(cfg, dtype="bfloat16", seed=9), Adam())
        t.backward_on_device(x, y)
        grads[mode] = t.grads.clone()
        assert torch.isfinite(grads[mode]).all() and float(grads[mode].abs().max()) > 0
        if mode == "pair":
            t.backward_on_device(x, y)
            assert torch.equal(t.grads, grads[mode])                    # repeats bit for bit
    monkeypatch.delenv("UBD_DILBWD", raising=False)
    assert torch.equal(grads["pair8"], grads["split"])
    a, b = grads["pair"].double(), grads["pair8"].double()
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= 1e-5 * scale, float((a - b).abs().max()) / scale
    # every layer but the dilation-16 one (and L3 .. L1 below it, which only see the bit-identical data gradient): identical
    off = 0
    for nm, shape in onet.weight_shapes(3, 0):
        sz = int(np.prod(shape))
        if not nm.startswith(layers):
            assert torch.equal(grads["pair"][off:off + sz], grads["pair8"][off:off + sz]), nm
        off += sz
    assert off == grads["pair"].numel()


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("n,hh,ww", [(2, 256, 256), (3, 72, 104), (1, 512, 384)])
def test_forward16_staged_dilated_kernel_equals_direct(monkeypatch, dtype, n, hh, ww):
    """The 16-bit forward dilated layers run on tiles of the dilation sub-grids staged in LDS (hardware zero fill = padding);
    same MFMA order and epilogue as the direct kernel (UBD_DILCONV16=direct, read when the handle is created): bit-identical
    logits, on maps with several tiles per phase, ragged ones and non-square ones."""
    cfg = NetConfig(grey=False)
    x = torch.from_numpy(synthetic.noise_images(3, n, hh, ww, 3)).cuda()
    out = {}
    for mode in ("staged", "direct"):
        if mode == "direct":
            monkeypatch.setenv("UBD_DILCONV16", "direct")
        else:
            monkeypatch.delenv("UBD_DILCONV16", raising=False)
        m = Model(cfg, dtype=dtype, seed=11)
        out[mode] = m.predict_on_device(x).clone()
        assert torch.isfinite(out[mode]).all()
    assert torch.equal(out["staged"], out["direct"])


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("cin,fml,u8,n,hh,ww", [(3, True, False, 2, 256, 256), (3, False, False, 3, 72, 104), (1, True, True, 2, 136, 200),
                                                (3, True, True, 1, 512, 384), (1, False, False, 2, 64, 36)])
def test_fused_stem16_equals_split(monkeypatch, dtype, cin, fml, u8, n, hh, ww):
    """The stem of the 16-bit pass in ONE kernel (default: L1 -> L2 -> L3, sep123_16.h; UBD_STEM16=fused12: L1 -> L2 fused + L3
    alone) against the three-kernel pass (UBD_STEM16=split, read when the handle is created): the fused kernels compute their
    inputs' patches in LDS operation for operation as the separate kernels compute them: bit-identical logits -- fml and 'same'
    padding, grey and RGB, float and uint8 + preprocessing input, ragged maps (tiles cut by both map borders)."""
    from ubdvss_amd.net import PreprocessingType
    cfg = NetConfig(grey=(cin == 1), fml_compatible=fml,
                    preprocessing=PreprocessingType.MOBILENET_LIKE if u8 else PreprocessingType.NONE)
    if u8:
        x = torch.from_numpy(np.random.default_rng(5).integers(0, 256, (n, hh, ww, cin), dtype=np.uint8)).cuda()
    else:
        x = torch.from_numpy(synthetic.noise_images(3, n, hh, ww, cin)).cuda()
    out = {}
    for mode in ("fused", "fused12", "split"):
        if mode != "fused":
            monkeypatch.setenv("UBD_STEM16", mode)
        else:
            monkeypatch.delenv("UBD_STEM16", raising=False)
        m = Model(cfg, dtype=dtype, seed=11)
        out[mode] = m.predict_on_device(x).clone()
        assert torch.isfinite(out[mode]).all()
    assert torch.equal(out["fused12"], out["split"])
    assert torch.equal(out["fused"], out["split"]), float((out["fused"] - out["split"]).abs().max())


@pytest.mark.parametrize("n,hh,ww", [(2, 256, 256), (3, 72, 104)])
def test_bf16_train_fused_stem_equals_split(monkeypatch, n, hh, ww):
    """Train step: the fused L1 -> L2 kernel also stores L1's activation (the backward pass reads it), every pixel once by the tile
    that owns it: loss and gradients bit-identical to the two-kernel forward."""
    from ubdvss_amd import Trainer, Adam
    cfg = NetConfig(grey=False)
    labels = synthetic.rectangle_maps(5, n, hh // 4, ww // 4)
    x = torch.from_numpy(synthetic.textured_images(6, labels, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
    y = torch.from_numpy(labels).cuda()
    grads, loss = {}, {}
    for mode in ("fused", "fused12", "split"):
        if mode != "fused":
            monkeypatch.setenv("UBD_STEM16", mode)
        else:
            monkeypatch.delenv("UBD_STEM16", raising=False)
        m = Model(cfg, dtype="bfloat16", seed=9)
        t = Trainer(m, Adam())
        t.backward_on_device(x, y)
        grads[mode], loss[mode] = t.grads.clone(), t.loss.clone()
        assert torch.isfinite(grads[mode]).all()
    for mode in ("fused", "fused12"):               # a1 and a2 are stored by the fused kernels too: every pixel once, same bits
        assert torch.equal(loss[mode], loss["split"]), mode
        assert torch.equal(grads[mode], grads["split"]), (mode, float((grads[mode] - grads["split"]).abs().max()))


@pytest.mark.parametrize("n,hh,ww,ncls", [(2, 256, 256, 0), (3, 72, 104, 0), (2, 64, 96, 3), (1, 40, 36, 0)])
def test_bf16_chained_partial_sum_reduction_equals_the_batched_launches(monkeypatch, n, hh, ww, ncls):
    """bf16 train step: every weight-gradient kernel totals, at its end, the per-block partial rows of the producer in front of it
    (backward.hip rp_reduce_tail; the last producer's rows go to one small stand-alone launch) instead of two batched reduction
    launches per pass (UBD_REDUCE=batched).  Same rows, same order of additions: loss and gradients are bit-identical."""
    from ubdvss_amd import Trainer, Adam
    cfg = NetConfig(class_names=[f"c{i}" for i in range(ncls)] if ncls else None, grey=False)
    labels = synthetic.rectangle_maps(25, n, hh // 4, ww // 4, n_classes=ncls)
    x = torch.from_numpy(synthetic.textured_images(26, labels, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
    y = torch.from_numpy(labels).cuda()
    grads, loss = {}, {}
    for mode in ("chained", "batched"):
        if mode == "batched":
            monkeypatch.setenv("UBD_REDUCE", "batched")
        else:
            monkeypatch.delenv("UBD_REDUCE", raising=False)
        t = Trainer(Model(cfg, dtype="bfloat16", seed=9), Adam())
        t.backward_on_device(x, y)
        grads[mode], loss[mode] = t.grads.clone(), t.loss.clone()
        assert torch.isfinite(grads[mode]).all() and float(grads[mode].abs().max()) > 0
        t.backward_on_device(x, y)
        assert torch.equal(t.grads, grads[mode])                 # every gradient repeats bit for bit (with classes too: the head's block sums have a fixed order since round 4)
    assert torch.equal(loss["chained"], loss["batched"])
    assert torch.equal(grads["chained"], grads["batched"])


@pytest.mark.parametrize("grey", [False, True])
@pytest.mark.parametrize("n,hh,ww", [(2, 256, 256), (3, 72, 104), (1, 40, 36), (2, 132, 68)])
def test_bf16_l1_backward_input_patch_by_lds_dma_equals_register_staging(monkeypatch, grey, n, hh, ww):
    """bf16 train step on preprocessed fp32 images: the backward kernel of L1 fetches its fp32 input patch by LDS-DMA (16-byte chunks, rows
    from the aligned boundary below the patch, zeros outside the image from the buffer descriptor; sepbwd16.h IN_MODE 2) instead of
    4-byte loads held in registers over phase 2 (UBD_SEPB16_X=regs).  Same values, same order of sums: bit-identical gradients."""
    from ubdvss_amd import Trainer, Adam
    cfg = NetConfig(grey=grey)
    cin = 1 if grey else 3
    labels = synthetic.rectangle_maps(45, n, hh // 4, ww // 4)
    x = torch.from_numpy(synthetic.textured_images(46, labels, 4, cin).astype(np.float32) / 127.5 - 1.0).cuda()
    y = torch.from_numpy(labels).cuda()
    grads, loss = {}, {}
    for mode in ("dma", "regs"):
        if mode == "regs":
            monkeypatch.setenv("UBD_SEPB16_X", "regs")
        else:
            monkeypatch.delenv("UBD_SEPB16_X", raising=False)
        t = Trainer(Model(cfg, dtype="bfloat16", seed=9), Adam())
        t.backward_on_device(x, y)
        grads[mode], loss[mode] = t.grads.clone(), t.loss.clone()
        assert torch.isfinite(grads[mode]).all() and float(grads[mode].abs().max()) > 0
    assert torch.equal(loss["dma"], loss["regs"])
    assert torch.equal(grads["dma"], grads["regs"]), float((grads["dma"] - grads["regs"]).abs().max())


@pytest.mark.parametrize("grey", [False, True])
def test_bf16_train_uint8_input_fused_stem_equals_split(monkeypatch, grey):
    """The train step fed uint8 pixels (preprocessing fused into L1; the register-staged input path of the one-kernel stem, which also stores a1
    and a2): loss and gradients bit-identical to the three-kernel forward, RGB and grey."""
    from ubdvss_amd import Trainer, Adam, PreprocessingType
    cfg = NetConfig(grey=grey, preprocessing=PreprocessingType.MOBILENET_LIKE)
    cin = 1 if grey else 3
    labels = synthetic.rectangle_maps(35, 3, 72 // 4, 136 // 4)
    x = torch.from_numpy(synthetic.textured_images(36, labels, 4, cin)).cuda()
    assert x.dtype == torch.uint8
    y = torch.from_numpy(labels).cuda()
    grads, loss = {}, {}
    for mode in ("fused", "fused12", "split"):
        if mode != "fused":
            monkeypatch.setenv("UBD_STEM16", mode)
        else:
            monkeypatch.delenv("UBD_STEM16", raising=False)
        t = Trainer(Model(cfg, dtype="bfloat16", seed=9), Adam())
        t.backward_on_device(x, y)
        grads[mode], loss[mode] = t.grads.clone(), t.loss.clone()
        assert torch.isfinite(grads[mode]).all() and float(grads[mode].abs().max()) > 0
    for mode in ("fused", "fused12"):
        assert torch.equal(loss[mode], loss["split"]), mode
        assert torch.equal(grads[mode], grads["split"]), (mode, float((grads[mode] - grads["split"]).abs().max()))


@pytest.mark.parametrize("n,hh,ww", [(2, 256, 256), (3, 72, 104)])
def test_bf16_train_staged_last_layer_equals_direct(monkeypatch, n, hh, ww):
    """Round 4: the LDS-staged kernel also runs the LAST dilated layer, with the 1 x 1 head in its epilogue (inference: logits only;
    train step: logits + the stored activation).  UBD_DILCONV16=direct keeps the direct kernel for every layer: loss and gradients of the
    bf16 train step are bit-identical (the forward equality is test_forward16_staged_dilated_kernel_equals_direct)."""
    from ubdvss_amd import Trainer, Adam
    cfg = NetConfig(grey=False)
    labels = synthetic.rectangle_maps(45, n, hh // 4, ww // 4)
    x = torch.from_numpy(synthetic.textured_images(46, labels, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
    y = torch.from_numpy(labels).cuda()
    grads, loss = {}, {}
    for mode in ("staged", "direct"):
        if mode == "direct":
            monkeypatch.setenv("UBD_DILCONV16", "direct")
        else:
            monkeypatch.delenv("UBD_DILCONV16", raising=False)
        t = Trainer(Model(cfg, dtype="bfloat16", seed=9), Adam())
        t.backward_on_device(x, y)
        grads[mode], loss[mode] = t.grads.clone(), t.loss.clone()
    assert torch.equal(loss["staged"], loss["direct"])
    assert torch.equal(grads["staged"], grads["direct"])
