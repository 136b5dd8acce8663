"""GPU parity: HIP forward pass (through the C ABI) vs the CPU oracle and the golden vectors.
Tolerance (fp32 path): max |logit_gpu - logit_oracle| <= 1e-3 absolute as north_star states; the
test additionally enforces a much tighter 2e-5 * max|logit| + 1e-6 bound, which an exact-fp32 MFMA
pipeline meets.  Binary maps must be identical on every pixel whose oracle logit is farther than
1e-3 from the threshold (in-margin count reported)."""
import os

import numpy as np
import pytest
import torch

from oracle import net_numpy as onet, net_torch as otorch
from ubdvss_amd import NetConfig, Model, synthetic

pytestmark = pytest.mark.gpu


def _model(cin, ncls, fml, weights):
    cfg = NetConfig(class_names=[f"c{i}" for i in range(ncls)] if ncls else None, grey=(cin == 1), fml_compatible=fml)
    m = Model(cfg)
    m.set_weights(weights)
    return m


def _check(lg_gpu, lg_ref, thr=-0.0):
    err = np.abs(lg_gpu.astype(np.float64) - lg_ref).max()
    tol = 2e-5 * np.abs(lg_ref).max() + 1e-6
    assert err <= 1e-3, err
    assert err <= tol, (err, tol)
    far = np.abs(lg_ref[..., 0] - thr) > 1e-3
    assert np.array_equal((lg_gpu[..., 0] > thr)[far], (lg_ref[..., 0] > thr)[far])
    return int((~far).sum())


def test_golden_cfg1(golden_dir):
    x = synthetic.noise_images(0, 1, 256, 256, 3)
    w = onet.init_weights(1, 3, 0)
    m = _model(3, 0, True, w)
    lg = m.predict(x)
    gold = np.load(os.path.join(golden_dir, "cfg1_logits.npy"))
    assert lg.shape == gold.shape == (1, 64, 64, 1) and lg.dtype == np.float32
    _check(lg, gold.astype(np.float64))


def test_golden_net_cases(golden_dir, manifest):
    for name in manifest["net_cases"]:
        d = np.load(os.path.join(golden_dir, f"net_{name}.npz"))
        cin, ncls, fml = int(d["c_in"]), int(d["n_classes"]), bool(d["fml"])
        m = _model(cin, ncls, fml, onet.unflatten_weights(d["params"], cin, ncls))
        _check(m.predict(d["x"]), d["logits"].astype(np.float64))


@pytest.mark.parametrize("cin,ncls,fml,n,hh,ww", [
    (3, 0, True, 3, 128, 192), (1, 5, True, 2, 64, 64), (3, 1, False, 1, 192, 64),
    (1, 0, True, 1, 512, 512), (3, 0, True, 2, 72, 100)])       # last: sizes that are only multiples of 4
def test_vs_numpy_oracle(cin, ncls, fml, n, hh, ww):
    w = onet.init_weights(100 + cin + ncls, cin, ncls, bias_scale=0.25)
    x = synthetic.noise_images(7, n, hh, ww, cin)
    ref = onet.forward(x.astype(np.float64), w, fml)
    lg = _model(cin, ncls, fml, w).predict(x)
    _check(lg, ref)


def test_random_shape_soak_vs_oracle(monkeypatch):
    """Random batch sizes and image sides (multiples of 4 from 16 to 160: tiles cut by both borders, maps narrower than a tile, dilation
    sub-grids of one pixel), grey / RGB, with / without classes, both padding rules, float and uint8 input, every stem variant (the
    one-kernel stem is forced on half of the cases: UBD_STEM is read when the handle is created).  UBD_FWD_SOAK_CASES scales it."""
    from ubdvss_amd.net import PreprocessingType
    rng = np.random.default_rng(77)
    for case in range(int(os.environ.get("UBD_FWD_SOAK_CASES", "16"))):
        cin, ncls, fml = int(rng.choice([1, 3])), int(rng.choice([0, 0, 2])), bool(rng.integers(0, 2))
        n, hh, ww = int(rng.integers(1, 4)), 4 * int(rng.integers(4, 41)), 4 * int(rng.integers(4, 41))
        u8 = bool(rng.integers(0, 2))
        stem = str(rng.choice(["fused123", "fused", "unfused", "cold123", ""]))
        if stem: monkeypatch.setenv("UBD_STEM", stem)
        else: monkeypatch.delenv("UBD_STEM", raising=False)
        w = onet.init_weights(300 + case, cin, ncls, bias_scale=0.25)
        cfg = NetConfig(class_names=[f"c{i}" for i in range(ncls)] if ncls else None, grey=(cin == 1), fml_compatible=fml,
                        preprocessing=PreprocessingType.MOBILENET_LIKE if u8 else PreprocessingType.NONE)
        m = Model(cfg)
        m.set_weights(w)
        if u8:
            x8 = rng.integers(0, 256, (n, hh, ww, cin), dtype=np.uint8)
            ref = onet.forward((x8.astype(np.float64) - 127.5) / 127.5, w, fml)
            lg = m.predict(x8)
        else:
            x = synthetic.noise_images(case, n, hh, ww, cin)
            ref = onet.forward(x.astype(np.float64), w, fml)
            lg = m.predict(x)
        try:
            _check(lg, ref)
        except AssertionError as e:
            raise AssertionError(f"case {case}: cin {cin} classes {ncls} fml {fml} {n} x {hh} x {ww} uint8 {u8} UBD_STEM={stem!r}: {e}")


def test_identity_kernels_kat():
    """centre-tap identity kernels in L4..L9 leave the (non-negative) L3 output unchanged, so the
    logits equal head(L3 output): isolates halo / dilation addressing."""
    w = onet.init_weights(5, 3, 0, bias_scale=0.1)
    ident = np.zeros((3, 3, 24, 24), np.float32); ident[1, 1] = np.eye(24)
    for i in range(9, 21, 2):
        w[i] = ident.copy(); w[i + 1] = np.zeros(24, np.float32)
    x = synthetic.noise_images(9, 1, 128, 128, 3)
    _, acts = onet.forward(x.astype(np.float64), w, return_all=True)
    ref = acts[2] @ w[21][0, 0].astype(np.float64) + w[22]
    _check(_model(3, 0, True, w).predict(x), ref)


def test_uint8_input_with_fused_preprocessing():
    from ubdvss_amd import PreprocessingType
    w = onet.init_weights(6, 3, 0, bias_scale=0.1)
    cfg = NetConfig(grey=False, preprocessing=PreprocessingType.MOBILENET_LIKE)
    m = Model(cfg); m.set_weights(w)
    xu8 = synthetic.noise_images(11, 2, 64, 64, 3, as_float=False)
    lg = m.predict_on_device(torch.from_numpy(xu8).cuda()).cpu().numpy()
    ref = onet.forward(onet.preprocess_mobilenet(xu8.astype(np.float64)), w)
    _check(lg, ref)


def test_full_size_batch_vs_torch_cpu():
    """BASELINE.json configs[1] shape at a reduced batch (oracle time): 4 x 512x512x3 fp32."""
    w = onet.init_weights(1, 3, 0)
    x = synthetic.noise_images(2, 4, 512, 512, 3)
    ref = otorch.forward_numpy(x, w, True, torch.float64)
    _check(_model(3, 0, True, w).predict(x), ref)


def test_linearity_property_full_size():
    """Size-independent property at the full bench size (32 x 512x512x3): with zero biases the net is
    positively homogeneous, f(2x) = 2 f(x) exactly in binary floating point (scaling by 2 commutes
    with every rounding), and the batch is processed image-independently."""
    w = onet.init_weights(1, 3, 0)                      # zero biases
    m = _model(3, 0, True, w)
    x = torch.from_numpy(synthetic.noise_images(2, 32, 512, 512, 3)).cuda()
    a = m.predict_on_device(x).clone()
    b = m.predict_on_device(2 * x).clone()
    assert torch.equal(2 * a, b)
    c = m.predict_on_device(x[5:6].contiguous())
    assert torch.equal(c[0], a[5])


def test_error_paths():
    m = _model(3, 0, True, onet.init_weights(1, 3, 0))
    with pytest.raises(ValueError):
        m.predict(np.zeros((1, 30, 32, 3), np.float32))
    with pytest.raises(ValueError):
        m.predict(np.zeros((1, 32, 32, 1), np.float32))
    with pytest.raises(ValueError):
        m.set_weights(onet.init_weights(1, 1, 0))


def test_packed_fragments_follow_the_parameters():
    """Model.predict_on_device reuses the packed weight fragments of its workspace (UBD_IN_PREPACKED) until the
    parameters change: in-place torch writes, set_weights and the trainer's Adam step must all invalidate them."""
    from ubdvss_amd import Trainer, Adam
    cfg = NetConfig(grey=False)
    m = Model(cfg, seed=3)
    x = torch.from_numpy(synthetic.noise_images(5, 2, 64, 64, 3)).cuda()
    a = m.predict_on_device(x).clone()
    b = m.predict_on_device(x).clone()                      # second call: prepacked path
    assert torch.equal(a, b)
    m.params.mul_(1.5)                                       # in-place torch write
    ref = Model(cfg, seed=3); ref.params.copy_(m.params)
    assert torch.equal(m.predict_on_device(x), ref.predict_on_device(x))
    w = m.get_weights(); w[0] = w[0] * 0.5
    m.set_weights(w); ref.set_weights(w)
    assert torch.equal(m.predict_on_device(x), ref.predict_on_device(x))
    tr = Trainer(m, Adam())
    y = torch.from_numpy(synthetic.rectangle_maps(6, 2, 16, 16)).cuda()
    tr.train_step_on_device(x, y)                            # parameters move through the C-ABI
    ref2 = Model(cfg, seed=3); ref2.params.copy_(m.params)
    assert torch.equal(m.predict_on_device(x), ref2.predict_on_device(x))


def test_load_keras_weights_export(tmp_path):
    """The migration path for trained reference models: np.savez(path, *keras_model.get_weights())."""
    cfg = NetConfig(class_names=["a", "b"], grey=True)
    w = onet.init_weights(11, 1, 2, bias_scale=0.1)
    path = str(tmp_path / "keras_weights.npz")
    np.savez(path, *w)
    m = Model(cfg, seed=0)
    m.load_keras_weights(path)
    for a, b in zip(m.get_weights(), w):
        assert np.array_equal(a, b)
    with pytest.raises(ValueError):
        Model(NetConfig(grey=False), seed=0).load_keras_weights(path)      # wrong architecture


def test_direct_dilated_kernel_path(monkeypatch):
    """UBD_DILCONV=direct selects the implicit-GEMM dilated kernel (and the unfused head) instead of the Winograd one;
    read when the handle is created.  Same oracle, same bounds."""
    monkeypatch.setenv("UBD_DILCONV", "direct")
    for cin, ncls, fml, n, hh, ww in ((3, 0, True, 2, 128, 128), (1, 2, False, 1, 72, 100)):
        w = onet.init_weights(300 + cin + ncls, cin, ncls, bias_scale=0.25)
        x = synthetic.noise_images(9, n, hh, ww, cin)
        ref = onet.forward(x.astype(np.float64), w, fml)
        _check(_model(cin, ncls, fml, w).predict(x), ref)


def test_fp32_mfma_winograd_path_and_split_product_path(monkeypatch):
    """UBD_DILCONV=wino32 keeps the round-2 form of the Winograd layer (products on v_mfma_f32_16x16x4_f32, wino.hip); the default since
    round 6 computes every fp32 product of the transform-domain GEMMs as an exact three-way bf16 split on v_mfma_f32_16x16x32_bf16
    (wino6.hip).  Both against the oracle with the SAME bounds, on shapes whose tile runs are ragged for every dilation (maps 18 x 25 and
    33 x 17: sub-grids of one pixel at dilation 16, runs of 1 / 2 / 4 / 8 tiles), with and without the head in the last layer's epilogue;
    and against each other: the two forms differ by rounding only (a few ulp of the layer's largest value per layer)."""
    res = {}
    for mode in ("wino32", ""):
        if mode: monkeypatch.setenv("UBD_DILCONV", mode)
        else: monkeypatch.delenv("UBD_DILCONV", raising=False)
        for cin, ncls, fml, n, hh, ww in ((3, 0, True, 2, 72, 100), (1, 2, False, 1, 132, 68), (3, 0, False, 3, 64, 256)):
            w = onet.init_weights(400 + cin + ncls, cin, ncls, bias_scale=0.25)
            x = synthetic.noise_images(13, n, hh, ww, cin)
            ref = onet.forward(x.astype(np.float64), w, fml)
            lg = _model(cin, ncls, fml, w).predict(x)
            _check(lg, ref)
            res.setdefault((cin, ncls, hh), []).append((lg, ref))
    for (a, ref), (b, _) in res.values():
        assert np.abs(a.astype(np.float64) - b).max() <= 4e-6 * np.abs(ref).max() + 1e-7


def test_split_product_layer_is_homogeneous_and_deterministic():
    """The three-way split commutes with powers of two (every piece scales exactly): with zero biases the layer satisfies
    f(2x) = 2 f(x) and f(x / 4) = f(x) / 4 BIT for bit; and two launches on the same input give the same bits (no atomics, fixed
    order of the six products).  One dilated layer through ubd_dilated_layer, every dilation, 5 x 72 x 100 x 24."""
    import ctypes
    from ubdvss_amd import _lib
    lib = _lib.load()
    m = Model(NetConfig(grey=False), seed=21)
    w = m.get_weights()
    for i in range(10, 22, 2):
        w[i] = np.zeros(24, np.float32)                      # biases of L4..L9
    m.set_weights(w)
    ws = torch.empty(int(lib.ubd_forward_workspace_bytes(m._h, 1, 4, 4)), dtype=torch.uint8, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.ubd_pack_weights(m._h, m.params.data_ptr(), ws.data_ptr(), ws.numel(), st), "pack")
    x = torch.rand((5, 72, 100, 24), device="cuda") - 0.3

    def layer(k, inp):
        y = torch.empty_like(inp)
        _lib.check(lib.ubd_dilated_layer(m._h, m.params.data_ptr(), k, inp.data_ptr(), y.data_ptr(), 5, 72, 100, ws.data_ptr(), st), "dil")
        return y
    for k in range(6):
        y1 = layer(k, x)
        assert torch.equal(layer(k, x), y1)
        assert torch.equal(layer(k, 2.0 * x), 2.0 * y1)
        assert torch.equal(layer(k, 0.25 * x), 0.25 * y1)
        assert float(y1.abs().max()) > 0.1


def test_broadcast_invalidates_packed_fragments(monkeypatch):
    """c10d collectives write ``model.params`` without bumping its version counter: a rank that predicted
    before ``Trainer.broadcast_weights`` must not keep its old packed fragments.  The broadcast is emulated by a
    version-preserving write (``.data.copy_``), which is what ``dist.broadcast`` does to a non-src rank's tensor."""
    from ubdvss_amd import Trainer, distributed
    cfg = NetConfig(grey=False)
    m = Model(cfg, seed=3)
    x = torch.from_numpy(synthetic.noise_images(5, 2, 64, 64, 3)).cuda()
    m.predict_on_device(x)                                   # packs the fragments of the seed-3 weights
    src = Model(cfg, seed=4)

    def fake_broadcast(flat_params, src=0, group=None):
        v = flat_params._version
        flat_params.data.copy_(src_params)
        assert flat_params._version == v                    # like c10d: no version bump
    src_params = src.params
    monkeypatch.setattr(distributed, "broadcast_parameters", fake_broadcast)
    Trainer(m).broadcast_weights()
    assert torch.equal(m.predict_on_device(x), src.predict_on_device(x))


def test_failed_forward_does_not_poison_the_prepacked_key():
    """A rejected ubd_forward call (batch over the 2^30-byte L3 limit) must not leave a key that makes the next call
    skip weight packing on a fresh workspace."""
    cfg = NetConfig(grey=False)
    m = Model(cfg, seed=3)
    ref = Model(cfg, seed=3)
    x = torch.from_numpy(synthetic.noise_images(5, 2, 64, 64, 3)).cuda()
    with pytest.raises(ValueError):
        m.predict_on_device(torch.zeros((2, 64, 64, 1), dtype=torch.float32, device="cuda"))   # wrong channel count
    with pytest.raises(RuntimeError, match="batch too large"):
        m.predict_on_device(torch.zeros((1, 13392, 13392, 3), dtype=torch.uint8, device="cuda"))  # quarter-res activation > 2^30 B
    assert m._packed_key is None                                  # the rejected call packed nothing
    assert torch.equal(m.predict_on_device(x), ref.predict_on_device(x))


def test_uint8_rule_is_the_same_at_every_entry_point():
    """uint8 images are raw pixels and take NetConfig's preprocessing on device, float images are fed as they are --
    Model.predict, ModelRunner.predict, Trainer.train_on_batch and the *_on_device calls agree."""
    from ubdvss_amd import PreprocessingType, ModelRunner, Trainer
    w = onet.init_weights(6, 3, 0, bias_scale=0.1)
    cfg = NetConfig(grey=False, preprocessing=PreprocessingType.MOBILENET_LIKE)
    m = Model(cfg); m.set_weights(w)
    xu8 = synthetic.noise_images(11, 2, 64, 64, 3, as_float=False)
    xf = onet.preprocess_mobilenet(xu8.astype(np.float64)).astype(np.float32)
    a = m.predict_on_device(torch.from_numpy(xu8).cuda()).cpu().numpy()
    assert np.array_equal(m.predict(xu8), a)
    _check(m.predict(xf), onet.forward(xf.astype(np.float64), w))
    _check(a, onet.forward(xf.astype(np.float64), w))
    det_u8 = ModelRunner(cfg).predict(m, xu8)[0]
    det_f = ModelRunner(cfg).predict(m, xf)[0]
    assert np.array_equal(det_u8, det_f)
    labels = synthetic.rectangle_maps(6, 2, 16, 16)
    m2 = Model(cfg); m2.set_weights(w)
    l_u8 = Trainer(m).train_on_batch(xu8, labels[..., None])
    l_f = Trainer(m2).train_on_batch(xf, labels[..., None])
    assert abs(l_u8 - l_f) <= 1e-5 * abs(l_f)


def test_net_manager_loads_reference_files(tmp_path, golden_dir):
    """A log dir as the reference leaves it -- Keras ``inference_model.h5`` + pickled NetConfig -- loads through
    NetManager(log_dir).load_model() (net.py:443-466) and predicts like the oracle on those weights."""
    import shutil
    from ubdvss_amd import NetManager, keras_h5
    from ubdvss_amd.net import weight_shapes
    shutil.copy(os.path.join(golden_dir, "keras_model_rgb.h5"), tmp_path / "inference_model.h5")
    mgr = NetManager(str(tmp_path), NetConfig(grey=False))
    mgr.save_config()
    mgr2 = NetManager(str(tmp_path))
    mgr2.load_model()
    w, _ = keras_h5.read_keras_weights(os.path.join(golden_dir, "keras_model_rgb.h5"))
    assert [a.shape for a in w] == [tuple(s) for s in weight_shapes(3, 0)]
    x = synthetic.noise_images(9, 1, 64, 64, 3)
    ref = onet.forward(x.astype(np.float64), w)                  # fixture weights are U(-1,1) noise: logits are huge
    got = mgr2.get_keras_model().predict(x)
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
    mgr2.save_model(7)                                        # net.py:418-420: numbered snapshot + current model
    assert (tmp_path / "model007.h5").exists() and (tmp_path / "model.h5").exists()       # Keras files, like the reference's
    w7, names7 = keras_h5.read_keras_weights(str(tmp_path / "model007.h5"))
    assert all(np.array_equal(a, b) for a, b in zip(w7, w)) and names7[0] == "separable_conv2d_1/depthwise_kernel:0"
    mgr2.save_inference()                                     # net.py:422-427: model_weights.h5 + inference_model.h5
    assert (tmp_path / "model_weights.h5").exists()
    mgr4 = NetManager(str(tmp_path)); mgr4.load_model()       # picks the inference model this package just wrote
    assert torch.equal(mgr4.get_keras_model().params, mgr2.get_keras_model().params)
    other = tmp_path / "other"; other.mkdir()
    mgr3 = NetManager(str(other), NetConfig(grey=False, max_image_side=1024))
    cfg3 = mgr3.load_another_model(str(tmp_path))
    assert cfg3.get_max_side() == 1024 and torch.equal(mgr3.get_keras_model().params, mgr2.get_keras_model().params)
    with pytest.raises(AssertionError):
        mgr2.load_model("somewhere/else.h5")


def test_fused_stem_path(monkeypatch):
    """Inference stem variants (UBD_STEM): "fused123" = L1 -> L2 -> L3 in ONE kernel, neither intermediate activation ever in
    memory (stem123.h; the default with the fml padding), "fused" = L1, then L2 -> L3 fused (stem23.h), "unfused" = three
    kernels (what training runs), "cold123" = the one kernel with single cold-started tiles as work units.  Same oracle, same bounds, all variants agree to fp32 rounding; UBD_TEST_NUM_CUS=2 makes every
    block walk several strips of tiles (register prefetch of the next input patch, the carried 33rd column across tiles and strip
    starts); uint8 input with the fused preprocessing and grey input go through the one-kernel stem too."""
    from ubdvss_amd import PreprocessingType
    cases = ((3, 0, True, 2, 128, 128), (1, 2, False, 1, 72, 100), (3, 1, True, 3, 64, 200), (3, 0, False, 1, 8, 8), (1, 0, True, 2, 4, 36),
             (3, 0, True, 2, 96, 512), (3, 0, False, 2, 40, 264), (1, 1, True, 2, 136, 72))
    outs = {}
    for mode in ("unfused", "fused", "fused123", "fused123_few_cus", "cold123", "cold123_few_cus"):
        monkeypatch.setenv("UBD_STEM", mode.replace("_few_cus", ""))
        if mode.endswith("few_cus"):
            monkeypatch.setenv("UBD_TEST_NUM_CUS", "2")
        else:
            monkeypatch.delenv("UBD_TEST_NUM_CUS", raising=False)
        for cin, ncls, fml, n, hh, ww in cases:
            w = onet.init_weights(400 + cin + ncls, cin, ncls, bias_scale=0.25)
            x = synthetic.noise_images(19, n, hh, ww, cin)
            ref = onet.forward(x.astype(np.float64), w, fml)
            mdl = _model(cin, ncls, fml, w)
            assert (mdl.num_cus == 2) == mode.endswith("few_cus")        # the override reached ubd_create
            lg = mdl.predict(x)
            _check(lg, ref)
            outs[(mode, cin, ncls, fml, n, hh, ww)] = lg
        # uint8 input, preprocessing fused into the first layer
        w = onet.init_weights(6, 3, 0, bias_scale=0.1)
        m = Model(NetConfig(grey=False, preprocessing=PreprocessingType.MOBILENET_LIKE)); m.set_weights(w)
        xu8 = synthetic.noise_images(11, 2, 72, 136, 3, as_float=False)
        _check(m.predict(xu8), onet.forward(onet.preprocess_mobilenet(xu8.astype(np.float64)), w))
    for cin, ncls, fml, n, hh, ww in cases:
        b = outs[("unfused", cin, ncls, fml, n, hh, ww)]
        for mode in ("fused", "fused123"):
            assert np.abs(outs[(mode, cin, ncls, fml, n, hh, ww)] - b).max() <= 1e-5 * max(1.0, np.abs(b).max()), (mode, cin, fml, hh, ww)
        assert np.array_equal(outs[("fused123", cin, ncls, fml, n, hh, ww)], outs[("fused123_few_cus", cin, ncls, fml, n, hh, ww)])   # tile -> block assignment is irrelevant
        # "cold123" (round 6; the default of launches too small for strips): the same kernel with one cold-started tile per work unit, the
        # tiles of a row 15 columns apart -- every output by the same instructions on the same values as in the strip walk: bit-identical
        if fml:                                                          # both are forms of the fml-padding kernel; other models take the separate kernels
            assert np.array_equal(outs[("cold123", cin, ncls, fml, n, hh, ww)], outs[("fused123", cin, ncls, fml, n, hh, ww)]), (cin, fml, hh, ww)
            assert np.array_equal(outs[("cold123_few_cus", cin, ncls, fml, n, hh, ww)], outs[("fused123", cin, ncls, fml, n, hh, ww)]), (cin, fml, hh, ww)


@pytest.mark.parametrize("few_cus", [False, True])
def test_unfused_call_then_prepacked_fused_call_on_a_poisoned_workspace(monkeypatch, few_cus):
    """a forward pass that takes the three separate stem kernels packs the weights and stores the packed key;
    the NEXT call on the same workspace is sent as UBD_IN_PREPACKED and may qualify for the one-kernel stem, whose strip-ticket /
    check-out counters live in the workspace (torch.empty: garbage).  They must have been zeroed by the call that packed,
    whichever stem variant it ran.  Real shapes (8 x 512^2 = 256 strips < 2 x 256 CUs, then 32 x 256^2 = 512 strips) and a
    two-CU handle (1 x 28 x 512 = 2 strips, then 4 x 32 x 64 = 8 strips)."""
    if few_cus:
        monkeypatch.setenv("UBD_TEST_NUM_CUS", "2")
        shapes = ((1, 28, 512), (4, 32, 64))
    else:
        shapes = ((8, 512, 512), (32, 256, 256))
    cfg = NetConfig(grey=False)
    w = onet.init_weights(77, 3, 0, bias_scale=0.2)
    m = Model(cfg); m.set_weights(w)
    lib, h = m._lib, m._h
    (na, ha, wa), (nb, hb, wb) = shapes
    nbytes = max(lib.ubd_forward_workspace_bytes(h, na, ha, wa), lib.ubd_forward_workspace_bytes(h, nb, hb, wb))
    m._ws = torch.full((int(nbytes),), 0xA5, dtype=torch.uint8, device="cuda")      # poisoned: tickets = 0xA5A5A5A5 (negative)
    xa = torch.from_numpy(synthetic.noise_images(3, na, ha, wa, 3)).cuda()
    xb = torch.from_numpy(synthetic.noise_images(4, nb, hb, wb, 3)).cuda()
    m.predict_on_device(xa)                                       # unfused stem, packs
    assert m._packed_key is not None
    got = m.predict_on_device(xb)                                 # prepacked + one-kernel stem
    fresh = Model(cfg); fresh.set_weights(w)
    assert torch.equal(got, fresh.predict_on_device(xb))
    assert torch.equal(m.predict_on_device(xb), got)              # and the counters reset themselves for the call after
    if few_cus:
        _check(got.cpu().numpy(), onet.forward(xb.cpu().numpy().astype(np.float64), w))


def test_graphed_forward_equals_the_launches_and_follows_the_weights():
    """Model.graphed_forward: the captured HIP graph of one shape is bit-identical to predict_on_device, Model.predict uses its static
    tensors for one image per call (the reference's latency protocol, predict.py:73-78), a parameter change is picked up, and another shape in between (which
    moves the workspace) does not leave a stale graph behind."""
    cfg = NetConfig(grey=True)
    m = Model(cfg, seed=3)
    x = torch.rand((1, 128, 192, 1), device="cuda")
    ref = m.predict_on_device(x).clone()
    gf = m.graphed_forward(1, 128, 192)
    for _ in range(3):
        assert torch.equal(gf(x), ref)
    assert np.array_equal(m.predict(x.cpu().numpy()), ref.cpu().numpy())
    big = torch.rand((4, 256, 256, 1), device="cuda")
    m.predict_on_device(big)                                  # a larger workspace: the graph of the small shape is re-captured
    assert torch.equal(gf(x), ref)
    w = m.get_weights()
    w[0] = w[0] * 1.5
    m.set_weights(w)
    ref2 = m.predict_on_device(x).clone()
    assert not torch.equal(ref2, ref)
    assert torch.equal(gf(x), ref2)
    x2 = torch.rand((1, 128, 192, 1), device="cuda")           # other images: copied into the graph's static input
    assert torch.equal(gf(x2), m.predict_on_device(x2))
    with pytest.raises(ValueError):
        gf(big)


def test_split_product_layer_fuzz_against_the_fp32_mfma_form():
    """tools/wino6_fuzz_layer.py: one dilated layer through ubd_dilated_layer, split-product form vs fp32-MFMA form, on random map sizes
    from 1 x 1 to 70 x 150 (any residue against the tile runs), batch sizes, every dilation, inputs scaled by 1e-3 / 1 / 100: every output
    written, difference <= 4e-6 of the layer's largest value."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "wino6_fuzz_layer.py")], env=dict(os.environ, FUZZ_CASES="150", FUZZ_SEED="7"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-600:] + r.stderr[-600:]
    assert "failures 0" in r.stdout


def test_zero_kernel_layer_writes_exactly_the_bias_at_full_size():
    """Regression test of two silent hazards found while building the split-product layer (profiles/r06_experiment_notes.txt A.6): with
    an all-zero 3x3 kernel every output of a dilated layer is relu(bias) EXACTLY, whatever the input -- any stale register (a vector
    add reading an MFMA result too early) or store-data register overwritten before the store has read it shows up as a wrong value.
    The second hazard hit ~0.1 % of the pixels of a full-size launch, different ones run to run: 32 x 128 x 128 maps, every dilation,
    three launches each, plus one small ragged map."""
    import ctypes
    from ubdvss_amd import _lib
    lib = _lib.load()
    m = Model(NetConfig(grey=False), seed=1)
    w = m.get_weights()
    bias = np.arange(1, 25, dtype=np.float32) - 4.5             # some negative: relu
    for i in range(9, 21, 2):
        w[i] = np.zeros_like(w[i]); w[i + 1] = bias.copy()
    m.set_weights(w)
    ws = torch.empty(int(lib.ubd_forward_workspace_bytes(m._h, 1, 4, 4)), dtype=torch.uint8, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.ubd_pack_weights(m._h, m.params.data_ptr(), ws.data_ptr(), ws.numel(), st), "pack")
    want = torch.from_numpy(np.maximum(bias, 0.0)).cuda()
    for shape in ((32, 128, 128), (3, 37, 91)):
        x = torch.rand((*shape, 24), device="cuda") * 50.0
        y = torch.empty_like(x)
        for k in range(6):
            for _ in range(3):
                y.fill_(-1.0)
                _lib.check(lib.ubd_dilated_layer(m._h, m.params.data_ptr(), k, x.data_ptr(), y.data_ptr(), *shape, ws.data_ptr(), st), "dil")
                torch.cuda.synchronize()
                bad = (y.reshape(-1, 24) != want).any(dim=1)
                assert not bool(bad.any()), (shape, k, int(bad.sum()))


def test_cold_tiles_equal_the_strip_walk_at_every_width(monkeypatch):
    """stem123.h COLD (the default of launches too small for strips; forced here with UBD_STEM=cold123): single cold-started tiles 15 L3
    columns apart against the strip walk (UBD_STEM=fused123) at the widths where the number of tiles per row, the last tile's share and the
    16-byte alignment of the patch rows change (W4 = 1 .. 129), fp32 (LDS-DMA and register staging) and uint8 input, grey and RGB, heights
    that cut the last tile row: logits BIT-EQUAL."""
    from ubdvss_amd import PreprocessingType
    rng = np.random.default_rng(5)
    for k, w4 in enumerate((1, 2, 15, 16, 17, 30, 31, 32, 33, 45, 46, 47, 61, 76, 127, 128, 129)):
        cin, u8 = (1 if k % 2 else 3), (k % 3 == 1)
        n, hh, ww = 1 + k % 2, 4 * int(rng.integers(1, 12)), 4 * w4
        w = onet.init_weights(900 + k, cin, 0, bias_scale=0.25)
        cfg = NetConfig(grey=(cin == 1), fml_compatible=True, preprocessing=PreprocessingType.MOBILENET_LIKE if u8 else PreprocessingType.NONE)
        x = rng.integers(0, 256, (n, hh, ww, cin), dtype=np.uint8) if u8 else synthetic.noise_images(60 + k, n, hh, ww, cin)
        if not u8 and k % 4 == 0: x = x[:, :, :, :] * np.float32(1.0) + np.float32(0.0)      # a fresh, 16-byte aligned array either way
        outs = {}
        for mode in ("fused123", "cold123"):
            monkeypatch.setenv("UBD_STEM", mode)
            m = Model(cfg); m.set_weights(w)
            outs[mode] = m.predict(x)
        assert np.array_equal(outs["cold123"], outs["fused123"]), (w4, hh, cin, u8, float(np.abs(outs["cold123"] - outs["fused123"]).max()))
        ref = onet.forward(((x.astype(np.float64) - 127.5) / 127.5) if u8 else x.astype(np.float64), w, True)
        _check(outs["cold123"], ref)
