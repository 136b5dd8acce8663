"""GPU parity of the use_bn=True branch (net.py:248-250: conv with bias, no activation -> BatchNormalization -> ReLU).  The device
path has no BatchNormalization kernel: ``Model.set_weights`` folds gamma / sqrt(moving_variance + 1e-3) into the layer's last
kernel and bias (ubdvss_amd.net.fold_batchnorm) and the ordinary HIP pass runs.  Checked here THROUGH libubd_hip.so against the
oracle's statement of the BN branch itself (oracle/net_numpy.forward_bn: unfolded fp64 arithmetic), fp32 and both 16-bit modes,
from a weight list and from a Keras ``model.save`` file of such a model (tests/golden/keras_model_rgb_bn.h5, written by h5py)."""
import os

import numpy as np
import pytest
import torch

from oracle import net_numpy as onet
from ubdvss_amd import NetConfig, Model, NetManager, keras_h5, synthetic
from ubdvss_amd.net import fold_batchnorm

pytestmark = pytest.mark.gpu

GATE16 = {"bfloat16": 2.5e-2, "float16": 5e-3}                  # tests/test_gpu_forward16.py GATE_FP64


def _bn_weights(seed, cin, ncls):
    """59 (+0) arrays in get_weights() order of a use_bn model: per hidden layer its conv arrays, then gamma, beta, moving mean,
    moving variance; the head last."""
    rng = np.random.default_rng(seed)
    w = onet.init_weights(seed, cin, ncls, bias_scale=0.3)
    wbn, i = [], 0
    for n_conv in (3, 3, 3, 2, 2, 2, 2, 2, 2):
        wbn += w[i:i + n_conv]
        wbn += [rng.uniform(0.5, 1.5, 24).astype(np.float32), rng.normal(0, 0.3, 24).astype(np.float32),
                rng.normal(0, 0.5, 24).astype(np.float32), rng.uniform(0.2, 2.0, 24).astype(np.float32)]
        i += n_conv
    return wbn + w[i:]


def _check32(lg, ref, thr=-0.0):
    err = np.abs(lg.astype(np.float64) - ref).max()
    assert err <= 2e-5 * np.abs(ref).max() + 1e-6, (err, np.abs(ref).max())
    far = np.abs(ref[..., 0] - thr) > max(1e-3, 2e-5 * np.abs(ref).max())
    assert np.array_equal((lg[..., 0] > thr)[far], (ref[..., 0] > thr)[far])


def _check16(lg, ref, dtype, thr=0.0):
    scale = float(np.abs(ref).max())
    assert np.abs(lg - ref).max() <= GATE16[dtype] * scale, (np.abs(lg - ref).max() / scale, dtype)
    decided = np.abs(ref[..., 0] - thr) > GATE16[dtype] * scale
    assert np.array_equal((lg[..., 0] > thr)[decided], (ref[..., 0] > thr)[decided])


@pytest.mark.parametrize("cin,ncls,fml,n,hh,ww", [(3, 0, True, 2, 96, 128), (1, 2, True, 1, 72, 100), (3, 1, False, 2, 64, 72)])
def test_bn_weight_list_vs_oracle_bn_branch(cin, ncls, fml, n, hh, ww):
    cfg = NetConfig(class_names=[f"c{i}" for i in range(ncls)] if ncls else None, grey=(cin == 1), fml_compatible=fml)
    wbn = _bn_weights(60 + cin + ncls, cin, ncls)
    assert len(wbn) == 59
    x = synthetic.noise_images(61, n, hh, ww, cin)
    ref = onet.forward_bn(x.astype(np.float64), wbn, fml)
    m = Model(cfg)
    m.set_weights(wbn)                                            # 59 arrays: folded on the way in
    _check32(m.predict(x), ref)
    for dtype in ("bfloat16", "float16"):
        m16 = Model(cfg, dtype=dtype)
        m16.set_weights(wbn)
        _check16(m16.predict(x), ref, dtype)


def test_bn_inside_the_one_kernel_stem_and_pipelined_runner(monkeypatch):
    """The folded model through the inference fast path (one-kernel stem, head in L9's epilogue) on a launch big enough for it."""
    monkeypatch.setenv("UBD_STEM", "fused123")
    cfg = NetConfig(grey=False)
    wbn = _bn_weights(71, 3, 0)
    x = synthetic.noise_images(72, 3, 128, 160, 3)
    m = Model(cfg)
    m.set_weights(wbn)
    _check32(m.predict(x), onet.forward_bn(x.astype(np.float64), wbn))


def test_bn_keras_file_vs_oracle_bn_branch(tmp_path, golden_dir):
    """keras_model_rgb_bn.h5 (model.save of a use_bn model, written by the real h5py): NetManager.load_model reads it (59 arrays),
    folds and runs; logits vs forward_bn on the arrays as stored."""
    import shutil
    src = os.path.join(golden_dir, "keras_model_rgb_bn.h5")
    arrays, _ = keras_h5.read_keras_weights(src)
    assert len(arrays) == 59
    # (the fixture's uniform(-1, 1) kernels give logits of ~1e6: every bound below is relative to max|logit|)
    x = synthetic.noise_images(81, 2, 64, 96, 3)
    ref = onet.forward_bn(x.astype(np.float64), arrays)
    shutil.copy(src, tmp_path / "inference_model.h5")
    cfg = NetConfig(grey=False)
    mgr = NetManager(str(tmp_path), cfg)
    mgr.load_model()
    lg = mgr.get_keras_model().predict(x)
    assert np.isfinite(lg).all()
    _check32(lg, ref)
    folded = fold_batchnorm(arrays)
    assert np.array_equal(np.concatenate([a.ravel() for a in folded]), mgr.get_keras_model().params.cpu().numpy())


def test_bn_golden_fixture(golden_dir):
    """tests/golden/net_bn_rgb.npz (oracle output; the reference's own when UBD_GOLDEN_DIR names a make_reference_golden.py run)."""
    d = np.load(os.path.join(golden_dir, "net_bn_rgb.npz"))
    wbn = [d["w%02d" % k] for k in range(59)]
    m = Model(NetConfig(grey=False))
    m.set_weights(wbn)
    _check32(m.predict(d["x"]), d["logits"].astype(np.float64))
    for dtype in ("bfloat16", "float16"):
        m16 = Model(NetConfig(grey=False), dtype=dtype)
        m16.set_weights(wbn)
        _check16(m16.predict(d["x"]), d["logits"].astype(np.float64), dtype)
