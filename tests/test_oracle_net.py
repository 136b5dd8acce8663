"""CPU: the two independent network restatements agree, analytic known-answer cases hold and the
golden vectors are reproduced."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import net_numpy as onet, net_torch as otorch
from ubdvss_amd import synthetic


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("cin,ncls,fml", [(3, 0, True), (1, 3, True), (3, 2, False), (1, 0, False)])
def test_numpy_vs_torch_fp64(cin, ncls, fml):
    w = onet.init_weights(1, cin, ncls, bias_scale=0.1)
    x = synthetic.noise_images(0, 2, 64, 128, cin).astype(np.float64)
    a = onet.forward(x, w, fml)
    b = otorch.forward_numpy(x, w, fml, torch.float64)
    assert a.shape == (2, 16, 32, 1 + ncls)
    assert np.abs(a - b).max() <= 1e-12


def test_param_counts():
    # grey 32 962, RGB 33 028 at n_cls = 0
    assert onet.n_params(1, 0) == 32962
    assert onet.n_params(3, 0) == 33028
    assert onet.n_params(3, 8) == 33028 + 25 * 8


def test_stride2_padding_equivalence():
    """top-left pad 1 + 'valid' stride 2 == symmetric pad 1 stride 2 on even sizes; TF 'SAME' differs."""
    rng = np.random.default_rng(0)
    x = rng.normal(size=(1, 16, 16, 3))
    w = onet.init_weights(2, 3, 0)
    fml = onet.separable_conv(x, w[0], w[1], w[2], 2, True)
    xt = torch.as_tensor(x).permute(0, 3, 1, 2)
    dw = torch.as_tensor(w[0]).permute(2, 3, 0, 1).double()
    sym = torch.nn.functional.conv2d(xt, dw, None, stride=2, padding=1, groups=3)
    pw = torch.as_tensor(w[1]).permute(3, 2, 0, 1).double()
    sym = torch.relu(torch.nn.functional.conv2d(sym, pw, torch.as_tensor(w[2]).double())).permute(0, 2, 3, 1).numpy()
    assert np.abs(fml - sym).max() < 1e-12
    same = onet.separable_conv(x, w[0], w[1], w[2], 2, False)
    assert np.abs(fml - same).max() > 1e-3


def test_identity_kernels_pass_through():
    """IdentityInitializer KAT (net.py:31-41): centre-tap identity kernels pass non-negative input
    through every dilated layer unchanged."""
    x = np.abs(np.random.default_rng(0).normal(size=(1, 20, 24, 24)))
    k = np.zeros((3, 3, 24, 24)); k[1, 1] = np.eye(24)
    y = x
    for d in onet.DILATIONS:
        y = onet.dilated_conv(y, k, np.zeros(24), d)
    assert np.array_equal(x, y)


def test_zero_input_is_bias_chain():
    w = onet.init_weights(3, 3, 0, bias_scale=0.3)
    lg = onet.forward(np.zeros((1, 64, 64, 3)), w)
    # interior pixels (far from the zero padding) all see the same bias chain
    inner = lg[0, 33 // 4 + 8: -(33 // 4 + 8)]
    assert lg.shape == (1, 16, 16, 1)
    v = w[2].astype(np.float64)
    v = np.maximum(v, 0)                                      # L1: relu(b)
    for i in (3, 6):                                           # L2, L3: depthwise(sum of taps) -> pointwise
        dwsum = w[i][:, :, :, 0].astype(np.float64).sum(axis=(0, 1))
        v = np.maximum((v * dwsum) @ w[i + 1][0, 0].astype(np.float64) + w[i + 2], 0)
    for i in range(9, 21, 2):
        v = np.maximum(v @ w[i].astype(np.float64).sum(axis=(0, 1)) + w[i + 1], 0)
    v = v @ w[21][0, 0].astype(np.float64) + w[22]
    # only the map centre is far enough (> 33 px in /4 units is impossible at 16x16), so check with a larger map
    lg = onet.forward(np.zeros((1, 320, 320, 3)), w)
    assert np.abs(lg[0, 40, 40] - v).max() < 1e-12


def test_threshold_formula():
    assert onet.logit_threshold(0.5) == 0.0 and np.signbit(onet.logit_threshold(0.5))   # -0.0
    assert abs(onet.logit_threshold(0.9) - np.log(9.0)) < 1e-12


def test_golden_cfg1(golden_dir, manifest):
    """BASELINE.json configs[0]: single 256x256x3 synthetic image, CPU forward-only map."""
    x = synthetic.noise_images(0, 1, 256, 256, 3)
    w = onet.init_weights(1, 3, 0)
    m = manifest["cfg1"]
    assert sha(x) == m["input_sha256"] and sha(onet.flatten_weights(w)) == m["weights_sha256"]
    gold = np.load(os.path.join(golden_dir, "cfg1_logits.npy"))
    assert sha(gold) == m["logits_sha256"]
    lg = otorch.forward_numpy(x, w, True, torch.float32)       # the other restatement, fp32
    assert np.abs(lg - gold).max() < 1e-5
    assert int((gold[..., 0] > -0.0).sum()) == m["n_positive"]


def test_golden_net_cases(golden_dir, manifest):
    for name in manifest["net_cases"]:
        d = np.load(os.path.join(golden_dir, f"net_{name}.npz"))
        w = onet.unflatten_weights(d["params"], int(d["c_in"]), int(d["n_classes"]))
        lg = otorch.forward_numpy(d["x"], w, bool(d["fml"]), torch.float64)
        assert np.abs(lg - d["logits"]).max() < 1e-6


def test_batchnorm_fold_equals_the_bn_branch():
    """net.py:248-250 (use_bn=True: conv -> BatchNormalization -> ReLU; never instantiated by the reference's own builder):
    ubdvss_amd.net.fold_batchnorm turns the 59 arrays of such a model into the 29 of the BN-free architecture; the folded
    model computes what the oracle's statement of the BN branch computes."""
    from ubdvss_amd.net import fold_batchnorm
    rng = np.random.default_rng(17)
    for cin, ncls in ((3, 0), (1, 2)):
        w = onet.init_weights(50 + cin, cin, ncls, bias_scale=0.3, dtype=np.float64)
        wbn, i = [], 0
        for n_conv in (3, 3, 3, 2, 2, 2, 2, 2, 2):
            wbn += w[i:i + n_conv]
            wbn += [rng.uniform(0.5, 1.5, 24), rng.normal(0, 0.3, 24), rng.normal(0, 0.5, 24), rng.uniform(0.2, 2.0, 24)]
            i += n_conv
        wbn += w[i:]
        assert len(wbn) == 59
        x = rng.normal(0, 1, (2, 24, 40, cin))
        ref = onet.forward_bn(x, wbn)
        folded = fold_batchnorm(wbn)
        assert [a.shape for a in folded] == [s for _, s in onet.weight_shapes(cin, ncls)]
        got = onet.forward(x, folded)
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())       # fp32 rounding of the folded arrays
        with pytest.raises(ValueError):
            fold_batchnorm(wbn[:-1])


def test_golden_bn_case(golden_dir):
    """tests/golden/net_bn_rgb.npz: input, the 59 get_weights() arrays of a use_bn=True model and its logits (generated by the
    oracle here; by the reference's conv_bn(use_bn=True) graph when UBD_GOLDEN_DIR points at make_reference_golden.py's output)."""
    from ubdvss_amd.net import fold_batchnorm
    d = np.load(os.path.join(golden_dir, "net_bn_rgb.npz"))
    wbn = [d["w%02d" % k] for k in range(59)]
    ref = onet.forward_bn(d["x"].astype(np.float64), wbn)
    assert np.abs(ref - d["logits"]).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    got = onet.forward(d["x"].astype(np.float64), fold_batchnorm(wbn))
    assert np.abs(got - d["logits"]).max() <= 2e-5 * max(1.0, np.abs(ref).max())
