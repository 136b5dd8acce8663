"""Persistent-kernel paths on oracle-checkable shapes.

Every stem / dilated / backward kernel is a persistent loop over tiles with prefetch, LDS double buffering and
counted waits; on the small shapes the CPU oracle finishes in seconds a block normally sees one tile and those
paths would only run at BASELINE.json's full sizes.  UBD_TEST_NUM_CUS makes the library size its grids for a
one-CU device, so each block walks dozens of tiles here (a prefetch racing with the reads of the previous tile was
found this way).  Same gates as test_gpu_forward.py / test_gpu_train.py / test_gpu_forward16.py.
"""
import numpy as np
import pytest
import torch

from oracle import net_numpy as onet, net_torch as otorch
from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture()
def one_cu(monkeypatch):
    monkeypatch.setenv("UBD_TEST_NUM_CUS", "1")


@pytest.mark.parametrize("dtype,tol", [("float32", 1e-3), ("bfloat16", 2e-2), ("float16", 4e-3)])
def test_forward_many_tiles_per_block(one_cu, dtype, tol):
    cin, ncls, n, hh, ww = 3, 2, 3, 136, 200
    cfg = NetConfig(class_names=["a", "b"], grey=False)
    m = Model(cfg, dtype=dtype, seed=0)
    assert m.num_cus == 1                                    # the override reached ubd_create: grids are sized for one CU
    w = onet.init_weights(5, cin, ncls, bias_scale=0.2)
    m.set_weights(w)
    x = synthetic.noise_images(6, n, hh, ww, cin)
    got = m.predict_on_device(torch.from_numpy(x).cuda()).cpu().numpy()
    ref = onet.forward(x.astype(np.float64), w, True, act_dtype=None if dtype == "float32" else dtype)
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= tol * scale + 1e-5


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_train_step_many_tiles_per_block(one_cu, dtype):
    cin, ncls, n, hh, ww = 3, 0, 2, 160, 224
    cfg = NetConfig(grey=False)
    model = Model(cfg, dtype=dtype, seed=0)
    assert model.num_cus == 1
    w = onet.init_weights(93, cin, ncls, bias_scale=0.2)
    w[-2] = (w[-2] * 4).astype(np.float32)
    model.set_weights(w)
    labels = synthetic.rectangle_maps(91, n, hh // 4, ww // 4)
    x = synthetic.textured_images(92, labels, 4, cin).astype(np.float32) / 127.5 - 1.0
    tr = Trainer(model, Adam())
    tr.backward_on_device(torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda())
    g1 = tr.grads.clone()
    tr.backward_on_device(torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda())
    if dtype == "bfloat16":
        assert torch.equal(g1, tr.grads)                     # fixed-order reductions: bit-identical
    g = g1.cpu().numpy().astype(np.float64)
    act = None if dtype == "float32" else dtype
    loss_ref, _, _, grads_ref = otorch.loss_and_grads(x, labels[..., None], w, False, True, dtype=torch.float32, act_dtype=act,
                                                      grad_dtype="bfloat16" if dtype == "bfloat16" else None)
    assert abs(float(tr.loss[0]) - loss_ref) <= 1e-3 * abs(loss_ref)
    off = 0
    for (nm, _), gr in zip(onet.weight_shapes(cin, ncls), grads_ref):
        k = gr.size
        err = np.linalg.norm(g[off:off + k] - gr.reshape(-1)) / max(np.linalg.norm(gr), 1e-30)
        assert err <= (1e-3 if dtype == "float32" else 5e-3), (nm, err)
        off += k
