"""GPU: the whole hot path on a model that really detects something.  Random-init weights give noise maps with 0-1
objects, so the model is first TRAINED on device (Trainer, the HIP train step) for a few hundred steps on synthetic
stripe-textured rectangles; then image -> logits -> threshold -> external components -> quads (ModelRunner.predict,
model_runner.py:105-138) must agree with the CPU oracle fed the same trained weights: logits within 1e-3, detection
maps identical outside a 1e-3 margin around the threshold, quads bit-exact on every image whose maps agree."""
import numpy as np
import pytest
import torch

from oracle import net_numpy as onet, cv_post as ocv
from ubdvss_amd import NetConfig, Model, ModelRunner, Trainer, Adam, synthetic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,n_cls", [("float32", 0), ("bfloat16", 3)])
def test_trained_model_image_to_quads(dtype, n_cls):
    cfg = NetConfig(class_names=[f"c{i}" for i in range(n_cls)] if n_cls else None, grey=False)
    train_model = Model(cfg, dtype=dtype, seed=5)
    tr = Trainer(train_model, Adam(lr=3e-3))
    labels = synthetic.rectangle_maps(70, 16, 48, 48, n_classes=n_cls)
    x = synthetic.textured_images(71, labels, 4, 3).astype(np.float32) / 127.5 - 1.0
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda()
    first = float(tr.train_step_on_device(xt, yt)[0])
    for _ in range(400):
        tr.train_step_on_device(xt, yt)
    last = float(tr.loss[0])
    assert last < 0.5 * first, (first, last)
    # inference in fp32 on the trained weights, fresh images of the same distribution
    w = train_model.get_weights()
    model = Model(cfg, seed=0)
    model.set_weights(w)
    labels2 = synthetic.rectangle_maps(72, 8, 48, 48, n_classes=n_cls)
    x2 = synthetic.textured_images(73, labels2, 4, 3).astype(np.float32) / 127.5 - 1.0
    runner = ModelRunner(cfg, pixel_threshold=0.5, max_objects_per_image=512)
    det, cls_logits, found = runner.predict(model, x2)
    ref = onet.forward(x2.astype(np.float64), w)
    lg = model.predict(x2)
    assert np.abs(lg - ref).max() <= 1e-3                                              # north_star: logits within 1e-3 fp32
    ref_det = (ref[..., 0] > 0.0)
    undecided = np.abs(ref[..., 0]) <= 1e-3
    assert not ((det[..., 0] != ref_det) & ~undecided).any()
    n_obj, n_checked = 0, 0
    for i in range(x2.shape[0]):
        if not np.array_equal(det[i, ..., 0], ref_det[i]):
            continue                                                                   # a pixel inside the margin flipped: not comparable
        q, c = ocv.postprocess(det[i, ..., 0].astype(np.uint8), ref[i, ..., 1:].astype(np.float32) if n_cls else None, 4, 5)
        got = np.array([o.bbox for o in found[i]]).reshape(-1, 8)
        assert np.array_equal(got, q), (i, got, q)
        if n_cls:
            assert [int(o.object_type) for o in found[i]] == [int(v) for v in c]
        n_obj += len(q); n_checked += 1
    print(f"{dtype}: loss {first:.3f} -> {last:.3f}; {n_obj} objects on {n_checked} of {x2.shape[0]} images compared bit-exact; "
          f"{int(undecided.sum())} pixels inside the margin")
    assert n_checked >= 6 and n_obj >= 8            # a trained model: several objects per image, not the 0-1 of random weights


@pytest.mark.parametrize("n_cls,u8", [(0, True), (2, False)])
def test_predict_stream_equals_predict(n_cls, u8):
    """ModelRunner.predict_stream (model_runner.py:60-67's loop as one pipeline: pinned staging, copy-in / compute / copy-out streams,
    the postprocess of batch k inside the stem kernel of batch k + 1) returns, batch by batch, what ModelRunner.predict returns --
    detection maps, class logits, object lists, rescaled boxes -- and the lists equal the ORACLE's on the device maps.  Seven batches
    of 32 x 256 x 256 (fused-stem size: 512 strips), weights that give several objects per image (a bias chain that thresholds the texture)."""
    cfg = NetConfig(class_names=[f"c{i}" for i in range(n_cls)] if n_cls else None, grey=False)
    model = Model(cfg, seed=2)
    tr = Trainer(model, Adam(lr=3e-3))
    labels = synthetic.rectangle_maps(170, 16, 64, 64, n_classes=n_cls)
    xtr = synthetic.textured_images(171, labels, 4, 3).astype(np.float32) / 127.5 - 1.0
    xt, yt = torch.from_numpy(xtr).cuda(), torch.from_numpy(labels).cuda()
    for _ in range(150):
        tr.train_step_on_device(xt, yt)
    torch.cuda.synchronize()
    batches = []
    for k in range(7):
        lab = synthetic.rectangle_maps(300 + k, 32, 64, 64, n_classes=n_cls)
        img = synthetic.textured_images(400 + k, lab, 4, 3)
        batches.append(img if u8 else img.astype(np.float32) / 127.5 - 1.0)
    if u8:
        from ubdvss_amd import PreprocessingType
        cfg_u8 = NetConfig(grey=False, preprocessing=PreprocessingType.MOBILENET_LIKE)
        m2 = Model(cfg_u8, seed=0); m2.set_weights(model.get_weights()); model, cfg = m2, cfg_u8

    class Meta:                                           # model_runner.py:140-148 reads xscale / yscale
        def __init__(self, xs, ys): self.xscale, self.yscale = xs, ys
    metas = [[Meta(1.0 + 0.1 * (i % 3), 0.9 + 0.05 * (i % 5)) for i in range(32)] for _ in range(7)]
    runner = ModelRunner(cfg, pixel_threshold=0.5, max_objects_per_image=512)
    want = [runner.predict(model, b, rescale=True, meta_infos=metas[k]) for k, b in enumerate(batches)]
    got = list(runner.predict_stream(model, batches, rescale=True, meta_infos=metas))
    assert len(got) == len(want) == 7
    n_obj = 0
    for k in range(7):
        assert np.array_equal(got[k][0], want[k][0]) and got[k][0].dtype == want[k][0].dtype
        assert np.array_equal(got[k][1], want[k][1])
        for i in range(32):
            a = [(tuple(int(v) for v in o.bbox), getattr(o, "object_type", None)) for o in got[k][2][i]]
            b = [(tuple(int(v) for v in o.bbox), getattr(o, "object_type", None)) for o in want[k][2][i]]
            assert a == b, (k, i, a, b)
            n_obj += len(a)
    # and against the oracle's postprocess on the device's detection maps (unrescaled)
    plain = list(runner.predict_stream(model, batches[:2]))
    for k in range(2):
        lg = model.predict(batches[k])
        for i in range(0, 32, 5):
            q, c = ocv.postprocess(plain[k][0][i, ..., 0].astype(np.uint8), lg[i, ..., 1:] if n_cls else None, 4, 5)
            assert np.array_equal(np.array([o.bbox for o in plain[k][2][i]]).reshape(-1, 8), q)
    assert n_obj >= 7 * 32, n_obj                       # about one object per image or more, not empty lists


def test_predict_stream_argument_errors_and_early_exit():
    """predict_stream with rescale=True and no meta_infos refuses up front (predict's rule); meta_infos shorter than the
    batches surface as a ValueError naming the batch, not as 'generator raised StopIteration'; a consumer that stops after the first
    result (its prefetched copies still in flight) leaves the runner and the model usable: the next stream gives the full, correct results."""
    cfg = NetConfig(grey=False)
    model = Model(cfg, seed=2)
    runner = ModelRunner(cfg, pixel_threshold=0.5, max_objects_per_image=256)
    batches = [synthetic.noise_images(50 + k, 8, 128, 128, 3) for k in range(5)]

    class Meta:
        xscale, yscale = 1.0, 1.0
    with pytest.raises(AssertionError):
        list(runner.predict_stream(model, batches, rescale=True))
    with pytest.raises(ValueError, match="meta_infos ended"):
        list(runner.predict_stream(model, batches, rescale=True, meta_infos=[[Meta()] * 8] * 2))
    want = [runner.predict(model, b) for b in batches]
    gen = runner.predict_stream(model, batches)
    first = next(gen)
    gen.close()                                            # early exit with batches 1, 2 prefetched
    assert np.array_equal(first[0], want[0][0])
    got = list(runner.predict_stream(model, batches))
    assert len(got) == 5
    for k in range(5):
        assert np.array_equal(got[k][0], want[k][0]) and np.array_equal(got[k][1], want[k][1])
        assert [[tuple(int(v) for v in o.bbox) for o in img] for img in got[k][2]] == [[tuple(int(v) for v in o.bbox) for o in img] for img in want[k][2]]
