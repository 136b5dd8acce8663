"""GPU: device label rasteriser (ubd_build_label_maps; reference SegmapManager.build_segmentation_map + _proper_round,
segmap_manager.py:81-133) against Pillow itself -- the engine the reference calls -- through the host mirror, bit-exact for
convex object quads (rotated rectangles, perspective quads, boxes touching / leaving the image, class values, painter's
order), and against the pinned oracle restatement for arbitrary quads."""
import numpy as np
import pytest
import torch
from PIL import Image
from scipy.spatial import ConvexHull

from oracle import label_raster as olr
from ubdvss_amd import SegmapManager, synthetic
from ubdvss_amd.data_markup import ObjectMarkup, ClassifiedObjectMarkup

pytestmark = pytest.mark.gpu


def _pil_maps(size, markups, scale):
    return np.stack([np.asarray(SegmapManager.build_segmentation_map(Image.new("L", size), m, scale=scale)).astype(np.int32) for m in markups])


def _convex_quad(rng, w, h):
    while True:
        p = np.stack([rng.integers(-8, w + 8, 4), rng.integers(-8, h + 8, 4)], axis=1)
        try:
            hull = ConvexHull(p)
        except Exception:
            continue
        if len(hull.vertices) == 4:
            return p[hull.vertices].reshape(-1)


@pytest.mark.parametrize("scale", [4, 1, 2])
def test_convex_markup_equals_pillow(scale):
    rng = np.random.default_rng(31 + scale)
    w, h = 64 * scale, 48 * scale
    markups = []
    for i in range(48):
        objs = []
        for _ in range(int(rng.integers(0, 7))):
            if rng.random() < 0.5:
                q = np.round(np.asarray(synthetic.random_quads(rng, h, w, 1, 1)[0])).astype(int).reshape(-1)
            else:
                q = _convex_quad(rng, w, h)
            objs.append(ClassifiedObjectMarkup(q, int(rng.integers(0, 6))) if i % 2 else ObjectMarkup(q))
        markups.append(objs)
    got = SegmapManager.build_segmentation_maps_on_device((w, h), markups, scale=scale).cpu().numpy()
    ref = _pil_maps((w, h), markups, scale)
    assert got.shape == ref.shape and np.array_equal(got, ref)
    assert (got > 0).any()


def test_full_size_batch_and_train_step_consumes_the_labels():
    """configs[2] label shape (64 maps of 128 x 128 from 512 x 512 images) + the maps feed ubd_train_step directly."""
    from ubdvss_amd import NetConfig, Model, Trainer
    rng = np.random.default_rng(40)
    markups = [[ObjectMarkup(np.round(np.asarray(q) * 4).astype(int).reshape(-1)) for q in synthetic.random_quads(rng, 128, 128)] for _ in range(64)]
    lab = SegmapManager.build_segmentation_maps_on_device((512, 512), markups, scale=4)
    assert np.array_equal(lab.cpu().numpy(), _pil_maps((512, 512), markups, 4))
    x = torch.from_numpy(synthetic.textured_images(41, lab[:4].cpu().numpy(), 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
    tr = Trainer(Model(NetConfig(grey=False), seed=1))
    tr.train_step_on_device(x, lab[:4])
    assert torch.isfinite(tr.loss).all() and float(tr.loss[0]) > 0


def test_arbitrary_quads_equal_the_pinned_restatement():
    """Self-intersecting / degenerate quads: the device kernel implements oracle/label_raster.py exactly (which is pinned
    against Pillow on CPU and differs from it only at concave corners of such quads)."""
    rng = np.random.default_rng(44)
    markups = [[ObjectMarkup(rng.integers(-10, 200, 8))] for _ in range(256)]
    got = SegmapManager.build_segmentation_maps_on_device((192, 160), markups, scale=4).cpu().numpy()
    for i, m in enumerate(markups):
        assert np.array_equal(got[i], olr.build_label_map(160, 192, [m[0].bbox], [1], 4)), m[0].bbox
