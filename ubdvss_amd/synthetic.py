"""Synthetic workloads for tests and bench.py.

The reference dataset is not released (README.md:9), so every measurement uses synthetic data of
the reference's shapes: uint8 noise images, label / detection maps made of random rotated
rectangles rasterised with PIL ``ImageDraw.polygon`` exactly like
``SegmapManager.build_segmentation_map`` (segmap_manager.py:81-104), and stripe-textured images
whose rectangles sit where the labels are.
"""
import numpy as np
from PIL import Image, ImageDraw


def noise_images(seed, n, height, width, c_in, as_float=True):
    """uint8 uniform noise, optionally mobilenet-preprocessed to float32 in [-1, 1]."""
    x = np.random.default_rng(seed).integers(0, 256, (n, height, width, c_in), dtype=np.uint8)
    if not as_float:
        return x
    return ((x.astype(np.float32) - 127.5) / 127.5).astype(np.float32)


def random_quads(rng, map_h, map_w, n_min=1, n_max=8, side_min=6, side_max=60):
    """1..8 random rotated rectangles (sides 6..60 px at 128x128, scaled with the map)."""
    s = min(map_h, map_w) / 128.0
    quads = []
    for _ in range(int(rng.integers(n_min, n_max + 1))):
        a = rng.uniform(side_min, side_max) * s
        b = rng.uniform(side_min, side_max) * s
        ang = rng.uniform(0, np.pi)
        cx = rng.uniform(0.15, 0.85) * map_w
        cy = rng.uniform(0.15, 0.85) * map_h
        ca, sa = np.cos(ang), np.sin(ang)
        pts = []
        for dx, dy in ((-a / 2, -b / 2), (a / 2, -b / 2), (a / 2, b / 2), (-a / 2, b / 2)):
            pts.append((cx + dx * ca - dy * sa, cy + dx * sa + dy * ca))
        quads.append(np.array(pts))
    return quads


def rectangle_maps(seed, n, map_h, map_w, n_classes=0, n_min=1, n_max=8, side_min=6, side_max=60):
    """(n, map_h, map_w) int32 label maps: 0 background, 1..max(n_classes,1) objects (n_min..n_max rotated rectangles per map,
    sides side_min..side_max pixels at 128 x 128; overlapping rectangles merge into one object)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, map_h, map_w), np.int32)
    for i in range(n):
        im = Image.new(mode='L', size=(map_w, map_h), color=0)
        draw = ImageDraw.Draw(im)
        for q in random_quads(rng, map_h, map_w, n_min, n_max, side_min, side_max):
            fill = int(rng.integers(1, n_classes + 1)) if n_classes > 0 else 1
            draw.polygon([(int(round(x)), int(round(y))) for x, y in q], fill=fill)
        out[i] = np.asarray(im, dtype=np.int32)
    return out


def crowded_maps(seed, n, map_h, map_w, cells=8, fill=0.8):
    """(n, map_h, map_w) int32 {0,1} maps with MANY separate objects: one small rotated rectangle (sides 25-55 % of a cell) in ``fill`` of the
    cells of a ``cells`` x ``cells`` grid, jittered inside its cell so that neighbours do not touch -- about 50 objects per map at 8 x 8."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, map_h, map_w), np.int32)
    ch, cw = map_h / cells, map_w / cells
    for i in range(n):
        im = Image.new(mode='L', size=(map_w, map_h), color=0)
        draw = ImageDraw.Draw(im)
        for gy in range(cells):
            for gx in range(cells):
                if rng.uniform() > fill:
                    continue
                a, b = rng.uniform(0.25, 0.55) * cw, rng.uniform(0.25, 0.55) * ch
                ang = rng.uniform(0, np.pi)
                cx, cy = (gx + rng.uniform(0.42, 0.58)) * cw, (gy + rng.uniform(0.42, 0.58)) * ch
                ca, sa = np.cos(ang), np.sin(ang)
                pts = [(cx + dx * ca - dy * sa, cy + dx * sa + dy * ca) for dx, dy in ((-a / 2, -b / 2), (a / 2, -b / 2), (a / 2, b / 2), (-a / 2, b / 2))]
                draw.polygon([(int(round(x)), int(round(y))) for x, y in pts], fill=1)
        out[i] = np.asarray(im, dtype=np.int32)
    return out


def textured_images(seed, labels, scale, c_in):
    """uint8 images (n, h*scale, w*scale, c_in): grey background noise, stripe texture (barcode-like)
    where ``labels`` > 0."""
    rng = np.random.default_rng(seed)
    n, h, w = labels.shape
    big = np.repeat(np.repeat(labels > 0, scale, axis=1), scale, axis=2)
    hh, ww = h * scale, w * scale
    xs = np.arange(ww)[None, None, :]
    stripes = ((xs // 3) % 2 * 255).astype(np.uint8)
    bg = rng.integers(96, 160, (n, hh, ww), dtype=np.uint8)
    img = np.where(big, np.broadcast_to(stripes, (n, hh, ww)), bg).astype(np.uint8)
    return np.repeat(img[..., None], c_in, axis=3)


def logits_from_maps(maps, n_classes=0, seed=0, margin=4.0, noise=0.5):
    """fp32 logits (n,h,w,1+n_classes) whose channel 0 is +margin on objects / -margin elsewhere and
    whose class channels favour the labelled class (plus noise) -- a stand-in for trained output
    used to benchmark / test the postprocess alone."""
    rng = np.random.default_rng(seed)
    n, h, w = maps.shape
    lg = (rng.normal(0, 1.0, (n, h, w, 1 + n_classes)) * noise).astype(np.float32)
    lg[..., 0] += np.where(maps > 0, margin, -margin).astype(np.float32)
    if n_classes > 0:
        idx = np.clip(maps - 1, 0, n_classes - 1)
        onehot = np.eye(n_classes, dtype=np.float32)[idx] * (maps > 0)[..., None]
        lg[..., 1:] += 3.0 * onehot
    return lg
