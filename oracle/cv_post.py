"""ctypes front-end of the C oracle (oracle/cv_post.c) plus numpy helpers.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED vs OpenCV 3.4
(absent here).  Mirrors the reference call sites
semantic_segmentation/utils.py:51-60, segmap_manager.py:41-69,
model_runner.py:121-148, utils.py:67-69, :135-138.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libubd_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "cv_post.c")
        if not os.path.exists(_SO) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO)):
            build()
        L = ctypes.CDLL(_SO)
        i32p = ctypes.POINTER(ctypes.c_int32)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        f32p = ctypes.POINTER(ctypes.c_float)
        f64p = ctypes.POINTER(ctypes.c_double)
        L.ubdo_find_external_contours.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  i32p, ctypes.c_int, i32p, ctypes.c_int]
        L.ubdo_find_external_contours.restype = ctypes.c_int
        L.ubdo_contour_area.argtypes = [i32p, ctypes.c_int]
        L.ubdo_contour_area.restype = ctypes.c_double
        L.ubdo_convex_hull.argtypes = [i32p, ctypes.c_int, i32p]
        L.ubdo_convex_hull.restype = ctypes.c_int
        L.ubdo_min_area_rect.argtypes = [i32p, ctypes.c_int, f32p]
        L.ubdo_box_points.argtypes = [f32p, f32p]
        L.ubdo_postprocess.argtypes = [u8p, ctypes.c_int, ctypes.c_int, f32p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_double, i32p, i32p, f64p, ctypes.c_int]
        L.ubdo_postprocess.restype = ctypes.c_int
        L.ubdo_fill_contour.argtypes = [i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, u8p]
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def find_contours(seg_map, approx_simple=True):
    """cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE|NONE) -> list of (n,2) int
    arrays in cv2's return order (last discovered first)."""
    img = np.ascontiguousarray(np.asarray(seg_map).reshape(seg_map.shape[0], seg_map.shape[1]) != 0, dtype=np.uint8)
    h, w = img.shape
    pts_cap = 4 * (h + 2) * (w + 2) + 16
    cont_cap = h * w // 2 + 16
    pts = np.zeros((pts_cap, 2), np.int32)
    offs = np.zeros(cont_cap + 1, np.int32)
    n = lib().ubdo_find_external_contours(_p(img, ctypes.c_uint8), h, w, int(approx_simple),
                                          _p(pts, ctypes.c_int32), pts_cap, _p(offs, ctypes.c_int32), cont_cap)
    assert n >= 0
    return [pts[offs[i]:offs[i + 1]].copy() for i in range(n - 1, -1, -1)]


def contour_area(cnt):
    c = np.ascontiguousarray(cnt, dtype=np.int32)
    return lib().ubdo_contour_area(_p(c, ctypes.c_int32), len(c))


def convex_hull(points):
    c = np.ascontiguousarray(points, dtype=np.int32)
    out = np.zeros((max(len(c), 1), 2), np.int32)
    n = lib().ubdo_convex_hull(_p(c, ctypes.c_int32), len(c), _p(out, ctypes.c_int32))
    return out[:n]


def min_area_rect(points):
    c = np.ascontiguousarray(points, dtype=np.int32)
    r = np.zeros(5, np.float32)
    lib().ubdo_min_area_rect(_p(c, ctypes.c_int32), len(c), _p(r, ctypes.c_float))
    return r


def box_points(rect5):
    r = np.ascontiguousarray(rect5, dtype=np.float32)
    out = np.zeros(8, np.float32)
    lib().ubdo_box_points(_p(r, ctypes.c_float), _p(out, ctypes.c_float))
    return out


def fill_contour(cnt, h, w):
    c = np.ascontiguousarray(cnt, dtype=np.int32)
    m = np.zeros((h, w), np.uint8)
    lib().ubdo_fill_contour(_p(c, ctypes.c_int32), len(c), h, w, _p(m, ctypes.c_uint8))
    return m


def postprocess(seg_map, seg_map_class_logits=None, scale=1, min_area_threshold=5, return_areas=False):
    """SegmapManager.postprocess (segmap_manager.py:41-69) for one image.
    Returns (quads (n,8) int32, class_ids (n,) int32 or None[, areas])."""
    img = np.ascontiguousarray(np.asarray(seg_map).reshape(seg_map.shape[0], seg_map.shape[1]) != 0, dtype=np.uint8)
    h, w = img.shape
    cap = h * w // 2 + 16
    quads = np.zeros((cap, 8), np.int32)
    cls = np.zeros(cap, np.int32)
    areas = np.zeros(cap, np.float64)
    if seg_map_class_logits is not None:
        lg = np.ascontiguousarray(seg_map_class_logits, dtype=np.float32)
        n_cls = lg.shape[-1]
        lgp = _p(lg, ctypes.c_float)
    else:
        n_cls, lgp = 0, None
    n = lib().ubdo_postprocess(_p(img, ctypes.c_uint8), h, w, lgp, n_cls, int(scale), float(min_area_threshold),
                               _p(quads, ctypes.c_int32), _p(cls, ctypes.c_int32), _p(areas, ctypes.c_double), cap)
    assert n >= 0
    res = (quads[:n].copy(), cls[:n].copy() if n_cls else None)
    return res + (areas[:n].copy(),) if return_areas else res


def np_softmax(logits, axis=-1):
    """utils.py:135-138."""
    x = logits - np.max(logits, axis=axis, keepdims=True)
    x = np.exp(x)
    return x / np.sum(x, axis=axis, keepdims=True)


def rescale_bbox(bbox, xscale, yscale):
    """utils.py:67-69 (truncation toward zero)."""
    scale = np.array([xscale, yscale] * (len(bbox) // 2))
    return (bbox * scale).astype(int)


def predict_postprocess(logits, n_classes, pixel_threshold=0.5, scale=4, min_area=5):
    """ModelRunner.predict after model.predict (model_runner.py:121-134).
    Returns (binary map (N,h,w,1) int, class logits, [ (quads, cls) per image ])."""
    eps = 1e-9
    thr = -np.log(1 / np.clip(pixel_threshold, eps, 1 - eps) - 1)
    det = np.where(logits[..., :1] > thr, 1, 0)
    cl = logits[..., 1:]
    found = [postprocess(det[i], cl[i] if n_classes > 0 else None, scale, min_area) for i in range(logits.shape[0])]
    return det, cl, found
