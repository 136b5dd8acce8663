"""Oracle: Pillow's polygon fill restated (training label maps).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The reference fills object quads with ``PIL.ImageDraw.polygon``
(semantic_segmentation/segmap_manager.py:93-103); the fill rule itself lives in Pillow (libImaging/Draw.c, third-party, not
in the reference tree).  This plain-Python restatement of that rule is PINNED against the installed Pillow (12.2.0) by
tests/test_oracle_raster.py: identical on every convex quadrilateral / rotated rectangle / axis-aligned box / single point
tried (tens of thousands); self-intersecting and zero-area quads can differ at concave corners (Pillow joins corners with
heuristics that are not restated) -- the tests report the rate.  The device kernel (ubdvss_amd/csrc/raster.hip) implements
exactly this function and is compared with Pillow itself on the GPU box.
"""
import math

import numpy as np

_f32 = np.float32


def round_up(f):        # Pillow's ROUND_UP: round half up
    return int(math.floor(f + 0.5)) if f >= 0 else -int(math.floor(abs(f) + 0.5))


def round_down(f):      # Pillow's ROUND_DOWN: round half down
    return int(math.ceil(f - 0.5)) if f >= 0 else -int(math.ceil(abs(f) - 0.5))


def proper_round(bbox, scale):
    """segmap_manager.py:106-133 on bbox / scale: floor a coordinate when at least two of the four coordinates on its axis
    are strictly larger, else ceil."""
    pts = np.asarray(bbox, dtype=np.float64).reshape(4, 2) / scale
    n_larger = (pts[None, :, :] > pts[:, None, :]).sum(axis=1)
    return np.where(n_larger > 1, np.floor(pts), np.ceil(pts)).reshape(-1).astype(np.int64)


def fill_polygon(out, pts, value):
    """ImageDraw.polygon(pts, fill=value) on the 2-D array ``out`` for a 4-vertex polygon given as 8 integers."""
    h, w = out.shape
    n = len(pts) // 2
    xs = [int(pts[2 * i]) for i in range(n)]
    ys = [int(pts[2 * i + 1]) for i in range(n)]

    def hline(x0, y, x1):
        if 0 <= y < h:
            if x0 > x1:
                x0, x1 = x1, x0
            x0, x1 = max(x0, 0), min(x1, w - 1)
            if x0 <= x1:
                out[y, x0:x1 + 1] = value

    edges, table = [], []
    ymin, ymax = h - 1, 0
    for i in range(n):
        x0, y0, x1, y1 = xs[i], ys[i], xs[(i + 1) % n], ys[(i + 1) % n]
        e = dict(x0=x0, y0=y0, x1=x1, y1=y1, xmin=min(x0, x1), xmax=max(x0, x1), ymin=min(y0, y1), ymax=max(y0, y1))
        e["dx"] = _f32(0) if y0 == y1 else _f32(_f32(x1 - x0) / _f32(y1 - y0))
        edges.append(e)
        ymin, ymax = min(ymin, e["ymin"]), max(ymax, e["ymax"])
        if y0 == y1:
            hline(e["xmin"], y0, e["xmax"])             # horizontal edges are drawn as they are
        else:
            table.append(e)
    ymin, ymax = max(ymin, 0), min(ymax, h)

    def x_at(e, y):                                      # float32, product and sum rounded separately
        return _f32(_f32(_f32(y - e["y0"]) * e["dx"]) + _f32(e["x0"]))

    spans = {}
    for y in range(ymin, ymax + 1):
        xx = []
        for e in table:
            if e["ymin"] <= y <= e["ymax"]:
                x = x_at(e, y)
                xx.append(x)
                if y == e["ymax"] and y < ymax:          # an edge's lower end point counts twice
                    xx.append(x)
        xx.sort()
        row = []
        x_pos = int(xx[0]) if xx else 0
        for i in range(1, len(xx), 2):
            x_end = round_down(xx[i])
            if x_end < x_pos:
                continue
            x_start = round_up(xx[i - 1])
            if x_pos > x_start:
                x_start = x_pos
                if x_end < x_start:
                    continue
            if x_start > x_end:
                continue
            row.append([x_start, x_end])
            x_pos = x_end + 1
        spans[y] = row
    if table:                                            # join the single pixel of a top / bottom corner to the neighbouring row
        tmin, tmax = min(e["ymin"] for e in table), max(e["ymax"] for e in table)
        for yv, ya in ((tmin, tmin + 1), (tmax, tmax - 1)):
            if tmin == tmax or yv not in spans or len(spans[yv]) != 1:
                continue
            if any(e["ymin"] == e["ymax"] == yv for e in edges):
                continue
            act = [e for e in table if e["ymin"] <= yv <= e["ymax"]]
            if len(act) != 2:
                continue
            ends = [{(e["x0"], e["y0"]), (e["x1"], e["y1"])} for e in act]
            common = [p for p in ends[0] & ends[1] if p[1] == yv]
            if len(common) != 1:
                continue
            adj = [x_at(e, ya) for e in act]
            v = float(common[0][0])
            if min(adj) > v:
                spans[yv][0][1] = max(spans[yv][0][1], round_up(_f32(min(adj) - _f32(1))))
            elif max(adj) < v:
                spans[yv][0][0] = min(spans[yv][0][0], round_up(_f32(max(adj) + _f32(1))))
    for y, row in spans.items():
        for a, b in row:
            hline(a, y, b)


def build_label_map(height, width, quads, values, scale):
    """SegmapManager.build_segmentation_map (segmap_manager.py:81-104) as an int32 array (height/scale, width/scale)."""
    out = np.zeros((height // scale, width // scale), np.int32)
    for q, v in zip(quads, values):
        fill_polygon(out, proper_round(q, scale), int(v))
    return out
