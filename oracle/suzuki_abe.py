"""Independent border follower, written from the PAPER and not from OpenCV's sources:

    S. Suzuki, K. Abe, "Topological Structural Analysis of Digitized Binary Images by Border Following",
    CVGIP 30 (1985) 32-46 -- Algorithm 1 (steps (1)-(4), sub-steps (3.1)-(3.5)) with the modifications of
    Appendix II ("Algorithm 2": follow only the OUTERMOST borders: marks +2 / -2 only, LNBD reset to 0 at the start
    of every row, a border is followed only from an outer-border starting point reached with LNBD <= 0).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Its purpose is to cross-check oracle/cv_post.c -- the restatement of OpenCV 3.4's
findContours(RETR_EXTERNAL, ...) that the reference calls at semantic_segmentation/utils.py:52 -- with code that shares nothing with it:
pure Python, the paper's own variable names ((i, j), (i1, j1) .. (i4, j4), NBD, LNBD), row / column coordinates.  PARITY UNPINNED like
the rest of the oracle: neither is OpenCV itself.
"""
import numpy as np

# the 8-neighbourhood in CLOCKWISE order (image coordinates: i grows downwards), starting east
_CW = [(0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1)]
_IDX = {d: k for k, d in enumerate(_CW)}


def outermost_borders(binary_map):
    """All outermost outer borders of a {0, !=0} map, each as the list of (x, y) = (j, i) pixels in the order the paper's step (3) visits
    them (the starting pixel first; a pixel that the border passes twice appears twice), borders in raster order of their starting pixels."""
    img = np.asarray(binary_map)
    h, w = img.shape[:2]
    f = np.zeros((h + 2, w + 2), np.int32)                   # the paper assumes a frame of 0-pixels
    f[1:-1, 1:-1] = (img.reshape(h, w) != 0)
    borders = []
    for i in range(1, h + 1):
        lnbd = 0                                             # Appendix II: reset at the start of every row
        for j in range(1, w + 1):
            fij = f[i, j]
            if fij == 0:
                continue
            if fij == 1 and f[i, j - 1] == 0 and lnbd <= 0:  # step (1)(a), only when the last border met on this row was left behind
                pts = _follow(f, i, j, i, j - 1)
                borders.append([(x - 1, y - 1) for (y, x) in pts])
            # step (4), Appendix II form: LNBD takes the (signed) mark of the pixel just passed
            if f[i, j] != 1:
                lnbd = f[i, j]
    return borders


def _follow(f, i, j, i2, j2):
    """step (3) with NBD = 2; returns the visited pixels (row, col) in order"""
    nbd = 2
    # (3.1) clockwise around (i, j), starting from (i2, j2)
    k0 = _IDX[(i2 - i, j2 - j)]
    first = None
    for t in range(8):
        di, dj = _CW[(k0 + t) % 8]
        if f[i + di, j + dj] != 0:
            first = (i + di, j + dj)
            break
    if first is None:
        f[i, j] = -nbd
        return [(i, j)]
    i1, j1 = first
    i2, j2 = i1, j1                                          # (3.2)
    i3, j3 = i, j
    out = []
    while True:
        # (3.3) counter-clockwise around (i3, j3), starting from the element after (i2, j2)
        k = _IDX[(i2 - i3, j2 - j3)]
        examined_east_zero = False
        found = None
        for t in range(1, 9):
            di, dj = _CW[(k - t) % 8]
            if f[i3 + di, j3 + dj] != 0:
                found = (i3 + di, j3 + dj)
                break
            if (di, dj) == (0, 1):
                examined_east_zero = True
        i4, j4 = found
        # (3.4)
        if examined_east_zero:
            f[i3, j3] = -nbd
        elif f[i3, j3] == 1:
            f[i3, j3] = nbd
        out.append((i3, j3))
        # (3.5)
        if (i4, j4) == (i, j) and (i3, j3) == (i1, j1):
            return out
        i2, j2 = i3, j3
        i3, j3 = i4, j4


def approx_simple(points):
    """CHAIN_APPROX_SIMPLE as the documentation states it ("compresses horizontal, vertical, and diagonal segments and leaves only their
    end points"): of the closed chain, a pixel stays iff the step into it and the step out of it differ."""
    n = len(points)
    if n <= 1:
        return list(points)
    keep = []
    for k in range(n):
        px, py = points[k - 1]
        cx, cy = points[k]
        nx, ny = points[(k + 1) % n]
        if (cx - px, cy - py) != (nx - cx, ny - cy):
            keep.append((cx, cy))
    return keep


def shoelace_area(points):
    """|signed area| of the closed polygon through the points (Green's theorem), exact in integers / 2"""
    n = len(points)
    s = 0
    for k in range(n):
        x0, y0 = points[k]
        x1, y1 = points[(k + 1) % n]
        s += x0 * y1 - x1 * y0
    return abs(s) / 2.0
