"""Label maps for random-shape soaks of the train step against the oracle: mostly POSITIVE (ones -- or class bands -- with one to three
rectangular holes of < 45 % of the area), so that k = min(n_pos, n_neg) = n_neg and the hard-negative term takes EVERY negative whatever
their order (losses.py:110-116).  With the usual sparse rectangles k < n_neg, the top-k choice is discontinuous, and a near-tie at the
k-th value makes kernel and oracle pick different pixels (round 3, and again in round 6 at 3 x 216 x 512 in fp32: head.b equal to 1e-8,
every other tensor off by 1 / k and more) -- a property of the objective, not a defect, but it hides defects in a soak."""
import numpy as np


def mostly_positive_maps(rng, n, mh, mw, n_classes=0):
    labels = np.ones((n, mh, mw), np.int32)
    if n_classes > 1:
        labels[:, :, mw // 2:] = 2
    for im in range(n):
        budget = int(0.45 * mh * mw)
        for _ in range(int(rng.integers(1, 4))):
            rh, rw = int(rng.integers(1, max(2, mh // 2))), int(rng.integers(1, max(2, mw // 2)))
            if rh * rw > budget:
                continue
            y0, x0 = int(rng.integers(0, mh - rh + 1)), int(rng.integers(0, mw - rw + 1))
            labels[im, y0:y0 + rh, x0:x0 + rw] = 0
            budget -= rh * rw
    labels[0, 0, 0] = 0                                           # at least one negative
    assert (labels > 0).sum() >= (labels == 0).sum()
    return labels
