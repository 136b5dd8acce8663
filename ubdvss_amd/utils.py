"""Small host helpers on box lists (behaviour of semantic_segmentation/utils.py:67-69 and :135-138)."""
import numpy as np


def rescale_bbox(bbox, xscale, yscale):
    """Scale the x coordinates (even positions) by ``xscale`` and the y coordinates (odd positions) by ``yscale``,
    then truncate toward zero exactly like ``ndarray.astype(int)`` does in the reference."""
    pts = np.asarray(bbox, dtype=np.float64).reshape(-1, 2) * np.array([xscale, yscale], dtype=np.float64)
    return pts.reshape(-1).astype(int)


def np_softmax(logits, axis=-1):
    """Numerically stable softmax (maximum subtracted first)."""
    z = np.asarray(logits) - np.max(logits, axis=axis, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=axis, keepdims=True)
