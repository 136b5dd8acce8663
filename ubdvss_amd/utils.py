"""Host helpers on box lists -- mirrors semantic_segmentation/utils.py:67-69, :135-138."""
import numpy as np


def rescale_bbox(bbox, xscale, yscale):
    """utils.py:67-69: multiply (x, y) pairs, truncate toward zero."""
    scale = np.array([xscale, yscale] * (len(bbox) // 2))
    return (bbox * scale).astype(int)


def np_softmax(logits, axis=-1):
    """utils.py:135-138."""
    x = logits - np.max(logits, axis=axis, keepdims=True)
    x = np.exp(x)
    return x / np.sum(x, axis=axis, keepdims=True)
