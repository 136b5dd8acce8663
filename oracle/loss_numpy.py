"""Oracle: numpy fp64 restatement of the ubdvss training losses with analytic
gradients w.r.t. the logits.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (Keras/TF
absent); cross-checked against torch autograd (oracle/net_torch.py).

Follows semantic_segmentation/losses.py:
  :13-17   weights 15 / 1 / 5, detection 1, classification 1
  :27-30   z = (y_true > 0), p = sigmoid(y_pred[..., :1])
  :86-126  binary_classification_loss: K.binary_crossentropy (Keras clips p to
           [1e-7, 1-1e-7] and goes back to logits => logit clamp), positive mean,
           negative mean, mean of top-k masked negatives with
           k = min(max(n_pos,1), max(n_neg,1)) over the flattened whole batch,
           NaN -> 0
  :65-83   classification_loss: masked sparse softmax CE / max(n_pos, 1)
  :47-62   total = 1*det + 1*cls
Tie rule at the k-th value follows tf.nn.top_k: lower flat index first.
"""
import numpy as np

L_POS, L_NEG, L_HARD = 15.0, 1.0, 5.0
L_DET, L_CLS = 1.0, 1.0
EPS = 1e-7
LOGIT_LO = float(np.log(EPS / (1 - EPS)))       # -16.118095...
LOGIT_HI = float(np.log((1 - EPS) / EPS))       # exact-arithmetic clip point (see note)
# Note: in fp32 Keras evaluates log(p/(1-p)) at p = float32(1-1e-7) = 1-1.1920929e-07,
# giving +15.942385; SURVEY.md section 9.3 records both.  The oracle exposes the
# clip points as parameters; the product uses the fp32 values by default.
LOGIT_LO_F32 = float(np.log(np.float32(EPS) / (np.float32(1) - np.float32(EPS))))
LOGIT_HI_F32 = float(np.log((np.float32(1) - np.float32(EPS)) / (np.float32(1) - (np.float32(1) - np.float32(EPS)))))


def _bce_from_logits(x, z, lo, hi):
    xc = np.clip(x, lo, hi)
    ce = np.maximum(xc, 0) - xc * z + np.log1p(np.exp(-np.abs(xc)))
    inside = (x >= lo) & (x <= hi)
    return xc, ce, inside


def topk_mask(values, k):
    """Boolean mask of the k largest entries of a flat array; ties at the k-th
    value resolved toward the lower flat index (tf.nn.top_k)."""
    n = values.size
    order = np.lexsort((np.arange(n), -values))      # primary: value desc, secondary: index asc
    mask = np.zeros(n, dtype=bool)
    mask[order[:k]] = True
    return mask


def detection_loss(y_true, y_pred, lo=LOGIT_LO_F32, hi=LOGIT_HI_F32, return_parts=False):
    """Returns (loss, dloss/dlogit0 with the shape of y_pred[..., 0])."""
    x = np.asarray(y_pred, dtype=np.float64)[..., 0]
    z = (np.asarray(y_true)[..., 0] > 0).astype(np.float64)
    xc, ce, inside = _bce_from_logits(x, z, lo, hi)
    n_pos = max(z.sum(), 1.0)
    n_neg = max((1 - z).sum(), 1.0)
    pos = (ce * z).sum() / n_pos
    ce_neg = ce * (1 - z)
    neg = ce_neg.sum() / n_neg
    k = int(min(n_pos, n_neg))
    sel = topk_mask(ce_neg.reshape(-1), k).reshape(x.shape)
    hard = ce_neg[sel].mean()
    if np.isnan(hard):
        hard = 0.0
    loss = L_POS * pos + L_NEG * neg + L_HARD * hard
    sig = 1.0 / (1.0 + np.exp(-xc))
    coef = L_POS * z / n_pos + L_NEG * (1 - z) / n_neg + L_HARD * (1 - z) * sel / k
    grad = (sig - z) * coef * inside
    if return_parts:
        return loss, grad, dict(pos=pos, neg=neg, hard=hard, n_pos=n_pos, n_neg=n_neg, k=k, sel=sel)
    return loss, grad


def classification_loss(y_true, y_pred):
    """Returns (loss, dloss/dlogits[..., 1:])."""
    yt = np.asarray(y_true)[..., 0]
    m = (yt > 0)
    labels = ((yt - 1) * m).astype(np.int64)
    logits = np.asarray(y_pred, dtype=np.float64)[..., 1:]
    mx = logits.max(axis=-1, keepdims=True)
    e = np.exp(logits - mx)
    s = e.sum(axis=-1, keepdims=True)
    logp = logits - mx - np.log(s)
    ce = -np.take_along_axis(logp, labels[..., None], axis=-1)[..., 0]
    denom = max(float(m.sum()), 1.0)
    loss = (ce * m).sum() / denom
    onehot = np.zeros_like(logits)
    np.put_along_axis(onehot, labels[..., None], 1.0, axis=-1)
    grad = (e / s - onehot) * m[..., None] / denom
    return loss, grad


def total_loss(y_true, y_pred, classification_mode, lo=LOGIT_LO_F32, hi=LOGIT_HI_F32):
    """Returns (loss, dloss/dy_pred) -- losses.py:20-24."""
    y_pred = np.asarray(y_pred, dtype=np.float64)
    det, gdet = detection_loss(y_true, y_pred, lo, hi)
    g = np.zeros_like(y_pred)
    g[..., 0] = L_DET * gdet if classification_mode else gdet
    if not classification_mode:
        return det, g
    cls, gcls = classification_loss(y_true, y_pred)
    g[..., 1:] = L_CLS * gcls
    return L_DET * det + L_CLS * cls, g


def batch_metrics(y_true, y_pred, classification_mode):
    """keras_metrics.py:110-172 on one batch (numpy restatement): detection pixel accuracy / precision / recall /
    f1 with pred = logit0 > 0, classification accuracy over positive pixels."""
    yt = np.asarray(y_true)[..., 0]
    t = (yt > 0).astype(np.int64)
    p = (np.asarray(y_pred)[..., 0] > 0).astype(np.int64)
    tp = float(((t == p) & (t == 1)).sum()); tn = float(((t == p) & (t == 0)).sum())
    fp = float(((t != p) & (p == 1)).sum()); fn = float(((t != p) & (p == 0)).sum())
    prec = tp / max(1.0, tp + fp); rec = tp / max(1.0, tp + fn)
    out = {"detection_pixel_acc": (tp + tn) / max(1.0, t.size), "detection_pixel_precision": prec,
           "detection_pixel_recall": rec, "detection_pixel_f1": 2 * prec * rec / (prec + rec) if prec + rec != 0 else 0.0}
    if classification_mode:
        labels = (yt - 1) * t
        pred_cls = np.argmax(np.asarray(y_pred)[..., 1:], axis=-1)
        out["classification_pixel_acc"] = float(((labels == pred_cls) * t).sum()) / max(1.0, float(t.sum()))
    return out
