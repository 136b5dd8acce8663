"""Per-batch pixel metrics of the train step -- host mirror of semantic_segmentation/keras_metrics.py:110-191.

The Keras graph evaluates these as extra outputs of every training step; here the fused loss kernel
(csrc/loss.hip) returns the raw counters with the loss (``loss[4:14]``) and the formulas below reproduce the
reference's definitions: accuracy = sum(equal) / max(1, N) (:33-36), precision = tp / max(1, tp + fp) (:66-76),
recall = tp / max(1, tp + fn) (:79-89), f1 = 2PR/(P+R) or 0 (:92-107), classification accuracy over positive
pixels (:160-172), and the loss components of losses.py:138-191.
"""


def metrics_from_loss_vector(loss16, classification_mode=False):
    """loss16: the 16-float vector written by ubd_loss / ubd_train_step (host list / numpy / tensor)."""
    v = [float(x) for x in loss16]
    total, det, cls, _k, pos_l, neg_l, hard_l, n_pos, tp, tn, fp, fn, cls_ok, n_pix = v[:14]
    precision = tp / max(1.0, tp + fp)
    recall = tp / max(1.0, tp + fn)
    out = {
        "loss": total,
        "detection_pixel_acc": (tp + tn) / max(1.0, n_pix),
        "detection_pixel_precision": precision,
        "detection_pixel_recall": recall,
        "detection_pixel_f1": 2.0 * precision * recall / (precision + recall) if precision + recall != 0 else 0.0,
        "positive_loss": pos_l, "negative_loss": neg_l, "hard_negative_loss": hard_l,
    }
    if classification_mode:
        out["classification_pixel_acc"] = cls_ok / max(1.0, n_pos)
        out["detection_loss"] = det
        out["classification_loss"] = cls
    return out


def get_all_metrics(classification_mode=False):
    """Names in the order Keras reports them (keras_metrics.py:175-191)."""
    names = ["detection_pixel_acc", "detection_pixel_precision", "detection_pixel_recall", "detection_pixel_f1"]
    if classification_mode:
        names.append("classification_pixel_acc")
    names += ["positive_loss", "negative_loss", "hard_negative_loss"]
    if classification_mode:
        names += ["detection_loss", "classification_loss"]
    return names
