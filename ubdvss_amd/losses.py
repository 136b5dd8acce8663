"""Training objectives -- host mirror of semantic_segmentation/losses.py:13-24.

``get_loss(classification_mode)`` returns a callable ``f(y_true, y_pred) -> scalar`` like the
reference; it evaluates the fused HIP loss kernel (weighted sigmoid-BCE with batch-global
hard-negative top-k, plus masked softmax-CE in classification mode) on the MI355X.
``loss_and_grad`` additionally returns d loss / d y_pred (what TF autodiff would produce).
"""
import ctypes

import numpy as np
import torch

from . import _lib

# losses.py:13-17
L_POSITIVE_WEIGHT = 15.
L_NEGATIVE_WEIGHT = 1.
L_HARD_NEGATIVE_WEIGHT = 5.
L_DETECTION_WEIGHT = 1.
L_CLASSIFICATION_WEIGHT = 1.

_handles = {}
_workspaces = _lib.StreamWorkspaces()   # (device, stream) -> uint8 scratch reused between calls on that stream (grown on demand, LRU-bounded)


def _handle(n_classes, device):
    key = (n_classes, str(device))
    if key not in _handles:
        lib = _lib.load()
        cfg = _lib.UbdConfig(1, n_classes, 1, _lib.UBD_F32)
        h = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.ubd_create(ctypes.byref(cfg), ctypes.byref(h)), "ubd_create")
        _handles[key] = h
    return _handles[key]


def _as_device(a, dtype, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a))).to(device=device, dtype=dtype)


def loss_and_grad(y_true, y_pred, want_grad=True):
    """y_true (N,h,w,1) or (N,h,w) labels 0..n_cls; y_pred (N,h,w,1+n_cls) logits.
    Returns (loss4 device tensor [total, detection, classification, k], dlogits or None)."""
    if not torch.cuda.is_available():
        raise RuntimeError("ubdvss_amd.losses needs an MI355X; there is no CPU fallback")
    lib = _lib.load()
    device = y_pred.device if isinstance(y_pred, torch.Tensor) and y_pred.is_cuda else torch.device(
        f"cuda:{torch.cuda.current_device()}")
    yp = _as_device(y_pred, torch.float32, device)
    n, h, w, k = yp.shape
    yt = _as_device(y_true, torch.int32, device).reshape(n, h, w)
    hd = _handle(k - 1, device)
    loss = torch.zeros(16, dtype=torch.float32, device=device)
    grad = torch.empty_like(yp) if want_grad else None
    nbytes = int(lib.ubd_loss_workspace_bytes(hd, n, h, w))
    # one scratch per (device, stream): calls on one stream are ordered, two streams never share header / histograms
    raw_stream = torch.cuda.current_stream(device).cuda_stream
    ws = _workspaces.get(device, raw_stream, nbytes)
    stream = ctypes.c_void_p(raw_stream)
    _lib.check(lib.ubd_loss(hd, yp.data_ptr(), yt.data_ptr(), n, h, w, loss.data_ptr(),
                            grad.data_ptr() if grad is not None else None, ws.data_ptr(), ws.numel(), stream), "ubd_loss")
    return loss, grad


def detection_loss(y_true, y_pred):
    """losses.py:33-44 (only channel 0 of y_pred is used)."""
    yp = y_pred[..., :1]
    return loss_and_grad(y_true, yp, want_grad=False)[0][1]


def detection_and_classification_loss(y_true, y_pred):
    """losses.py:47-62."""
    return loss_and_grad(y_true, y_pred, want_grad=False)[0][0]


def get_loss(classification_mode=False):
    """losses.py:20-24."""
    if classification_mode:
        return detection_and_classification_loss
    return detection_loss
