"""Oracle #2: torch-CPU restatement of the ubdvss network, loss and Adam step.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (no Keras /
TF here).  Independent of oracle/net_numpy.py (F.conv2d vs shifted matmuls); the
two are cross-checked in tests/test_oracle_net.py.  Also the CPU baseline
("port") timed by bench.py beside the GPU numbers, as BASELINE.md section 3
prescribes (oneDNN convolutions, all host threads).

Follows semantic_segmentation/net.py:225-252, :278-314; losses.py:13-17, :27-30,
:33-44, :47-62, :65-83, :86-126; train.py:110 (Keras Adam defaults).
"""
import numpy as np
import torch
import torch.nn.functional as F

DILATIONS = (1, 2, 4, 8, 16, 1)

# losses.py:13-17
L_POSITIVE_WEIGHT = 15.0
L_NEGATIVE_WEIGHT = 1.0
L_HARD_NEGATIVE_WEIGHT = 5.0
L_DETECTION_WEIGHT = 1.0
L_CLASSIFICATION_WEIGHT = 1.0
KERAS_EPS = 1e-7
# Keras evaluates the clip in fp32: p in [1e-7f, 1-1e-7f] -> logit in [-16.118095, +15.942385]
_E = np.float32(KERAS_EPS)
LOGIT_LO_F32 = float(np.log(_E / (np.float32(1) - _E)))
LOGIT_HI_F32 = float(np.log((np.float32(1) - _E) / (np.float32(1) - (np.float32(1) - _E))))


def to_torch_weights(weights, dtype=torch.float32, requires_grad=False):
    """Keras-ordered numpy list -> list of torch tensors in conv2d layout (OIHW)."""
    out = []
    for w in weights:
        t = torch.as_tensor(np.asarray(w), dtype=dtype)
        if t.ndim == 4:
            t = t.permute(3, 2, 0, 1).contiguous()      # HWIO -> OIHW
        out.append(t.clone().requires_grad_(requires_grad))
    return out


def _grad_round(t, grad_dtype):
    """Identity in the forward pass; the gradient that flows back through `t` is rounded to `grad_dtype` (models a
    gradient tensor stored in 16 bit by the backward kernels)."""
    if grad_dtype is None or not t.requires_grad:
        return t
    dt = _ACT_DTYPES[grad_dtype]
    t.register_hook(lambda g: g.to(dt).to(g.dtype))
    return t


def _sep(x, dw, pw, b, stride, fml, z_grad_dtype=None, dw_grad_dtype=None, act_dtype=None):
    """z_grad_dtype rounds the gradient w.r.t. the pre-activation, dw_grad_dtype the gradient w.r.t. the depthwise
    output (the two gradient tensors the bf16 backward kernels store for a separable layer); act_dtype: depthwise /
    pointwise kernels and the depthwise output are used in that 16-bit type (straight-through)."""
    c = x.shape[1]
    dw, pw = _ste_round(dw, act_dtype), _ste_round(pw, act_dtype)
    dwk = dw.permute(1, 0, 2, 3)                          # (1,C,3,3) -> (C,1,3,3)
    if stride == 2:
        if fml:
            x = F.pad(x, (1, 0, 1, 0))                    # left 1, top 1 (net.py:231)
        else:
            x = F.pad(x, (0, 1, 0, 1))                    # TF SAME s2 on even sizes
        x = F.conv2d(x, dwk, None, stride=2, padding=0, groups=c)
    else:
        x = F.conv2d(x, dwk, None, stride=1, padding=1, groups=c)
    x = _grad_round(_ste_round(x, act_dtype), dw_grad_dtype)
    return F.relu(_grad_round(F.conv2d(x, pw, b), z_grad_dtype))


_ACT_DTYPES = {"bfloat16": torch.bfloat16, "float16": torch.float16}


def _ste_round(t, act_dtype):
    """Round to a 16-bit storage type with a straight-through gradient (the 16-bit train step stores rounded
    activations / multiplies with a rounded kernel copy, and back-propagates as if the rounding were identity)."""
    if act_dtype is None:
        return t
    return t + (t.detach().to(_ACT_DTYPES[act_dtype]).to(t.dtype) - t.detach())


def forward(x_nhwc, tw, fml_compatible=True, act_dtype=None, grad_dtype=None):
    """x_nhwc: torch (N,H,W,C).  tw: list from to_torch_weights.  Returns NHWC logits.
    act_dtype "bfloat16"/"float16": hidden activations (including the depthwise output of a separable layer) and all
    3x3 / depthwise / pointwise kernels are rounded to that type (straight-through in the backward pass) --
    BASELINE.json configs[2..4].
    grad_dtype: the gradient tensors of the bf16 train step are rounded to that type: the gradient w.r.t. every
    pre-activation (L1..L9; those of L1 and L2 only ever exist tile-wise in LDS) and w.r.t. the depthwise output of
    L2 and L3."""
    x = x_nhwc.permute(0, 3, 1, 2)
    i = 0
    for li, stride in enumerate((2, 1, 2)):
        x = _ste_round(_sep(x, tw[i], tw[i + 1], tw[i + 2], stride, fml_compatible, grad_dtype,
                            grad_dtype if li >= 1 else None, act_dtype), act_dtype)
        i += 3
    for d in DILATIONS:
        z = _grad_round(F.conv2d(x, _ste_round(tw[i], act_dtype), tw[i + 1], padding=d, dilation=d), grad_dtype)
        x = _ste_round(F.relu(z), act_dtype)
        i += 2
    x = F.conv2d(x, tw[i], tw[i + 1])
    return x.permute(0, 2, 3, 1)


def forward_numpy(x, weights, fml_compatible=True, dtype=torch.float32):
    with torch.no_grad():
        tw = to_torch_weights(weights, dtype)
        y = forward(torch.as_tensor(np.asarray(x), dtype=dtype), tw, fml_compatible)
    return y.numpy()


# ----------------------------------------------------------------------------- loss
def detection_loss(y_true, y_pred, lo=LOGIT_LO_F32, hi=LOGIT_HI_F32):
    """losses.py:33-44 + :86-126 with K.binary_crossentropy's clip restated as the
    exact logit clamp: x' = clamp(x, logit(eps), logit(1-eps)),
    zero gradient where the clamp is active."""
    x = y_pred[..., 0]
    z = (y_true[..., 0] > 0).to(x.dtype)
    xc = torch.clamp(x, lo, hi)
    ce = torch.clamp(xc, min=0) - xc * z + torch.log1p(torch.exp(-xc.abs()))
    n_pos = torch.clamp(z.sum(), min=1)
    n_neg = torch.clamp((1 - z).sum(), min=1)
    pos = (ce * z).sum() / n_pos
    ce_neg = ce * (1 - z)
    neg = ce_neg.sum() / n_neg
    k = int(torch.minimum(n_pos, n_neg).item())
    top, _ = torch.topk(ce_neg.reshape(-1), k, sorted=False)
    hard = top.mean()
    if torch.isnan(hard):
        hard = torch.zeros_like(hard)
    return L_POSITIVE_WEIGHT * pos + L_NEGATIVE_WEIGHT * neg + L_HARD_NEGATIVE_WEIGHT * hard


def classification_loss(y_true, y_pred):
    """losses.py:65-83."""
    m = (y_true[..., 0] > 0)
    labels = ((y_true[..., 0] - 1) * m).long()
    logits = y_pred[..., 1:]
    ce = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1), reduction="none")
    mf = m.reshape(-1).to(y_pred.dtype)
    return (ce * mf).sum() / torch.clamp(mf.sum(), min=1)


def total_loss(y_true, y_pred, classification_mode):
    """losses.py:20-24, :47-62."""
    det = detection_loss(y_true, y_pred)
    if not classification_mode:
        return det
    return L_DETECTION_WEIGHT * det + L_CLASSIFICATION_WEIGHT * classification_loss(y_true, y_pred)


def from_torch_grads(tw):
    """grads of to_torch_weights tensors -> Keras-ordered numpy list (HWIO)."""
    out = []
    for t in tw:
        g = t.grad
        if g.ndim == 4:
            g = g.permute(2, 3, 1, 0)
        out.append(g.contiguous().numpy().copy())
    return out


def loss_and_grads(x, y_true, weights, classification_mode, fml_compatible=True, dtype=torch.float64, act_dtype=None,
                   grad_dtype=None):
    """Returns (loss, logits, dlogits, [grads in Keras order])."""
    tw = to_torch_weights(weights, dtype, requires_grad=True)
    xt = torch.as_tensor(np.asarray(x), dtype=dtype)
    yt = torch.as_tensor(np.asarray(y_true), dtype=dtype)
    logits = forward(xt, tw, fml_compatible, act_dtype, grad_dtype)
    logits.retain_grad()
    loss = total_loss(yt, logits, classification_mode)
    loss.backward()
    return (float(loss.detach()), logits.detach().numpy(), logits.grad.numpy().copy(), from_torch_grads(tw))


def adam_step(params, grads, m, v, t, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7):
    """Keras 2.2 Adam (train.py:110): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    p -= lr_t * m / (sqrt(v) + eps).  Flat numpy arrays, t starts at 1."""
    lr_t = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m = beta1 * m + (1 - beta1) * grads
    v = beta2 * v + (1 - beta2) * grads * grads
    params = params - lr_t * m / (np.sqrt(v) + eps)
    return params, m, v
