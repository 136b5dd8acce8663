"""Oracle: numpy restatement of the ubdvss dilated FCN forward pass.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: Keras/TF
cannot run in the build container; this follows the reference source line by
line and is cross-checked against oracle/net_torch.py.

Follows (all paths relative to the reference root):
  semantic_segmentation/net.py:225-252   conv_bn (ZeroPadding2D((1,0),(1,0)) +
                                         'valid' when fml-compatible stride 2,
                                         SeparableConv2D / Conv2D, bias, relu)
  semantic_segmentation/net.py:278-314   _build_dilated_conv_model (3 separable
                                         stem layers s=2,1,2; six dense 3x3 convs
                                         with dilation 1,2,4,8,16,1; 1x1 head)
  semantic_segmentation/net.py:217-218   preprocess_image_mobilenet

Layout: activations NHWC, kernels in Keras shapes (HWIO; depthwise (3,3,C,1)),
weight list in ``model.get_weights()`` order.
"""
import numpy as np

N_FILTERS = 24
DILATIONS = (1, 2, 4, 8, 16, 1)


def weight_shapes(c_in=3, n_classes=0):
    """Keras ``get_weights()`` order: [(name, shape), ...]  (net.py:292-311)."""
    k_out = 1 + n_classes
    shapes = []
    cin = c_in
    for li in (1, 2, 3):
        shapes += [(f"l{li}.dw", (3, 3, cin, 1)), (f"l{li}.pw", (1, 1, cin, N_FILTERS)),
                   (f"l{li}.b", (N_FILTERS,))]
        cin = N_FILTERS
    for li in range(4, 10):
        shapes += [(f"l{li}.k", (3, 3, N_FILTERS, N_FILTERS)), (f"l{li}.b", (N_FILTERS,))]
    shapes += [("head.k", (1, 1, N_FILTERS, k_out)), ("head.b", (k_out,))]
    return shapes


def n_params(c_in=3, n_classes=0):
    return int(sum(np.prod(s) for _, s in weight_shapes(c_in, n_classes)))


def init_weights(seed, c_in=3, n_classes=0, bias_scale=0.0, dtype=np.float32):
    """glorot_uniform kernels (Keras fans: fan_in=kh*kw*in, fan_out=kh*kw*out),
    zero biases (Keras defaults, net.py:226,245).  ``bias_scale`` > 0 draws
    U(-s, s) biases instead so that tests exercise the bias path.
    Reproduces the distribution, not Keras's RNG stream."""
    rng = np.random.default_rng(seed)
    out = []
    for name, shape in weight_shapes(c_in, n_classes):
        if len(shape) == 1:
            w = rng.uniform(-bias_scale, bias_scale, shape) if bias_scale > 0 else np.zeros(shape)
        else:
            kh, kw, cin, cout = shape
            if name.endswith(".dw"):      # depthwise kernel (3,3,C,1): Keras fans on the 4-d shape
                fan_in, fan_out = kh * kw * cin, kh * kw * cout
            else:
                fan_in, fan_out = kh * kw * cin, kh * kw * cout
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            w = rng.uniform(-lim, lim, shape)
        out.append(w.astype(dtype))
    return out


def flatten_weights(weights):
    return np.concatenate([np.asarray(w).reshape(-1) for w in weights])


def unflatten_weights(flat, c_in=3, n_classes=0):
    out, off = [], 0
    for _, shape in weight_shapes(c_in, n_classes):
        n = int(np.prod(shape))
        out.append(np.asarray(flat[off:off + n]).reshape(shape))
        off += n
    assert off == len(flat), (off, len(flat))
    return out


def preprocess_mobilenet(image):
    """net.py:217-218."""
    return (image - 127.5) / 127.5


def _pad_for(x, stride, dilation, fml_compatible):
    """Zero padding of conv_bn (net.py:229-232 + Keras 'same'/'valid' rules)."""
    if stride == 2:
        if fml_compatible:          # ZeroPadding2D(((1,0),(1,0))) then 'valid'
            return np.pad(x, ((0, 0), (1, 0), (1, 0), (0, 0)))
        # TF 'SAME', stride 2, k=3: pad_total = max((ceil(H/2)-1)*2+3-H, 0), extra at the end
        n, h, w, c = x.shape
        ph = max((-(-h // 2) - 1) * 2 + 3 - h, 0)
        pw = max((-(-w // 2) - 1) * 2 + 3 - w, 0)
        return np.pad(x, ((0, 0), (ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2), (0, 0)))
    d = dilation
    return np.pad(x, ((0, 0), (d, d), (d, d), (0, 0)))


def _taps(xp, stride, dilation, out_h, out_w):
    for ky in range(3):
        for kx in range(3):
            y0, x0 = ky * dilation, kx * dilation
            yield ky, kx, xp[:, y0:y0 + (out_h - 1) * stride + 1:stride,
                             x0:x0 + (out_w - 1) * stride + 1:stride, :]


def _out_hw(x, xp, stride, dilation):
    span = 2 * dilation + 1
    return (xp.shape[1] - span) // stride + 1, (xp.shape[2] - span) // stride + 1


def separable_conv(x, dw, pw, b, stride, fml_compatible=True, relu=True, act_dtype=None):
    """SeparableConv2D: depthwise 3x3 (multiplier 1, no bias/activation) ->
    pointwise 1x1 + bias + relu (net.py:234-246).  act_dtype: the depthwise output is stored in that 16-bit type
    before the pointwise product (16-bit configs)."""
    xp = _pad_for(x, stride, 1, fml_compatible)
    oh, ow = _out_hw(x, xp, stride, 1)
    acc = np.zeros((x.shape[0], oh, ow, x.shape[3]), dtype=x.dtype)
    for ky, kx, xs in _taps(xp, stride, 1, oh, ow):
        acc = acc + xs * dw[ky, kx, :, 0].astype(x.dtype)
    acc = round_to(acc, act_dtype)
    y = acc @ pw[0, 0].astype(x.dtype) + b.astype(x.dtype)
    return np.maximum(y, 0) if relu else y


def dilated_conv(x, k, b, dilation, relu=True):
    """Conv2D 3x3 'same', dilation d: cross-correlation, zero pad d per side."""
    xp = _pad_for(x, 1, dilation, True)
    oh, ow = x.shape[1], x.shape[2]
    acc = np.zeros((x.shape[0], oh, ow, k.shape[3]), dtype=x.dtype)
    for ky, kx, xs in _taps(xp, 1, dilation, oh, ow):
        acc = acc + xs @ k[ky, kx].astype(x.dtype)
    y = acc + b.astype(x.dtype)
    return np.maximum(y, 0) if relu else y


def round_to(a, act_dtype):
    """Round an fp64/fp32 array to bfloat16 (round-to-nearest-even on the fp32 bit pattern) or float16 and
    return it widened again -- models 16-bit activation storage (BASELINE.json configs[2..4])."""
    if act_dtype is None:
        return a
    a32 = np.asarray(a, dtype=np.float32)
    if act_dtype == "float16":
        return a32.astype(np.float16).astype(a.dtype)
    if act_dtype == "bfloat16":
        u = a32.view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32).astype(a.dtype)
    raise ValueError(act_dtype)


def forward(x, weights, fml_compatible=True, dtype=np.float64, return_all=False, act_dtype=None):
    """x: (N,H,W,C_in) already preprocessed.  Returns logits (N,H/4,W/4,1+n_cls);
    with return_all also the list of the 9 hidden activations.
    act_dtype "bfloat16"/"float16" (the 16-bit configs of BASELINE.json): every hidden activation -- including the
    depthwise output inside a separable layer -- is rounded to that type when it is stored, and every 3x3 / depthwise /
    pointwise kernel of L1..L9 is used in that type; biases, the 1x1 head and all accumulation stay in `dtype`, logits
    are not rounded."""
    x = np.asarray(x, dtype=dtype)
    w = [np.asarray(a, dtype=dtype) for a in weights]
    acts = []
    i = 0
    for stride in (2, 1, 2):
        x = round_to(separable_conv(x, round_to(w[i], act_dtype), round_to(w[i + 1], act_dtype), w[i + 2], stride,
                                    fml_compatible, act_dtype=act_dtype), act_dtype)
        acts.append(x)
        i += 3
    for d in DILATIONS:
        x = round_to(dilated_conv(x, round_to(w[i], act_dtype), w[i + 1], d), act_dtype)
        acts.append(x)
        i += 2
    logits = x @ w[i][0, 0] + w[i + 1]
    return (logits, acts) if return_all else logits


def forward_bn(x, weights_bn, fml_compatible=True, eps=1e-3, dtype=np.float64):
    """The use_bn=True branch of conv_bn (net.py:248-250; never instantiated by the reference's own model builder, net.py:292-304
    passes no use_bn): conv with bias and NO activation, keras BatchNormalization in inference mode (defaults: epsilon 1e-3,
    gamma / beta / moving mean / moving variance per output channel), then ReLU.  weights_bn: per hidden layer its conv arrays
    followed by [gamma, beta, moving_mean, moving_variance] (the get_weights() order of such a model), head last."""
    x = np.asarray(x, dtype=dtype)
    w = [np.asarray(a, dtype=dtype) for a in weights_bn]

    def bn_relu(z, g, b, m, v):
        return np.maximum((z - m) / np.sqrt(v + eps) * g + b, 0.0)
    i = 0
    for stride in (2, 1, 2):
        z = separable_conv(x, w[i], w[i + 1], w[i + 2], stride, fml_compatible, relu=False)
        x = bn_relu(z, *w[i + 3:i + 7])
        i += 7
    for d in DILATIONS:
        z = dilated_conv(x, w[i], w[i + 1], d, relu=False)
        x = bn_relu(z, *w[i + 2:i + 6])
        i += 6
    return x @ w[i][0, 0] + w[i + 1]


def logit_threshold(pixel_threshold=0.5):
    """model_runner.py:37-38."""
    eps = 1e-9
    return -np.log(1 / np.clip(pixel_threshold, eps, 1 - eps) - 1)


def binary_map(logits, pixel_threshold=0.5):
    """model_runner.py:121-124: np.where(logit0 > thr, 1, 0) (strict >)."""
    return np.where(logits[..., :1] > logit_threshold(pixel_threshold), 1, 0)
