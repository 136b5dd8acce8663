"""Network definition + config -- host mirror of semantic_segmentation/net.py.

``NetConfig`` keeps the reference's constructor and getters (net.py:73-214).
``NetManager.build_model`` (net.py:273-314) creates a ``Model`` whose ``predict`` replaces
``keras.Model.predict`` (called at model_runner.py:119 and predict.py:74-76) with the HIP
forward pass of libubd_hip.so.  Weights live in one flat fp32 device vector in Keras
``get_weights()`` order; ``get_weights`` / ``set_weights`` use the Keras shapes.
"""
import copy
import ctypes
import logging
import os
import pickle
from enum import Enum

import numpy as np
import torch

from . import _lib

N_FILTERS = 24                      # net.py:289
DILATIONS = (1, 2, 4, 8, 16, 1)     # net.py:298-304


class PreprocessingType(Enum):      # net.py:62-64
    NONE = 0
    MOBILENET_LIKE = 1


supported_preprocessing_types = {
    "none": PreprocessingType.NONE,
    "mobilenet_like": PreprocessingType.MOBILENET_LIKE,
}


def preprocess_image_mobilenet(image):      # net.py:217-218
    return (image - 127.5) / 127.5


def depreprocess_image_mobilenet(image):    # net.py:221-222
    return image * 127.5 + 127.5


class NetConfig:
    """System / network configuration with the reference's constructor arguments and getter names
    (semantic_segmentation/net.py:73-214), so that code written against the reference runs unchanged.
    Extra: ``class_names`` may be passed directly instead of a file (``object_types_fname``).

    Fields that reach the kernels: ``grey`` (1 or 3 input channels), ``fml_compatible`` (stride-2 padding rule),
    number of classes, ``scale`` (=4), ``min_pixels_for_detection`` (contourArea threshold), ``preprocessing``.
    """

    _OVERRIDABLE = {"side_multiple": "_side_multiple", "max_image_side": "_max_side",
                    "min_pixels_for_detection": "_min_pixels_for_detection"}

    def __init__(self, object_types_fname=None, scale=4, fml_compatible=True, no_classification=False,
                 side_multiple=64, max_image_side=512, min_pixels_for_detection=5,
                 preprocessing=PreprocessingType.NONE, grey=True, class_names=None):
        names = None
        if class_names is not None:
            names = [str(c) for c in class_names]
        elif object_types_fname is not None:
            names = self._load_class_names(object_types_fname)
        self._class_names = names
        self._class_name_to_id = {c: k for k, c in enumerate(names)} if names is not None else {}
        self._is_classification_supported = names is not None and not no_classification
        self._grey, self._scale, self._fml_compatible = grey, scale, fml_compatible
        self._preprocessing = preprocessing
        self._side_multiple, self._max_side = side_multiple, max_image_side
        self._min_pixels_for_detection = min_pixels_for_detection

    @staticmethod
    def from_others(base_config, side_multiple=None, max_image_side=None, min_pixels_for_detection=None):
        """Copy of ``base_config`` with the architecture-independent knobs replaced where a value is given."""
        clone = copy.deepcopy(base_config)
        given = {"side_multiple": side_multiple, "max_image_side": max_image_side,
                 "min_pixels_for_detection": min_pixels_for_detection}
        for key, value in given.items():
            if value:
                setattr(clone, NetConfig._OVERRIDABLE[key], value)
        return clone

    @staticmethod
    def _load_class_names(path):
        if not os.path.exists(path):
            raise AssertionError(f"File with object class names {path} does not exist")
        with open(path) as f:
            return [ln.strip() for ln in f if ln.strip()]

    # -- getters (names of the reference) ---------------------------------------------------------
    def is_grey(self): return self._grey
    def get_scale(self): return self._scale
    def get_min_pixels_for_detection(self): return self._min_pixels_for_detection
    def get_side_multiple(self): return self._side_multiple
    def get_max_side(self): return self._max_side
    def is_fml_compatible(self): return self._fml_compatible
    def get_preprocessing_type(self): return self._preprocessing
    def get_class_names(self): return self._class_names
    def get_n_classes(self): return len(self._class_names)
    def get_class_name(self, class_id): return self._class_names[class_id]
    def get_class_id(self, class_name): return self._class_name_to_id[class_name]
    def is_classification_supported(self): return self._is_classification_supported

    def is_class_supported(self, class_name):
        return self._class_names is None or class_name in self._class_name_to_id

    def _pre_pair(self):
        table = {PreprocessingType.NONE: (lambda x: x, lambda x: x),
                 PreprocessingType.MOBILENET_LIKE: (preprocess_image_mobilenet, depreprocess_image_mobilenet)}
        if self._preprocessing not in table:
            raise ValueError("Unknown preprocessing type")
        return table[self._preprocessing]

    def get_preprocessing_fn(self): return self._pre_pair()[0]
    def get_depreprocessing_fn(self): return self._pre_pair()[1]

    def log_classification_mode(self):
        if self._is_classification_supported:
            logging.info(f"Training classification with object types: {self._class_names}")
        elif self._class_names is not None:
            logging.info(f"Training WITHOUT classification, detection only for types: {self._class_names}")
        else:
            logging.info("Training WITHOUT classification, detection for any barcode in datasets")

    def __str__(self):
        rows = [f"\t{k[1:]}={v}" for k, v in vars(self).items() if k.startswith("_")]
        return "\n".join(["Net Config:"] + rows)


class _ReferenceUnpickler(pickle.Unpickler):
    """``config.pkl`` files written by the reference name ``semantic_segmentation.net.NetConfig`` /
    ``PreprocessingType`` (net.py:468-469); the attribute names are the same here, so those classes map onto this
    module's.  Nothing else of the reference package is resolvable (and nothing else is needed)."""

    def find_class(self, module, name):
        if module in ("semantic_segmentation.net", "net") and name in ("NetConfig", "PreprocessingType"):
            return globals()[name]
        return super().find_class(module, name)


def load_reference_pickle(path):
    with open(path, "rb") as f:
        cfg = _ReferenceUnpickler(f).load()
    if isinstance(cfg, NetConfig) and not hasattr(cfg, "_class_name_to_id"):   # the reference sets it only with a class file
        cfg._class_name_to_id = {}
    return cfg


def weight_shapes(c_in, n_classes):
    """Keras ``model.get_weights()`` order and shapes of the built model (net.py:292-311)."""
    shapes, cin = [], c_in
    for _ in range(3):
        shapes += [(3, 3, cin, 1), (1, 1, cin, N_FILTERS), (N_FILTERS,)]
        cin = N_FILTERS
    for _ in DILATIONS:
        shapes += [(3, 3, N_FILTERS, N_FILTERS), (N_FILTERS,)]
    shapes += [(1, 1, N_FILTERS, 1 + n_classes), (1 + n_classes,)]
    return shapes


def fold_batchnorm(weights_bn, eps=1e-3):
    """``get_weights()`` of a model built with ``conv_bn(..., use_bn=True)`` (net.py:225-252: conv with bias, no activation ->
    BatchNormalization -> ReLU; Keras default epsilon 1e-3) -> the 29 arrays of the BN-free architecture that computes the
    same function at inference time: with s = gamma / sqrt(moving_variance + eps) per output channel, the layer's last kernel
    (pointwise of a separable layer, the HWIO kernel of a Conv2D) is scaled by s along its output axis and the bias becomes
    (bias - moving_mean) * s + beta.  Per hidden layer the input holds its conv arrays followed by gamma, beta, moving_mean,
    moving_variance; the 1x1 head (no BN) comes last.  Inference only: the train step has no BatchNormalization."""
    w = [np.asarray(a, dtype=np.float64) for a in weights_bn]
    if len(w) != 3 * 7 + 6 * 6 + 2:
        raise ValueError(f"expected {3 * 7 + 6 * 6 + 2} arrays of a use_bn model, got {len(w)}")
    out, i = [], 0
    for n_conv in (3, 3, 3, 2, 2, 2, 2, 2, 2):
        conv, (gamma, beta, mean, var) = w[i:i + n_conv], w[i + n_conv:i + n_conv + 4]
        s = gamma / np.sqrt(var + eps)
        out += [a.astype(np.float32) for a in conv[:-2]]
        out.append((conv[-2] * s).astype(np.float32))       # output channels are the last axis of both kernel kinds
        out.append(((conv[-1] - mean) * s + beta).astype(np.float32))
        i += n_conv + 4
    out += [w[i].astype(np.float32), w[i + 1].astype(np.float32)]
    return out


_DTYPES = {"float32": _lib.UBD_F32, "bfloat16": _lib.UBD_BF16, "float16": _lib.UBD_F16}


class Model:
    """The dilated FCN on one MI355X.  ``predict`` has ``keras.Model.predict`` semantics
    (numpy NHWC in, numpy logits out); ``predict_on_device`` keeps everything in HBM."""

    def __init__(self, net_config, dtype="float32", device=None, seed=None):
        if not torch.cuda.is_available():
            raise RuntimeError("ubdvss_amd.Model needs an MI355X (torch.cuda unavailable); there is no CPU fallback")
        self._lib = _lib.load()
        self.net_config = net_config
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.c_in = 1 if net_config.is_grey() else 3
        self.n_classes = net_config.get_n_classes() if net_config.is_classification_supported() else 0
        self.k_out = 1 + self.n_classes
        self.dtype = dtype
        cfg = _lib.UbdConfig(self.c_in, self.n_classes, int(net_config.is_fml_compatible()), _DTYPES[dtype])
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ubd_create(ctypes.byref(cfg), ctypes.byref(handle)), "ubd_create")
        self._h = handle
        n = self._lib.ubd_param_count(self._h)
        self.params = torch.zeros(n, dtype=torch.float32, device=self.device)
        self._packed_key = None      # (workspace, params, versions) the packed fragments in the workspace belong to
        self._graphed = {}           # (n, H, W, dtype) -> GraphedForward
        self._weights_epoch = 0      # bumped by whoever writes self.params through a raw pointer (Trainer)
        self._ws = None
        self._pp_ws = None
        self.set_weights(self._glorot_init(seed))

    @property
    def num_cus(self):
        """Compute units the handle sizes its persistent grids for (the device's, or UBD_TEST_NUM_CUS at creation)."""
        return int(self._lib.ubd_num_cus(self._h))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.ubd_destroy(h)

    # ---------------------------------------------------------------- weights
    def _glorot_init(self, seed):
        """Keras defaults: glorot_uniform kernels, zero biases (net.py:226,245)."""
        rng = np.random.default_rng(seed)
        out = []
        for shape in weight_shapes(self.c_in, self.n_classes):
            if len(shape) == 1:
                out.append(np.zeros(shape, np.float32))
            else:
                kh, kw, cin, cout = shape
                lim = np.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
                out.append(rng.uniform(-lim, lim, shape).astype(np.float32))
        return out

    def invalidate_packed_weights(self):
        """Call after writing ``self.params`` through a raw pointer or a c10d collective (neither bumps the tensor's
        version counter): the next forward pass re-packs the MFMA weight fragments."""
        self._weights_epoch += 1
        self._packed_key = None

    def count_params(self):
        return int(self.params.numel())

    def get_weights(self):
        flat = self.params.detach().cpu().numpy()
        out, off = [], 0
        for shape in weight_shapes(self.c_in, self.n_classes):
            k = int(np.prod(shape))
            out.append(flat[off:off + k].reshape(shape).copy())
            off += k
        return out

    def set_weights(self, weights):
        shapes = weight_shapes(self.c_in, self.n_classes)
        if len(weights) == len(shapes) + 4 * 9:             # a use_bn=True model (net.py:248-250): inference-time fold
            weights = fold_batchnorm(weights)
        if len(weights) != len(shapes):
            raise ValueError(f"expected {len(shapes)} weight arrays, got {len(weights)}")
        for w, s in zip(weights, shapes):
            if tuple(np.shape(w)) != tuple(s):
                raise ValueError(f"weight shape {np.shape(w)} != expected {s}")
        flat = np.concatenate([np.asarray(w, np.float32).reshape(-1) for w in weights])
        self.params.copy_(torch.from_numpy(flat))

    def save_weights(self, path):
        np.savez(path, params=self.params.detach().cpu().numpy(), c_in=self.c_in, n_classes=self.n_classes)

    def load_weights(self, path):
        d = np.load(path)
        if int(d["c_in"]) != self.c_in or int(d["n_classes"]) != self.n_classes:
            raise ValueError("weight file was saved for a different architecture")
        self.params.copy_(torch.from_numpy(d["params"]))

    def save_keras_h5(self, path, whole_model=True):
        """The file ``keras.Model.save`` (``whole_model``) or ``save_weights`` writes for this architecture (net.py:418-427):
        the reference's ``NetManager.load_model`` / ``keras.models.load_model`` reads it (ubdvss_amd.keras_h5_writer)."""
        from . import keras_h5_writer
        keras_h5_writer.write_keras_model(path, self.get_weights(), self.c_in, self.n_classes,
                                          bool(self.net_config.is_fml_compatible()), whole_model)

    def load_keras_weights(self, path):
        """Weights exported from the reference's Keras model (net.py:418-494 keeps them in HDF5, which needs h5py):
        run ``np.savez(path, *model.get_weights())`` once in the Keras environment; the arrays arrive as arr_0, arr_1,
        ... in ``get_weights()`` order, which is this model's flat parameter order."""
        d = np.load(path)
        names = sorted((k for k in d.files if k.startswith("arr_")), key=lambda k: int(k[4:]))
        if not names:
            raise ValueError("no arr_<i> entries: export with np.savez(path, *model.get_weights())")
        self.set_weights([d[k] for k in names])

    def load_keras_h5(self, path):
        """Weights of a model file the reference wrote (``model.h5`` / ``inference_model.h5`` / ``model_weights.h5``,
        net.py:418-427), parsed by ubdvss_amd.keras_h5 (no h5py): ``get_weights()`` order = this flat order."""
        from . import keras_h5
        arrays, names = keras_h5.read_keras_weights(path)
        try:
            self.set_weights(arrays)
        except ValueError as e:
            raise ValueError(f"{path}: {e} (Keras weights: {names})") from None

    # ---------------------------------------------------------------- forward
    def _workspace(self, attr, nbytes):
        ws = getattr(self, attr)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            setattr(self, attr, ws)
        return ws

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _weights_key(self):
        """what the packed weight fragments at the head of the forward workspace were made from (stream apart)"""
        return (self._ws.data_ptr() if self._ws is not None else None, self.params.data_ptr(), self.params._version, self._weights_epoch)

    def graphed_forward(self, n, height, width, dtype=torch.float32):
        """A forward pass of ONE fixed shape as a captured HIP graph (``GraphedForward``): static input / output tensors and the pass's
        launches replayed with one host call (for hosts that cannot afford seven launch calls per image; on the device it is no faster
        than the launches).  Cached per shape; bit-identical to ``predict_on_device`` (the same launches)."""
        key = (int(n), int(height), int(width), dtype)
        gf = self._graphed.get(key)
        if gf is None:
            gf = self._graphed[key] = GraphedForward(self, (int(n), int(height), int(width), self.c_in), dtype)
        return gf

    def predict_on_device(self, images, out=None, preprocessing=None, postprocess=None, _prepacked=False):
        """images: torch tensor (N,H,W,C_in) on this device, float32 (fed as is) or uint8 (the
        NetConfig preprocessing is fused into the first layer).  Returns fp32 logits (N,H/4,W/4,K).
        postprocess: None, or the keyword arguments of ``postprocess_on_device`` for ANOTHER batch's logits (with
        ``outputs`` preallocated): that postprocess is enqueued together with this forward pass
        (ubd_forward_postprocess: inside the stem kernel's first blocks when the one-kernel stem runs)."""
        if images.device != self.device:
            raise ValueError("images must live on the model's device")
        if images.dim() != 4 or images.shape[3] != self.c_in:
            raise ValueError(f"expected NHWC images with {self.c_in} channels, got {tuple(images.shape)}")
        images = images.contiguous()
        n, hh, ww, _ = images.shape
        if hh % 4 or ww % 4:
            raise ValueError("image height and width must be multiples of 4")
        if images.dtype == torch.uint8:
            in_dtype = _lib.UBD_IN_U8
            pre = self.net_config.get_preprocessing_type() if preprocessing is None else preprocessing
            pre = _lib.UBD_PRE_MOBILENET if pre == PreprocessingType.MOBILENET_LIKE else _lib.UBD_PRE_NONE
        elif images.dtype == torch.float32:
            in_dtype, pre = _lib.UBD_IN_F32, _lib.UBD_PRE_NONE
        else:
            raise ValueError(f"unsupported image dtype {images.dtype}")
        if out is None:
            out = torch.empty((n, hh // 4, ww // 4, self.k_out), dtype=torch.float32, device=self.device)
        nbytes = self._lib.ubd_forward_workspace_bytes(self._h, n, hh, ww)
        ws = self._workspace("_ws", nbytes)
        # the packed weight fragments at the head of the workspace stay valid until the parameters change; the key
        # names the stream too (a pack on another stream is not ordered before this call) and is only stored once
        # the call that packed has been accepted
        stream = self._stream()
        key = (ws.data_ptr(), self.params.data_ptr(), self.params._version, self._weights_epoch, stream.value)
        if key == self._packed_key or (_prepacked and self._packed_key is not None and key[:4] == self._packed_key[:4]):
            in_dtype |= _lib.UBD_IN_PREPACKED       # _prepacked: the caller (GraphedForward) has synchronised behind the call that packed
        self._packed_key = None
        with torch.cuda.device(self.device):
            if postprocess is None:
                _lib.check(self._lib.ubd_forward(self._h, self.params.data_ptr(), images.data_ptr(), in_dtype, pre,
                                                 n, hh, ww, out.data_ptr(), ws.data_ptr(), ws.numel(), stream),
                           "ubd_forward")
            else:
                pl = postprocess["logits"]
                if not pl.is_contiguous() or pl.dtype != torch.float32 or pl.shape[3] != self.k_out:
                    raise ValueError("postprocess job: logits must be a contiguous fp32 (N,h,w,K) tensor of this model")
                pn, pmh, pmw, _ = pl.shape
                bmap, quads, classes, counts = postprocess["outputs"]
                cap = int(postprocess.get("cap", quads.shape[1]))
                pws = self._workspace("_pp_ws", self._lib.ubd_postprocess_workspace_bytes(self._h, pn, pmh, pmw, cap))
                _lib.check(self._lib.ubd_forward_postprocess(
                    self._h, self.params.data_ptr(), images.data_ptr(), in_dtype, pre, n, hh, ww, out.data_ptr(), ws.data_ptr(), ws.numel(),
                    pl.data_ptr(), pn, pmh, pmw, float(postprocess["logit_threshold"]), int(postprocess["scale"]),
                    float(postprocess["min_area"]), bmap.data_ptr() if bmap is not None else None, quads.data_ptr(),
                    classes.data_ptr() if classes is not None else None, counts.data_ptr(), cap, pws.data_ptr(), pws.numel(), stream),
                    "ubd_forward_postprocess")
        self._packed_key = key
        return out

    def predict(self, images, batch_size=None):
        """keras.Model.predict: numpy (N,H,W,C_in) -> numpy float32 logits (N,H/4,W/4,K).  Float images are fed as
        they are (the reference preprocesses on the host, data_generators.py:113,148); uint8 images are raw pixels
        and get ``NetConfig``'s preprocessing fused into the first layer, exactly like ``predict_on_device``."""
        x = np.asarray(images)
        if x.dtype != np.uint8:
            x = x.astype(np.float32, copy=False)
        x = np.ascontiguousarray(x)
        if x.ndim == 4 and x.shape[0] == 1 and x.shape[3] == self.c_in and x.shape[1] % 4 == 0 and x.shape[2] % 4 == 0:
            # one image per call is the reference's latency protocol (predict.py:73-78): the static input / output tensors of this shape
            # (nothing is allocated per call) and the pass's seven launches.  Replaying them as a captured graph (gf(images)) is one host
            # call but no faster: 0.127 against 0.118 ms numpy in / numpy out for a 512 x 512 image since the stem is one kernel.
            gf = self.graphed_forward(1, x.shape[1], x.shape[2], torch.uint8 if x.dtype == np.uint8 else torch.float32)
            gf.x.copy_(torch.from_numpy(x), non_blocking=True)
            return self.predict_on_device(gf.x, out=gf.out).cpu().numpy()
        xt = torch.from_numpy(x).to(self.device)
        return self.predict_on_device(xt).cpu().numpy()

    # ---------------------------------------------------------------- postprocess
    def alloc_postprocess_outputs(self, n, mh, mw, cap, want_map=True):
        """Result buffers of ``ubd_postprocess``.  Nothing is cleared: the library writes ``counts[i]`` for every image and the first
        ``min(counts[i], cap)`` entries of an image's lists; entries behind a list's end are unspecified (no fill kernels on this
        path -- rounds 1-4 zero-filled 1 MB per call through two torch kernels)."""
        dev = self.device
        bmap = torch.empty((n, mh, mw), dtype=torch.int32, device=dev) if want_map else None
        quads = torch.empty((n, cap, 8), dtype=torch.int32, device=dev)
        classes = torch.empty((n, cap), dtype=torch.int32, device=dev) if self.n_classes > 0 else None
        counts = torch.empty((n,), dtype=torch.int32, device=dev)
        return bmap, quads, classes, counts

    def postprocess_on_device(self, logits, logit_threshold, scale, min_area, cap=256, want_map=True, outputs=None):
        """logits: fp32 (N,h,w,K) device tensor.  Returns (binary_map int32 (N,h,w) or None,
        quads int32 (N,cap,8), classes int32 (N,cap) or None, counts int32 (N)).  Runs on the current
        torch stream; `outputs` may hold preallocated result tensors (alloc_postprocess_outputs)."""
        logits = logits.contiguous()
        n, mh, mw, k = logits.shape
        if k != self.k_out:
            raise ValueError(f"logits have {k} channels, model has {self.k_out}")
        dev = self.device
        if outputs is None:
            outputs = self.alloc_postprocess_outputs(n, mh, mw, cap, want_map)
        bmap, quads, classes, counts = outputs
        nbytes = self._lib.ubd_postprocess_workspace_bytes(self._h, n, mh, mw, cap)
        ws = self._workspace("_pp_ws", nbytes)
        with torch.cuda.device(dev):
            _lib.check(self._lib.ubd_postprocess(
                self._h, logits.data_ptr(), n, mh, mw, float(logit_threshold), int(scale), float(min_area),
                bmap.data_ptr() if bmap is not None else None, quads.data_ptr(),
                classes.data_ptr() if classes is not None else None, counts.data_ptr(), cap,
                ws.data_ptr(), ws.numel(), self._stream()), "ubd_postprocess")
        return bmap, quads, classes, counts


class GraphedForward:
    """``Model.predict_on_device`` of one fixed shape, captured once into a HIP graph and replayed: one host call instead of seven
    launches (the stem kernel + six dilated layers at batch 1; with so few launches the replay is no faster on the device -- 0.077
    against 0.072 ms for a 512 x 512 image -- but it costs the host one call and allocates nothing).  Owns a static input and a static output tensor; ``__call__``
    copies the caller's images into the static input unless they ARE that tensor, makes sure the packed weight fragments are current
    (a parameter change re-packs through one ordinary call, then re-captures), replays, and returns the static output (valid until
    the next call).  The library allocates nothing inside a call and takes every pointer from the caller, which is what
    makes the launches capturable."""

    def __init__(self, model, shape, dtype=torch.float32):
        self.model = model
        n, hh, ww, _ = shape
        self.x = torch.zeros(shape, dtype=dtype, device=model.device)
        self.out = torch.empty((n, hh // 4, ww // 4, model.k_out), dtype=torch.float32, device=model.device)
        self._graph = None
        self._key = None

    def _capture(self):
        m = self.model
        m.predict_on_device(self.x, out=self.out)          # sizes the workspace, packs the weights on the current stream
        m.predict_on_device(self.x, out=self.out)
        torch.cuda.synchronize(m.device)                    # the pack is complete: the captured launches may assume it
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            m.predict_on_device(self.x, out=self.out, _prepacked=True)
        self._graph, self._key = g, m._weights_key()

    def __call__(self, images):
        m = self.model
        if tuple(images.shape) != tuple(self.x.shape) or images.dtype != self.x.dtype:
            raise ValueError(f"this graph was captured for {tuple(self.x.shape)} {self.x.dtype}, got {tuple(images.shape)} {images.dtype}")
        if images.data_ptr() != self.x.data_ptr():
            self.x.copy_(images, non_blocking=True)
        if self._graph is None or self._key != m._weights_key():
            self._capture()                                 # first use, new parameters, or the workspace moved
        self._graph.replay()
        return self.out


class NetManager:
    """Builds / saves / loads the model (net.py:255-494) with the reference's method names, arguments and FILES: models are
    written as Keras 2.2 HDF5 (``keras.Model.save`` layout, ubdvss_amd.keras_h5_writer -- the reference's ``load_model`` opens
    them) next to the pickled NetConfig, and read back from there; a Keras ``.h5`` the reference wrote and the ``.npz`` files
    of earlier versions of this package are read too."""

    CURRENT_MODEL_FILENAME = "model.h5"
    INFERENCE_MODEL_FILENAME = "inference_model.h5"
    MODEL_WEIGHTS_FILENAME = "model_weights.h5"
    PICKLED_CONFIG_FILENAME = "config.pkl"

    def __init__(self, log_dir, net_config=None):
        self._log_dir = log_dir
        if net_config is not None:
            self._net_config = net_config
        else:
            self.load_config()
        self._model = None

    def build_model(self, dtype="float32", seed=None):
        self._model = Model(self._net_config, dtype=dtype, seed=seed)
        self._net_config._scale = 4                     # net.py:314
        return self._net_config

    def get_keras_model(self):                          # name kept for drop-in use (net.py:415-416)
        return self._model

    def get_model(self):
        return self._model

    def save_model(self, step=None):
        """net.py:418-420: a numbered snapshot ``model{step:03d}.h5`` plus the current model ``model.h5``, both in the layout
        of ``keras.Model.save`` (``step`` may be omitted here; the reference requires it).  Also writes config.pkl."""
        if step is not None:
            self._model.save_keras_h5(os.path.join(self._log_dir, "model{:03d}.h5".format(step)))
        self._model.save_keras_h5(os.path.join(self._log_dir, self.CURRENT_MODEL_FILENAME))
        self.save_config()

    def save_inference(self):
        """net.py:422-427: ``model_weights.h5`` (``save_weights`` layout) and ``inference_model.h5`` (``save`` layout; the
        reference rebuilds the graph in between to drop the optimizer -- there is no graph to rebuild here)."""
        self._model.save_keras_h5(os.path.join(self._log_dir, self.MODEL_WEIGHTS_FILENAME), whole_model=False)
        self._model.save_keras_h5(os.path.join(self._log_dir, self.INFERENCE_MODEL_FILENAME))

    def load_another_model(self, another_log_dir, dtype="float32"):
        """net.py:429-441: model and architecture-dependent configuration from ``another_log_dir``, the
        architecture-independent knobs (side multiple, max image side, min pixels) of this manager's configuration.
        (The reference passes its own config as ``from_others``' second positional argument, net.py:440, which makes
        it the side multiple; the documented intent is implemented here.)"""
        other = NetManager(another_log_dir, self._net_config)
        other.load_config()
        other.load_model(dtype=dtype)
        self._model = other._model
        mine = self._net_config
        self._net_config = NetConfig.from_others(other._net_config, side_multiple=mine.get_side_multiple(),
                                                 max_image_side=mine.get_max_side(),
                                                 min_pixels_for_detection=mine.get_min_pixels_for_detection())
        self._model.net_config = self._net_config
        return self._net_config

    def load_model(self, path_to_model=None, dtype="float32"):
        """net.py:443-466: models live in the log dir next to their config, so ``path_to_model`` must stay None (the
        reference asserts the same); preference: inference model, then current model; within each, the Keras ``.h5`` (written
        by the reference or by this package; read by ubdvss_amd.keras_h5, no h5py needed) before a legacy ``.npz``."""
        assert path_to_model is None, "Programmer! Models are stored in log_dir, near their config. " \
                                      "If you load model from other location model and config will not match most likely."
        candidates = []
        for name in (self.INFERENCE_MODEL_FILENAME, self.CURRENT_MODEL_FILENAME):
            stem = os.path.splitext(name)[0]
            candidates += [stem + ".h5", stem + ".npz"]
        for name in candidates:
            cand = os.path.join(self._log_dir, name)
            if os.path.exists(cand):
                return self._load_model(cand, dtype)
        raise FileNotFoundError(f"Model not found in dir {self._log_dir}. Must contain "
                                f"at least one of the following files {candidates}")

    def _load_model(self, path_to_model, dtype="float32"):
        logging.info(f"loading model from {path_to_model}")
        self._model = Model(self._net_config, dtype=dtype)
        if path_to_model.endswith(".h5"):
            self._model.load_keras_h5(path_to_model)
        else:
            self._model.load_weights(path_to_model)
        return self._net_config

    def save_config(self):                              # net.py:468-469
        with open(os.path.join(self._log_dir, self.PICKLED_CONFIG_FILENAME), 'wb') as f:
            pickle.dump(self._net_config, f)

    def load_config(self):                              # net.py:471-472; also reads a config.pkl the reference wrote
        self._net_config = load_reference_pickle(os.path.join(self._log_dir, self.PICKLED_CONFIG_FILENAME))
