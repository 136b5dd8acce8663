"""Writer for the reference's model files -- Keras 2.2.x HDF5 (``keras.Model.save`` / ``save_weights``), without h5py.

The reference stores models through Keras (net.py:418-427: ``model.save(model.h5)``, ``save_weights`` + ``save(inference_model.h5)``)
and loads them with ``keras.models.load_model`` (net.py:474-494).  This module emits that file layout in plain Python + numpy so that
a model trained here can be handed BACK to the reference ("interchange"): the HDF5 subset libhdf5 1.8 / 1.10
writes with h5py's default ``libver`` -- superblock v0, v1 object headers, old-style groups (symbol table message, one v1 B-tree
node, one symbol-table node, local heap), contiguous little-endian float32 datasets, v1 attribute messages holding fixed-length
strings (scalars and 1-D arrays).  Layout of the Keras side (keras/engine/saving.py, 2.2.4):

  model.save:      root attrs ``keras_version``, ``backend``, ``model_config`` (JSON of {"class_name": "Model", "config": get_config()}),
                   group ``/model_weights`` = the save_weights layout; no ``training_config`` (Keras then returns the model uncompiled
                   with a warning -- the optimizer state of this package's Adam is not Keras's)
  save_weights:    attrs ``layer_names``, ``backend``, ``keras_version``; one group per layer with attr ``weight_names`` and one
                   dataset per weight (names like ``conv2d_1/kernel:0``: the '/' makes a nested group)

``model_config`` is the functional-API config Keras 2.2.4 produces for net.py:278-314 (InputLayer, ZeroPadding2D, SeparableConv2D,
Conv2D with their full ``get_config`` dictionaries and ``inbound_nodes``).  Files are checked by reading them back with the real
h5py where one is installed (tests/test_keras_h5.py) and with this package's own reader.
"""
import json
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INTERNAL_K = 32, 16            # symbol-table node: up to 2 * LEAF_K entries (one node per group suffices here)
KERAS_VERSION = b"2.2.4"


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


# ------------------------------------------------------------------------------------------------ messages
def _msg(mtype, data):
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), 0) + data


def _dt_f32():
    return struct.pack("<BBBBI", 0x11, 0x20, 0x1F, 0x00, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)


def _dt_f64():
    return struct.pack("<BBBBI", 0x11, 0x20, 0x3F, 0x00, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)


def _dt_str(size):
    return struct.pack("<BBBBI", 0x13, 0x01, 0x00, 0x00, size)          # fixed length, null-padded, ASCII (what h5py writes for numpy 'S')


def _ds(shape):
    if shape is None:
        return struct.pack("<BBB5x", 1, 0, 0)                            # scalar
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", int(d)) for d in shape)


def _attr(name, dt, ds, data):
    nm = name.encode("utf8") + b"\0"
    return _msg(0x000C, struct.pack("<BxHHH", 1, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + data)


def _attr_bytes(name, value):
    value = bytes(value)
    return _attr(name, _dt_str(max(1, len(value))), _ds(None), value if value else b"\0")


def _attr_names(name, values):
    """1-D array of byte strings (``layer_names`` / ``weight_names``); an empty list is what numpy makes of it: float64, shape (0,)."""
    if not values:
        return _attr(name, _dt_f64(), _ds((0,)), b"")
    vals = [v.encode("utf8") if isinstance(v, str) else bytes(v) for v in values]
    size = max(len(v) for v in vals)
    return _attr(name, _dt_str(size), _ds((len(vals),)), b"".join(v.ljust(size, b"\0") for v in vals))


def _object_header(msgs):
    body = b"".join(msgs)
    return struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body


# ------------------------------------------------------------------------------------------------ file assembly
class _Writer:
    def __init__(self):
        self.buf = bytearray(96)                                         # the superblock is filled in last

    def alloc(self, data):
        while len(self.buf) % 8:
            self.buf.append(0)
        addr = len(self.buf)
        self.buf += data
        return addr

    def dataset(self, array):
        a = np.ascontiguousarray(array, dtype="<f4")
        raw = a.tobytes()
        daddr = self.alloc(raw) if raw else UNDEF
        msgs = [_msg(0x0001, _ds(a.shape)), _msg(0x0003, _dt_f32()), _msg(0x0005, bytes([2, 1, 0, 0])),
                _msg(0x0008, struct.pack("<BBQQ", 3, 1, daddr, len(raw)))]
        return self.alloc(_object_header(msgs))

    def group(self, children, attr_msgs=()):
        """children: {name: object header address}.  Returns (object header address, B-tree address, heap address)."""
        if len(children) > 2 * LEAF_K:
            raise ValueError("too many links for one symbol-table node")
        names = sorted(children, key=lambda s: s.encode("utf8"))         # the B-tree orders links by strcmp
        heap = bytearray(8)                                              # offset 0: the empty name (key 0 of the B-tree)
        offs = {}
        for nm in names:
            offs[nm] = len(heap)
            heap += _pad8(nm.encode("utf8") + b"\0")
        free_off = len(heap)
        heap += struct.pack("<QQ", 1, 16)                                # one free block at the end: (next = 1: end of list, size)
        heap_data = self.alloc(bytes(heap))
        heap_addr = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_off, heap_data))
        snod = bytearray(b"SNOD" + struct.pack("<BxH", 1, len(names)))
        for nm in names:
            snod += struct.pack("<QQII16x", offs[nm], children[nm], 0, 0)
        snod += bytes(8 + 2 * LEAF_K * 40 - len(snod))
        snod_addr = self.alloc(bytes(snod))
        tree = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, UNDEF, UNDEF))
        tree += struct.pack("<Q", 0)                                     # key 0: the empty string
        if names:
            tree += struct.pack("<QQ", snod_addr, offs[names[-1]])       # child 0, key 1: the largest name in it
        tree += bytes(24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8 - len(tree))
        tree_addr = self.alloc(bytes(tree))
        hdr = self.alloc(_object_header([_msg(0x0011, struct.pack("<QQ", tree_addr, heap_addr))] + list(attr_msgs)))
        return hdr, tree_addr, heap_addr

    def finish(self, root):
        hdr, tree, heap = root
        while len(self.buf) % 8:
            self.buf.append(0)
        sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBxHHI", 0, 0, 0, 0, 0, 8, 8, LEAF_K, INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.buf), UNDEF)
        sb += struct.pack("<QQII", 0, hdr, 1, 0) + struct.pack("<QQ", tree, heap)       # root entry, cached B-tree / heap addresses
        assert len(sb) == 96
        self.buf[:96] = sb
        return bytes(self.buf)


# ------------------------------------------------------------------------------------------------ the Keras side
def model_layers(c_in, n_classes, fml_compatible=True):
    """[(layer name, class name, [(weight name, shape)], config extras)] in ``model.layers`` order for net.py:278-314."""
    layers = [("input_1", "InputLayer", [], {"c_in": c_in})]
    sep = conv = pad = 0
    cin = c_in
    for stride in (2, 1, 2):
        padding = "same"
        if stride == 2 and fml_compatible:                               # net.py:229-232
            pad += 1
            layers.append((f"zero_padding2d_{pad}", "ZeroPadding2D", [], {}))
            padding = "valid"
        sep += 1
        n = f"separable_conv2d_{sep}"
        layers.append((n, "SeparableConv2D", [(f"{n}/depthwise_kernel:0", (3, 3, cin, 1)), (f"{n}/pointwise_kernel:0", (1, 1, cin, 24)),
                                              (f"{n}/bias:0", (24,))], {"strides": stride, "padding": padding}))
        cin = 24
    for d in (1, 2, 4, 8, 16, 1):
        conv += 1
        n = f"conv2d_{conv}"
        layers.append((n, "Conv2D", [(f"{n}/kernel:0", (3, 3, 24, 24)), (f"{n}/bias:0", (24,))],
                       {"filters": 24, "kernel": 3, "dilation": d, "activation": "relu"}))
    conv += 1
    n = f"conv2d_{conv}"
    layers.append((n, "Conv2D", [(f"{n}/kernel:0", (1, 1, 24, 1 + n_classes)), (f"{n}/bias:0", (1 + n_classes,))],
                   {"filters": 1 + n_classes, "kernel": 1, "dilation": 1, "activation": "linear"}))
    return layers


_GLOROT = {"class_name": "VarianceScaling", "config": {"scale": 1.0, "mode": "fan_avg", "distribution": "uniform", "seed": None}}
_ZEROS = {"class_name": "Zeros", "config": {}}


def model_config(c_in, n_classes, fml_compatible=True):
    """``{"class_name": "Model", "config": model.get_config()}`` as Keras 2.2.4 serialises the model of net.py:278-314."""
    out, prev = [], None
    for name, cls, _, x in model_layers(c_in, n_classes, fml_compatible):
        if cls == "InputLayer":
            cfg = {"batch_input_shape": [None, None, None, c_in], "dtype": "float32", "sparse": False, "name": name}
        elif cls == "ZeroPadding2D":
            cfg = {"name": name, "trainable": True, "padding": [[1, 0], [1, 0]], "data_format": "channels_last"}
        elif cls == "SeparableConv2D":
            cfg = {"name": name, "trainable": True, "filters": 24, "kernel_size": [3, 3], "strides": [x["strides"]] * 2, "padding": x["padding"],
                   "data_format": "channels_last", "dilation_rate": [1, 1], "activation": "relu", "use_bias": True,
                   "bias_initializer": _ZEROS, "bias_regularizer": None, "activity_regularizer": None, "bias_constraint": None,
                   "depth_multiplier": 1, "depthwise_initializer": _GLOROT, "pointwise_initializer": _GLOROT,
                   "depthwise_regularizer": None, "pointwise_regularizer": None, "depthwise_constraint": None, "pointwise_constraint": None}
        else:
            cfg = {"name": name, "trainable": True, "filters": x["filters"], "kernel_size": [x["kernel"]] * 2, "strides": [1, 1], "padding": "same",
                   "data_format": "channels_last", "dilation_rate": [x["dilation"]] * 2, "activation": x["activation"], "use_bias": True,
                   "kernel_initializer": _GLOROT, "bias_initializer": _ZEROS, "kernel_regularizer": None, "bias_regularizer": None,
                   "activity_regularizer": None, "kernel_constraint": None, "bias_constraint": None}
        out.append({"name": name, "class_name": cls, "config": cfg, "inbound_nodes": [] if prev is None else [[[prev, 0, 0, {}]]]})
        prev = name
    return {"class_name": "Model", "config": {"name": "dilated_conv", "layers": out, "input_layers": [["input_1", 0, 0]],
                                              "output_layers": [[prev, 0, 0]]}}


def write_keras_model(path, weights, c_in, n_classes, fml_compatible=True, whole_model=True):
    """``weights``: the arrays of ``Model.get_weights()`` (this package's flat order = Keras's).  ``whole_model``: the layout of
    ``keras.Model.save`` (what the reference's ``load_model`` opens); otherwise that of ``save_weights``."""
    layers = model_layers(c_in, n_classes, fml_compatible)
    expected = [shape for _, _, ws, _ in layers for _, shape in ws]
    if len(weights) != len(expected):
        raise ValueError(f"expected {len(expected)} weight arrays, got {len(weights)}")
    w = _Writer()
    it = iter(weights)
    layer_groups = {}
    for name, _, ws, _ in layers:
        inner = {}
        for wname, shape in ws:
            arr = np.asarray(next(it), dtype=np.float32)
            if tuple(arr.shape) != tuple(shape):
                raise ValueError(f"{wname}: shape {arr.shape} != expected {shape}")
            inner[wname.split("/", 1)[1]] = w.dataset(arr)
        children = {}
        if inner:
            children[name] = w.group(inner)[0]                           # "<layer>/<weight>:0" -> nested group <layer>
        layer_groups[name] = w.group(children, [_attr_names("weight_names", [wn for wn, _ in ws])])[0]
    wattrs = [_attr_names("layer_names", [n for n, _, _, _ in layers]), _attr_bytes("backend", b"tensorflow"),
              _attr_bytes("keras_version", KERAS_VERSION)]
    if whole_model:
        mw = w.group(layer_groups, wattrs)[0]
        cfg = json.dumps(model_config(c_in, n_classes, fml_compatible)).encode("utf8")
        root = w.group({"model_weights": mw}, [_attr_bytes("keras_version", KERAS_VERSION), _attr_bytes("backend", b"tensorflow"),
                                               _attr_bytes("model_config", cfg)])
    else:
        root = w.group(layer_groups, wattrs)
    with open(path, "wb") as f:
        f.write(w.finish(root))
