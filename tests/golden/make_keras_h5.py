"""Writes the HDF5 fixtures of tests/test_keras_h5.py with the REAL libhdf5 (h5py), in the exact layout Keras 2.2.x
emits (keras/engine/saving.py: save_model / save_weights_to_hdf5_group), so that the pure-Python reader
ubdvss_amd/keras_h5.py is checked against files produced by the library the reference uses -- not by itself.

Run with an interpreter that has h5py (not a dependency of this repo; in the build container:
    /opt/conda/bin/python3.9 tests/golden/make_keras_h5.py
).  Weight VALUES are seeded noise (no trained ubdvss model is published, README.md:9); layer / weight names and
attribute types are those of the reference's model (net.py:278-314).
"""
import json
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def model_layers(c_in, n_classes, use_bn=False):
    """(layer name, [(weight name, shape)]) in model.layers order for net.py:278-314 (fml-compatible); use_bn: the
    conv_bn(..., use_bn=True) branch (net.py:248-250): a BatchNormalization and an Activation layer behind every hidden conv."""
    layers = [("input_1", [])]
    sep, conv, pad = 0, 0, 0
    bn = [0]

    def bn_layers():
        if not use_bn:
            return []
        bn[0] += 1
        n = f"batch_normalization_{bn[0]}"
        return [(n, [(f"{n}/gamma:0", (24,)), (f"{n}/beta:0", (24,)), (f"{n}/moving_mean:0", (24,)), (f"{n}/moving_variance:0", (24,))]),
                (f"activation_{bn[0]}", [])]
    cin = c_in
    for stride in (2, 1, 2):
        if stride == 2:
            pad += 1
            layers.append((f"zero_padding2d_{pad}", []))
        sep += 1
        n = f"separable_conv2d_{sep}"
        layers.append((n, [(f"{n}/depthwise_kernel:0", (3, 3, cin, 1)), (f"{n}/pointwise_kernel:0", (1, 1, cin, 24)),
                           (f"{n}/bias:0", (24,))]))
        layers += bn_layers()
        cin = 24
    for _ in range(6):
        conv += 1
        n = f"conv2d_{conv}"
        layers.append((n, [(f"{n}/kernel:0", (3, 3, 24, 24)), (f"{n}/bias:0", (24,))]))
        layers += bn_layers()
    conv += 1
    n = f"conv2d_{conv}"
    layers.append((n, [(f"{n}/kernel:0", (1, 1, 24, 1 + n_classes)), (f"{n}/bias:0", (1 + n_classes,))]))
    return layers


def save_weights_to_hdf5_group(g, layers, rng, **dset_kw):
    g.attrs["layer_names"] = [name.encode("utf8") for name, _ in layers]
    g.attrs["backend"] = b"tensorflow"
    g.attrs["keras_version"] = b"2.2.4"
    for name, weights in layers:
        lg = g.create_group(name)
        lg.attrs["weight_names"] = [w.encode("utf8") for w, _ in weights]
        for wname, shape in weights:
            val = rng.uniform(-1, 1, shape).astype(np.float32)
            if wname.endswith("moving_variance:0"):              # a variance is positive; gamma near one
                val = rng.uniform(0.2, 2.0, shape).astype(np.float32)
            elif wname.endswith("gamma:0"):
                val = rng.uniform(0.5, 1.5, shape).astype(np.float32)
            d = lg.create_dataset(wname, val.shape, dtype=val.dtype, **dset_kw)
            d[:] = val


def write_model(path, c_in, n_classes, seed, whole_model, libver=None, use_bn=False, **dset_kw):
    rng = np.random.default_rng(seed)
    layers = model_layers(c_in, n_classes, use_bn)
    with h5py.File(path, "w", libver=libver) as f:
        if whole_model:                                          # keras.Model.save
            f.attrs["keras_version"] = b"2.2.4"
            f.attrs["backend"] = b"tensorflow"
            f.attrs["model_config"] = json.dumps({"class_name": "Model", "config": {"name": "model_1", "layers": [n for n, _ in layers]}}).encode("utf8")
            f.attrs["training_config"] = json.dumps({"optimizer_config": {"class_name": "Adam"}, "loss": "detection_loss"}).encode("utf8")
            save_weights_to_hdf5_group(f.create_group("model_weights"), layers, rng, **dset_kw)
            og = f.create_group("optimizer_weights")
            og.attrs["weight_names"] = [b"Adam/iterations:0"]
            og.create_dataset("Adam/iterations:0", (), dtype=np.int64)[()] = 7
        else:                                                    # keras.Model.save_weights
            save_weights_to_hdf5_group(f, layers, rng, **dset_kw)


if __name__ == "__main__":
    write_model(os.path.join(HERE, "keras_model_rgb.h5"), 3, 0, 100, True)                         # model.save, default libver
    write_model(os.path.join(HERE, "keras_weights_grey_cls2.h5"), 1, 2, 101, False)                # save_weights
    write_model(os.path.join(HERE, "keras_model_grey_gzip.h5"), 1, 0, 102, True,                    # chunked + shuffle + deflate
                chunks=True, compression="gzip", shuffle=True)
    write_model(os.path.join(HERE, "keras_model_rgb_bn.h5"), 3, 0, 104, True, use_bn=True)          # conv_bn(use_bn=True) model
    # libver="latest" (not what Keras writes): v2 object headers + compact link messages for small groups ...
    rng = np.random.default_rng(103)
    with h5py.File(os.path.join(HERE, "tiny_latest.h5"), "w", libver="latest") as f:
        save_weights_to_hdf5_group(f, [("conv2d_1", [("conv2d_1/kernel:0", (3, 3, 2, 2)), ("conv2d_1/bias:0", (2,))]),
                                       ("conv2d_2", [("conv2d_2/kernel:0", (1, 1, 2, 1)), ("conv2d_2/bias:0", (1,))])], rng)
    # ... and dense (fractal-heap) link storage once a group has more than 8 links: the reader must refuse it by name
    with h5py.File(os.path.join(HERE, "tiny_latest_dense.h5"), "w", libver="latest") as f:
        save_weights_to_hdf5_group(f, [(f"zero_padding2d_{i}", []) for i in range(12)], rng)
    print("h5py", h5py.__version__, "hdf5", h5py.version.hdf5_version)
