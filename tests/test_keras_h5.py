"""CPU: the pure-Python reader of the reference's Keras HDF5 model files (ubdvss_amd/keras_h5.py; reference
net.py:418-494) against fixtures written by the real libhdf5 / h5py in Keras's layout (tests/golden/make_keras_h5.py),
and the reference-pickled NetConfig loader (net.py:468-472)."""
import os
import pickle
import sys
import types

import numpy as np
import pytest

from ubdvss_amd import keras_h5
from ubdvss_amd.net import NetConfig, NetManager, PreprocessingType, weight_shapes, load_reference_pickle


def _expected(seed, shapes):
    rng = np.random.default_rng(seed)
    return [rng.uniform(-1, 1, s).astype(np.float32) for s in shapes]


@pytest.mark.parametrize("fname,c_in,n_cls,seed", [
    ("keras_model_rgb.h5", 3, 0, 100),             # keras.Model.save (weights under /model_weights)
    ("keras_weights_grey_cls2.h5", 1, 2, 101),     # keras.Model.save_weights (weights at the root)
    ("keras_model_grey_gzip.h5", 1, 0, 102),       # chunked + shuffle + deflate datasets
])
def test_full_model_files(golden_dir, fname, c_in, n_cls, seed):
    arrays, names = keras_h5.read_keras_weights(os.path.join(golden_dir, fname))
    shapes = weight_shapes(c_in, n_cls)
    assert [a.shape for a in arrays] == [tuple(s) for s in shapes]
    assert names[0] == "separable_conv2d_1/depthwise_kernel:0" and names[-1] == "conv2d_7/bias:0"
    for a, e in zip(arrays, _expected(seed, shapes)):
        assert a.dtype == np.float32 and np.array_equal(a, e)


def test_new_style_groups_and_v2_headers(golden_dir):
    arrays, names = keras_h5.read_keras_weights(os.path.join(golden_dir, "tiny_latest.h5"))
    assert names == ["conv2d_1/kernel:0", "conv2d_1/bias:0", "conv2d_2/kernel:0", "conv2d_2/bias:0"]
    for a, e in zip(arrays, _expected(103, [(3, 3, 2, 2), (2,), (1, 1, 2, 1), (1,)])):
        assert np.array_equal(a, e)


def test_unsupported_features_fail_by_name(golden_dir, tmp_path):
    with pytest.raises(keras_h5.KerasH5Error, match="dense link storage"):
        keras_h5.read_keras_weights(os.path.join(golden_dir, "tiny_latest_dense.h5"))
    bad = tmp_path / "x.h5"
    bad.write_bytes(b"not hdf5 at all" * 10)
    with pytest.raises(keras_h5.KerasH5Error, match="not an HDF5 file"):
        keras_h5.read_keras_weights(str(bad))


def test_attributes_of_model_save(golden_dir):
    f = keras_h5.H5File(os.path.join(golden_dir, "keras_model_rgb.h5"))
    assert f.root.attrs["keras_version"] == b"2.2.4" and f.root.attrs["backend"] == b"tensorflow"   # variable-length strings
    assert b"Model" in f.root.attrs["model_config"]
    assert int(f.root["optimizer_weights"]["Adam"]["iterations:0"].read()) == 7                      # scalar int64 dataset
    assert len(f.root["model_weights"].attrs["layer_names"]) == 13


def test_reference_pickled_config_loads(tmp_path):
    """config.pkl written by the reference names the class semantic_segmentation.net.NetConfig (net.py:468-469)."""
    pkg = types.ModuleType("semantic_segmentation")
    mod = types.ModuleType("semantic_segmentation.net")
    pkg.__path__ = []

    class RefNetConfig:                       # stand-in with the reference's attribute names (net.py:98-132)
        pass
    RefNetConfig.__module__, RefNetConfig.__qualname__, RefNetConfig.__name__ = "semantic_segmentation.net", "NetConfig", "NetConfig"
    import enum
    RefPre = enum.Enum("PreprocessingType", {"NONE": 0, "MOBILENET_LIKE": 1}, module="semantic_segmentation.net")
    mod.NetConfig, mod.PreprocessingType = RefNetConfig, RefPre
    sys.modules["semantic_segmentation"], sys.modules["semantic_segmentation.net"] = pkg, mod
    try:
        c = RefNetConfig()
        c._class_names = ["ean13", "qr"]; c._class_name_to_id = {"ean13": 0, "qr": 1}; c._is_classification_supported = True
        c._grey, c._scale, c._fml_compatible, c._preprocessing = False, 4, True, RefPre.MOBILENET_LIKE
        c._side_multiple, c._max_side, c._min_pixels_for_detection = 64, 1024, 7
        blob = pickle.dumps(c)
    finally:
        del sys.modules["semantic_segmentation"], sys.modules["semantic_segmentation.net"]
    (tmp_path / "config.pkl").write_bytes(blob)
    with pytest.raises(Exception):
        pickle.loads(blob)                                        # the reference package is not importable here
    cfg = load_reference_pickle(str(tmp_path / "config.pkl"))
    assert isinstance(cfg, NetConfig) and cfg.get_n_classes() == 2 and cfg.get_class_id("qr") == 1
    assert cfg.get_preprocessing_type() is PreprocessingType.MOBILENET_LIKE and cfg.get_max_side() == 1024
    assert not cfg.is_grey() and cfg.get_min_pixels_for_detection() == 7
    assert NetManager(str(tmp_path)).build_model.__self__._net_config.get_scale() == 4     # NetManager(log_dir) loads it


def test_batchnorm_model_file_is_read_and_folded(golden_dir):
    """A model built with conv_bn(use_bn=True) (net.py:248-250): the file carries a batch_normalization_k group behind every
    hidden conv; the reader returns get_weights() order (conv arrays, then gamma, beta, moving_mean, moving_variance per layer)
    and fold_batchnorm turns the 59 arrays into the 29 of the BN-free architecture."""
    from ubdvss_amd.net import fold_batchnorm
    arrays, names = keras_h5.read_keras_weights(os.path.join(golden_dir, "keras_model_rgb_bn.h5"))
    assert len(arrays) == 59
    assert names[:7] == ["separable_conv2d_1/depthwise_kernel:0", "separable_conv2d_1/pointwise_kernel:0", "separable_conv2d_1/bias:0",
                         "batch_normalization_1/gamma:0", "batch_normalization_1/beta:0", "batch_normalization_1/moving_mean:0",
                         "batch_normalization_1/moving_variance:0"]
    assert names[-8:-2] == ["conv2d_6/kernel:0", "conv2d_6/bias:0", "batch_normalization_9/gamma:0", "batch_normalization_9/beta:0",
                            "batch_normalization_9/moving_mean:0", "batch_normalization_9/moving_variance:0"]
    assert all(a.min() > 0 for a, n in zip(arrays, names) if n.endswith("moving_variance:0"))
    folded = fold_batchnorm(arrays)
    assert [a.shape for a in folded] == [tuple(s) for s in weight_shapes(3, 0)]
    s = arrays[3] / np.sqrt(arrays[6].astype(np.float64) + 1e-3)
    assert np.allclose(folded[1], arrays[1] * s, rtol=1e-6) and np.allclose(folded[2], (arrays[2] - arrays[5]) * s + arrays[4], rtol=1e-5, atol=1e-6)
    assert np.array_equal(folded[0], arrays[0]) and np.array_equal(folded[-1], arrays[-1])


# ------------------------------------------------------------------------------------------------ the writer (net.py:418-427)
H5PY_PYTHON = "/opt/conda/bin/python3.9"          # an interpreter with the real h5py / libhdf5 (build container; not a dependency)

_H5PY_CHECK = r"""
import sys, json, h5py, numpy as np
path, want_model = sys.argv[1], sys.argv[2] == "1"
f = h5py.File(path, "r")
out = {"root_attrs": sorted(f.attrs.keys())}
g = f["model_weights"] if want_model else f
out["layer_names"] = [n.decode() for n in g.attrs["layer_names"]]
out["backend"] = g.attrs["backend"].decode(); out["keras_version"] = g.attrs["keras_version"].decode()
names, sums, shapes = [], [], []
for n in out["layer_names"]:
    for w in g[n].attrs["weight_names"]:
        d = g[n][w.decode()]
        assert d.dtype == np.float32
        names.append(w.decode()); shapes.append(list(d.shape)); sums.append(float(np.asarray(d[()], np.float64).sum()))
out["weight_names"], out["shapes"], out["sums"] = names, shapes, sums
if want_model:
    cfg = json.loads(f.attrs["model_config"].decode())
    out["classes"] = [l["class_name"] for l in cfg["config"]["layers"]]
    out["cfg_names"] = [l["name"] for l in cfg["config"]["layers"]]
    out["model_class"] = cfg["class_name"]; out["outputs"] = cfg["config"]["output_layers"]
print(json.dumps(out))
"""


@pytest.mark.parametrize("c_in,n_cls,fml,whole", [(3, 0, True, True), (1, 2, False, False), (3, 3, True, True), (1, 0, True, True)])
def test_writer_round_trip_and_real_h5py(tmp_path, c_in, n_cls, fml, whole):
    """ubdvss_amd.keras_h5_writer emits what keras.Model.save / save_weights emit (net.py:418-427): read back by this package's
    reader, and -- where an interpreter with the real h5py exists -- by libhdf5 itself: same layer names, weight names, shapes,
    values and attributes as Keras writes; model_config parses into the layer list of net.py:278-314."""
    import json
    import subprocess
    from ubdvss_amd import keras_h5_writer
    shapes = weight_shapes(c_in, n_cls)
    w = _expected(200 + c_in + n_cls, shapes)
    path = str(tmp_path / "out.h5")
    keras_h5_writer.write_keras_model(path, w, c_in, n_cls, fml, whole)
    arrays, names = keras_h5.read_keras_weights(path)
    assert len(arrays) == len(w) and all(np.array_equal(a, b) and a.dtype == np.float32 for a, b in zip(arrays, w))
    assert names[0] == "separable_conv2d_1/depthwise_kernel:0" and names[-1] == "conv2d_7/bias:0"
    with pytest.raises(ValueError):
        keras_h5_writer.write_keras_model(path + "x", w[:-1], c_in, n_cls, fml, whole)
    if not os.path.exists(H5PY_PYTHON):
        pytest.skip("no interpreter with h5py here: the libhdf5 half of the check needs one")
    r = subprocess.run([H5PY_PYTHON, "-c", _H5PY_CHECK, path, "1" if whole else "0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["weight_names"] == names and got["shapes"] == [list(s) for s in shapes]
    assert np.allclose(got["sums"], [float(a.astype(np.float64).sum()) for a in w], rtol=0, atol=1e-9)
    assert got["backend"] == "tensorflow" and got["keras_version"] == "2.2.4"
    want_layers = ["input_1"] + (["zero_padding2d_1"] if fml else []) + ["separable_conv2d_1", "separable_conv2d_2"] + \
                  (["zero_padding2d_2"] if fml else []) + ["separable_conv2d_3"] + [f"conv2d_{i}" for i in range(1, 8)]
    assert got["layer_names"] == want_layers
    if whole:
        assert got["root_attrs"] == ["backend", "keras_version", "model_config"] and got["model_class"] == "Model"
        assert got["cfg_names"] == want_layers and got["outputs"] == [["conv2d_7", 0, 0]]
        assert set(got["classes"]) <= {"InputLayer", "ZeroPadding2D", "SeparableConv2D", "Conv2D"}


def test_writer_matches_the_layout_h5py_gives_keras(golden_dir, tmp_path):
    """Same tree as the fixture written by real h5py in Keras's layout (tests/golden/make_keras_h5.py): layer groups, weight
    names and attribute names agree; the values written are the values read."""
    from ubdvss_amd import keras_h5_writer
    ref = keras_h5.H5File(os.path.join(golden_dir, "keras_model_rgb.h5"))
    arrays, names = keras_h5.read_keras_weights(os.path.join(golden_dir, "keras_model_rgb.h5"))
    path = str(tmp_path / "model.h5")
    keras_h5_writer.write_keras_model(path, arrays, 3, 0, True, True)
    mine = keras_h5.H5File(path)
    assert sorted(mine.root["model_weights"].keys()) == sorted(ref.root["model_weights"].keys())
    assert keras_h5._names(mine.root["model_weights"].attrs, "layer_names") == keras_h5._names(ref.root["model_weights"].attrs, "layer_names")
    for layer in ref.root["model_weights"]:
        assert keras_h5._names(mine.root["model_weights"][layer].attrs, "weight_names") == keras_h5._names(ref.root["model_weights"][layer].attrs, "weight_names")
    for k in ("keras_version", "backend", "model_config"):
        assert k in mine.root.attrs and k in ref.root.attrs
    again, names2 = keras_h5.read_keras_weights(path)
    assert names2 == names and all(np.array_equal(a, b) for a, b in zip(again, arrays))
