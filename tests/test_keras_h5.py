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
    arrays = [np.abs(a) + 0.1 if n.endswith("moving_variance:0") else a for a, n in zip(arrays, names)]   # the fixture's noise can be negative
    folded = fold_batchnorm(arrays)
    assert [a.shape for a in folded] == [tuple(s) for s in weight_shapes(3, 0)]
    s = arrays[3] / np.sqrt(arrays[6].astype(np.float64) + 1e-3)
    assert np.allclose(folded[1], arrays[1] * s, rtol=1e-6) and np.allclose(folded[2], (arrays[2] - arrays[5]) * s + arrays[4], rtol=1e-5, atol=1e-6)
    assert np.array_equal(folded[0], arrays[0]) and np.array_equal(folded[-1], arrays[-1])
