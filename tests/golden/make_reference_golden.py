"""Regenerates the golden fixtures FROM THE REFERENCE ITSELF -- the script that turns "parity unpinned" into "pinned".

It cannot run in the build container or on the GPU box (Keras 2.2 / TensorFlow 1.x / OpenCV 3.4 are not installed and
there is no network) and it is never imported by tests, bench.py or the product.  A maintainer runs it once in an
environment that can import the reference:

    python3.6 -m venv ref && . ref/bin/activate
    pip install "tensorflow==1.13.*" "keras==2.2.4" "opencv-python>=3.4,<4.0" "numpy<1.20" pillow h5py
    python tests/golden/make_reference_golden.py /path/to/ubdvss  /tmp/ref_golden
    UBD_GOLDEN_DIR=/tmp/ref_golden python -m pytest tests -q          # CPU: oracle vs reference;  -m gpu: HIP vs reference

It imports semantic_segmentation.{net,losses,utils,segmap_manager} from the path given on the command line, feeds
them the SAME seeded inputs as tests/golden/make_golden.py (inputs / parameters are read back from the committed
fixtures, so both generators see identical bytes) and writes files with the same names and keys, which the tests
consume through UBD_GOLDEN_DIR (tests/conftest.py).  What it pins, by reference file:line:
  net.py:278-314 (+ :225-252)   logits of the built Keras model for given get_weights()  -> net_*.npz, cfg1_logits.npy
  losses.py:33-126              detection / total loss values and tf.gradients           -> loss_case.npz, manifest
  utils.py:51-60, segmap_manager.py:41-69   cv2 contours -> quads -> class vote            -> manifest post_rect / post_stress
  net.py:248-250 (use_bn=True)  logits of a model built with BatchNormalization layers   -> net_bn_rgb.npz (round 4)
  net.py:443-494 (load_model)   a model.h5 written by ubdvss_amd/keras_h5_writer.py, opened by the REFERENCE's own
                                NetManager.load_model and run: proves the weight interchange back to the reference (round 4)
Only this repo's numpy-only helpers (synthetic.py, oracle/net_numpy.py) are loaded besides the reference, by file
path, to rebuild the seeded inputs; nothing of the reference is copied anywhere.
"""
import argparse
import hashlib
import importlib.util
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def keras_weight_list(flat, model):
    """flat fp32 vector in get_weights() order -> list shaped like model.get_weights()."""
    out, off = [], 0
    for w in model.get_weights():
        out.append(np.asarray(flat[off:off + w.size], np.float32).reshape(w.shape))
        off += w.size
    assert off == flat.size, "parameter count of the Keras model differs from the fixture"
    return out


def build_reference_model(ref_net, log_dir, c_in, n_classes, fml):
    types_file = None
    if n_classes > 0:
        types_file = os.path.join(log_dir, "types_%d.txt" % n_classes)
        with open(types_file, "w") as f:
            f.write("\n".join("c%d" % i for i in range(n_classes)))
    cfg = ref_net.NetConfig(types_file, fml_compatible=bool(fml), grey=(c_in == 1))     # net.py:98-132
    mgr = ref_net.NetManager(log_dir, cfg)
    mgr.build_model()                                                                   # net.py:273-314
    return mgr.get_keras_model()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("reference_root", help="checkout of asmekal/ubdvss (the directory that contains semantic_segmentation/)")
    ap.add_argument("out_dir")
    args = ap.parse_args()
    sys.path.insert(0, os.path.abspath(args.reference_root))
    import keras.backend as K                                                          # noqa: E402
    import tensorflow as tf                                                            # noqa: E402
    from semantic_segmentation import net as ref_net, losses as ref_losses            # noqa: E402
    from semantic_segmentation.segmap_manager import SegmapManager                    # noqa: E402
    synthetic = _load_by_path("ubd_synthetic", os.path.join(REPO, "ubdvss_amd", "synthetic.py"))
    onet = _load_by_path("ubd_onet", os.path.join(REPO, "oracle", "net_numpy.py"))

    os.makedirs(args.out_dir, exist_ok=True)
    with open(os.path.join(HERE, "manifest.json")) as f:
        manifest = json.load(f)
    manifest["generated_by"] = "reference (keras %s, tf %s)" % (__import__("keras").__version__, tf.__version__)

    # ---- cfg1: single 256x256x3 image, forward-only (BASELINE.json configs[0])
    x = synthetic.noise_images(0, 1, 256, 256, 3)
    w = onet.init_weights(1, 3, 0)
    model = build_reference_model(ref_net, args.out_dir, 3, 0, True)
    model.set_weights(keras_weight_list(onet.flatten_weights(w).astype(np.float32), model))
    logits = model.predict(x).astype(np.float32)                                       # model_runner.py:119
    np.save(os.path.join(args.out_dir, "cfg1_logits.npy"), logits)
    manifest["cfg1"] = dict(input_sha256=sha(x), weights_sha256=sha(onet.flatten_weights(w)), logits_sha256=sha(logits),
                            n_positive=int((logits[..., 0] > -0.0).sum()))

    # ---- small network cases: inputs and parameters come from the committed fixtures
    for name in manifest["net_cases"]:
        d = np.load(os.path.join(HERE, "net_%s.npz" % name))
        model = build_reference_model(ref_net, args.out_dir, int(d["c_in"]), int(d["n_classes"]), int(d["fml"]))
        model.set_weights(keras_weight_list(d["params"], model))
        lg = model.predict(d["x"]).astype(np.float32)
        np.savez_compressed(os.path.join(args.out_dir, "net_%s.npz" % name), x=d["x"], params=d["params"], logits=lg,
                            c_in=d["c_in"], n_classes=d["n_classes"], fml=d["fml"])

    # ---- postprocess (segmap_manager.py:41-69 -> utils.py:51-60 -> cv2)
    maps = np.load(os.path.join(HERE, "post_rect.npz"))["maps"].astype(np.int32)
    lg = synthetic.logits_from_maps(maps, 4, seed=5, noise=0.0)
    thr = -np.log(1 / np.clip(0.5, 1e-9, 1 - 1e-9) - 1)                                # model_runner.py:37-38
    out = []
    for i in range(maps.shape[0]):
        det = np.where(lg[i, ..., :1] > thr, 1, 0)                                     # model_runner.py:124
        objs = SegmapManager.postprocess(det, lg[i, ..., 1:], scale=4, min_area_threshold=5)
        out.append(dict(quads=[[int(v) for v in o.bbox] for o in objs], classes=[int(o.object_type) for o in objs]))
    manifest["post_rect"] = out
    shutil.copy(os.path.join(HERE, "post_rect.npz"), args.out_dir)
    stress = np.load(os.path.join(HERE, "post_stress_maps.npy"))
    manifest["post_stress"] = [[[int(v) for v in o.bbox] for o in SegmapManager.postprocess(m[..., None], None, scale=4, min_area_threshold=5)]
                               for m in stress]
    shutil.copy(os.path.join(HERE, "post_stress_maps.npy"), args.out_dir)

    # ---- loss values and gradients (losses.py:33-126 through TF autodiff)
    d = np.load(os.path.join(HERE, "loss_case.npz"))
    yt = d["y_true"].astype(np.float32)[..., None]
    yp_t = K.placeholder(shape=d["y_pred"].shape)
    yp1_t = K.placeholder(shape=d["y_pred"][..., :1].shape)
    yt_t = K.placeholder(shape=yt.shape)
    l_det = ref_losses.detection_loss(yt_t, yp1_t)
    l_all = ref_losses.detection_and_classification_loss(yt_t, yp_t)
    sess = K.get_session()
    v_det, g_det = sess.run([l_det, tf.gradients(l_det, yp1_t)[0]], {yt_t: yt, yp1_t: d["y_pred"][..., :1]})
    v_all, g_all = sess.run([l_all, tf.gradients(l_all, yp_t)[0]], {yt_t: yt, yp_t: d["y_pred"]})
    np.savez_compressed(os.path.join(args.out_dir, "loss_case.npz"), y_true=d["y_true"], y_pred=d["y_pred"],
                        g_det=np.asarray(g_det, np.float32), g_all=np.asarray(g_all, np.float32))
    manifest["loss_case"] = dict(det=float(np.mean(v_det)), total=float(np.mean(v_all)))

    # ---- round 4: the use_bn=True branch of conv_bn (net.py:248-250; the reference's own builder never passes use_bn, so the
    #      graph is built here from the reference's conv_bn with use_bn=True, layer for layer as net.py:292-311 does)
    from keras.layers import Input, Conv2D                                            # noqa: E402
    from keras.models import Model as KModel                                          # noqa: E402
    rng = np.random.default_rng(17)
    inp = Input(shape=(None, None, 3))
    t = ref_net.conv_bn(inp, 24, strides=(2, 2), separable=True, use_strides_compatible_with_fml=True, use_bn=True)
    t = ref_net.conv_bn(t, 24, dilation_rate=1, separable=True, use_bn=True)
    t = ref_net.conv_bn(t, 24, strides=(2, 2), separable=True, use_strides_compatible_with_fml=True, use_bn=True)
    for dil in (1, 2, 4, 8, 16, 1):
        t = ref_net.conv_bn(t, 24, dilation_rate=dil, separable=False, use_bn=True)
    bn_model = KModel(inputs=inp, outputs=Conv2D(1, (1, 1), padding='same', activation=None)(t))
    wbn = []
    for a in bn_model.get_weights():                                                   # get_weights() order = the 59-array list the tests use
        wbn.append(rng.normal(0, 0.3, a.shape).astype(np.float32))
    names = [v.name for v in bn_model.weights]
    for i, nm in enumerate(names):
        if "moving_variance" in nm: wbn[i] = rng.uniform(0.2, 2.0, wbn[i].shape).astype(np.float32)
        if "gamma" in nm: wbn[i] = rng.uniform(0.5, 1.5, wbn[i].shape).astype(np.float32)
    bn_model.set_weights(wbn)
    xb = synthetic.noise_images(61, 2, 96, 128, 3)
    np.savez_compressed(os.path.join(args.out_dir, "net_bn_rgb.npz"), x=xb, logits=bn_model.predict(xb).astype(np.float32),
                        **{"w%02d" % i: a for i, a in enumerate(wbn)})
    manifest["net_bn_rgb"] = dict(weight_names=names)

    # ---- round 4: weight interchange BACK to the reference: a file written by this repo's pure-Python HDF5 writer, opened by the
    #      reference's NetManager.load_model (net.py:443-494 -> keras.models.load_model) and run
    kw = _load_by_path("ubd_keras_h5_writer", os.path.join(REPO, "ubdvss_amd", "keras_h5_writer.py"))
    back_dir = os.path.join(args.out_dir, "written_by_ubdvss_amd")
    os.makedirs(back_dir, exist_ok=True)
    d = np.load(os.path.join(HERE, "net_rgb_det.npz"))
    wl = onet.unflatten_weights(d["params"], 3, 0)
    kw.write_keras_model(os.path.join(back_dir, "inference_model.h5"), wl, 3, 0, bool(d["fml"]), True)
    mgr = ref_net.NetManager(back_dir, ref_net.NetConfig(None, fml_compatible=bool(d["fml"]), grey=False))
    mgr.load_model()
    lg_back = mgr.get_keras_model().predict(d["x"]).astype(np.float32)
    model = build_reference_model(ref_net, args.out_dir, 3, 0, int(d["fml"]))
    model.set_weights(keras_weight_list(d["params"], model))
    assert np.array_equal(lg_back, model.predict(d["x"]).astype(np.float32)), "the reference computes different logits from the written file"
    manifest["h5_written_by_ubdvss_amd_loads_in_reference"] = True

    # files that do not depend on the reference travel unchanged (HDF5 reader fixtures, 16-bit oracle fixtures)
    for fn in os.listdir(HERE):
        if fn.endswith(".h5") or fn.startswith("net16_"):
            shutil.copy(os.path.join(HERE, fn), args.out_dir)
    with open(os.path.join(args.out_dir, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("reference fixtures written to", args.out_dir)


if __name__ == "__main__":
    main()
