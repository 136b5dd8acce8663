"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

PARITY UNPINNED: the reference (Keras/TF/OpenCV) cannot be imported in the build container,
so these vectors are produced by oracle/ (numpy fp64 network, C restatement of the OpenCV
routines) -- not by the reference itself.  They pin the oracle against regressions and give
the GPU tests fixed inputs/outputs.  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import net_numpy as onet, loss_numpy as oloss, cv_post as ocv  # noqa: E402
from ubdvss_amd import synthetic  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    manifest = {}
    # ---- cfg1 (BASELINE.json configs[0]): single 256x256x3 image, forward-only map
    x = synthetic.noise_images(0, 1, 256, 256, 3)
    w = onet.init_weights(1, 3, 0)
    logits = onet.forward(x.astype(np.float64), w).astype(np.float32)
    np.save(os.path.join(HERE, "cfg1_logits.npy"), logits)
    manifest["cfg1"] = dict(input_sha256=sha(x), weights_sha256=sha(onet.flatten_weights(w)),
                            logits_sha256=sha(logits), n_positive=int((logits[..., 0] > -0.0).sum()))
    # ---- small network cases with biases / classes / grey / non-fml
    cases = []
    for name, (cin, ncls, fml, hh, ww) in {
        "rgb_det": (3, 0, True, 64, 64), "grey_cls3": (1, 3, True, 64, 128),
        "rgb_cls2_nofml": (3, 2, False, 128, 64)}.items():
        x = synthetic.noise_images(10 + len(cases), 2, hh, ww, cin)
        w = onet.init_weights(20 + len(cases), cin, ncls, bias_scale=0.2)
        lg = onet.forward(x.astype(np.float64), w, fml).astype(np.float32)
        np.savez_compressed(os.path.join(HERE, f"net_{name}.npz"), x=x, params=onet.flatten_weights(w).astype(np.float32),
                            logits=lg, c_in=cin, n_classes=ncls, fml=int(fml))
        cases.append(name)
    manifest["net_cases"] = cases
    # ---- 16-bit activation variants (configs[2] bf16, configs[4] fp16): oracle with the SAME storage roundings
    x = synthetic.noise_images(40, 2, 64, 96, 3)
    w = onet.init_weights(41, 3, 2, bias_scale=0.2)
    ref64 = onet.forward(x.astype(np.float64), w).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "net16_rgb_cls2.npz"), x=x, params=onet.flatten_weights(w).astype(np.float32),
                        logits_f64=ref64, c_in=3, n_classes=2, fml=1,
                        logits_bfloat16=onet.forward(x.astype(np.float64), w, act_dtype="bfloat16").astype(np.float32),
                        logits_float16=onet.forward(x.astype(np.float64), w, act_dtype="float16").astype(np.float32))
    # ---- the use_bn=True branch (net.py:248-250): 59 arrays in get_weights() order (conv arrays, gamma, beta, moving mean, moving
    #      variance per hidden layer; head last) -- same file name and keys as tests/golden/make_reference_golden.py writes
    rng = np.random.default_rng(17)
    wb = onet.init_weights(63, 3, 0, bias_scale=0.3)
    wbn, i = [], 0
    for n_conv in (3, 3, 3, 2, 2, 2, 2, 2, 2):
        wbn += wb[i:i + n_conv]
        wbn += [rng.uniform(0.5, 1.5, 24).astype(np.float32), rng.normal(0, 0.3, 24).astype(np.float32),
                rng.normal(0, 0.5, 24).astype(np.float32), rng.uniform(0.2, 2.0, 24).astype(np.float32)]
        i += n_conv
    wbn += wb[i:]
    xb = synthetic.noise_images(61, 2, 96, 128, 3)
    np.savez_compressed(os.path.join(HERE, "net_bn_rgb.npz"), x=xb, logits=onet.forward_bn(xb.astype(np.float64), wbn).astype(np.float32),
                        **{"w%02d" % k: a for k, a in enumerate(wbn)})
    # ---- postprocess: rectangle maps -> quads
    maps = synthetic.rectangle_maps(3, 8, 128, 128, n_classes=4)
    lg = synthetic.logits_from_maps(maps, 4, seed=5, noise=0.0)     # deterministic: rebuilt from the maps by the tests
    det, cl, found = ocv.predict_postprocess(lg, 4, 0.5, 4, 5)
    np.savez_compressed(os.path.join(HERE, "post_rect.npz"), maps=maps.astype(np.uint8))
    manifest["post_rect"] = [dict(quads=q.tolist(), classes=c.tolist()) for q, c in found]
    # stress maps: all ones, Bernoulli(0.5), ring with nested blob
    rng = np.random.default_rng(7)
    stress = np.zeros((3, 64, 64), np.uint8)
    stress[0] = 1
    stress[1] = rng.random((64, 64)) < 0.5
    stress[2, 8:56, 8:56] = 1; stress[2, 16:48, 16:48] = 0; stress[2, 24:40, 24:40] = 1
    np.save(os.path.join(HERE, "post_stress_maps.npy"), stress)
    manifest["post_stress"] = [ocv.postprocess(m, None, 4, 5)[0].tolist() for m in stress]
    # ---- loss
    rng = np.random.default_rng(11)
    yt = synthetic.rectangle_maps(12, 2, 32, 32, n_classes=3)
    yp = rng.normal(0, 2, (2, 32, 32, 4)).astype(np.float32)
    yp[0, 0, 0, 0] = 30.0; yp[0, 0, 1, 0] = -30.0          # exercise the Keras clip
    l_det, g_det = oloss.total_loss(yt[..., None], yp[..., :1], False)
    l_all, g_all = oloss.total_loss(yt[..., None], yp, True)
    np.savez_compressed(os.path.join(HERE, "loss_case.npz"), y_true=yt.astype(np.uint8), y_pred=yp,
                        g_det=g_det.astype(np.float32), g_all=g_all.astype(np.float32))
    manifest["loss_case"] = dict(det=float(l_det), total=float(l_all))
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("golden fixtures written:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
