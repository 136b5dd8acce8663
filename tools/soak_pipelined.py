"""Soak of the pipelined runner (postprocess of batch k inside the stem kernel of batch k + 1) against the serial path on random batch
shapes: batch sizes around the CU count / strip count boundaries, non-square images, with and without classes.  Prints MISMATCHES: n."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, ModelRunner, synthetic
torch.cuda.set_device(0)
rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
bad = 0
cases = int(os.environ.get("CASES", 40))


def diffs(got, ref):
    """Names of the result tensors that differ; lists are compared up to their counts (the pipelined runner reuses its result
    buffers: entries behind a list's end may be stale)."""
    names = ("logits", "map", "quads", "classes", "counts")
    live = torch.arange(ref[2].shape[1], device=ref[2].device)[None, :] < ref[4][:, None]
    out = []
    for nm, a, b in zip(names, got, ref):
        if a is None and b is None:
            continue
        d = a != b
        if nm == "quads": d = d & live[..., None]
        if nm == "classes": d = d & live
        if bool(d.any()):
            out.append(f"{nm} ({int(d.sum())} entries, first at {tuple(int(v) for v in d.nonzero()[0])})")
    return out
for case in range(cases):
    n_cls = int(rng.choice([0, 0, 3]))
    hh, ww = int(rng.choice([128, 192, 256, 320, 512])), int(rng.choice([128, 256, 384, 512]))
    n_img = int(rng.integers(1, 41))
    if n_img * hh * ww > 36 * 512 * 512: n_img = max(1, 36 * 512 * 512 // (hh * ww))
    cfg = NetConfig(class_names=[f"c{i}" for i in range(n_cls)] if n_cls else None, grey=False)
    model = Model(cfg, seed=5 + case)
    serial, piped = ModelRunner(cfg), ModelRunner(cfg, pipelined=True)
    batches = []
    for k in range(4):
        labels = synthetic.rectangle_maps(70 + k + 10 * case, n_img, hh // 4, ww // 4)
        batches.append(torch.from_numpy(synthetic.textured_images(80 + k, labels, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda())
    ref = [[t.clone() if t is not None else None for t in serial.predict_on_device(model, b)] for b in batches]
    got = [piped.predict_on_device(model, b) for b in batches[:1]]
    ok = True
    for k in range(1, 4):
        got.append(piped.predict_on_device(model, batches[k]))
        torch.cuda.synchronize()
        for msg in diffs(got[k - 1], ref[k - 1]):
            ok = False
            print(f"   batch {k - 1}: {msg}", flush=True)
        got[k - 1] = [t.clone() if t is not None else None for t in got[k - 1]]
    piped.synchronize()
    for msg in diffs(got[3], ref[3]):
        ok = False
        print(f"   batch 3: {msg}", flush=True)
    strips_per_cu = n_img * (hh // 16) / model.num_cus          # >= 2: the one-kernel stem (and the in-kernel postprocess) runs
    print(f"case {case}: {n_img} x {hh} x {ww}, classes {n_cls}, strips per CU {strips_per_cu:.2f}: {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += (not ok)
print("MISMATCHES:", bad)
