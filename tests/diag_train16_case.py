"""Diagnostic (not collected by pytest): per-tensor gradient errors of one 16-bit train-step case against the rounding-aware oracle
evaluated in fp32 and in fp64.  python tests/diag_train16_case.py dtype cin ncls fml n hh ww"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import net_numpy as onet, net_torch as otorch
from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic


def run(dtype, cin, ncls, fml, n, hh, ww):
    cfg = NetConfig(class_names=[f"c{i}" for i in range(ncls)] if ncls else None, grey=(cin == 1), fml_compatible=fml)
    model = Model(cfg, dtype=dtype, seed=0)
    w = onet.init_weights(90 + cin, cin, ncls, bias_scale=0.2)
    w[-2] = (w[-2] * 4).astype(np.float32)
    model.set_weights(w)
    labels = synthetic.rectangle_maps(91, n, hh // 4, ww // 4, n_classes=ncls)
    x = synthetic.textured_images(92, labels, 4, cin).astype(np.float32) / 127.5 - 1.0
    tr = Trainer(model, Adam())
    tr.backward_on_device(torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda())
    g = tr.grads.cpu().numpy().astype(np.float64)
    l4 = tr.loss.cpu().numpy()
    gdt = "bfloat16" if dtype == "bfloat16" else None
    refs = {}
    for odt in (torch.float32, torch.float64):
        loss_ref, _, _, grads_ref = otorch.loss_and_grads(x, labels[..., None], w, ncls > 0, fml, dtype=odt, act_dtype=dtype, grad_dtype=gdt)
        refs[odt] = (loss_ref, grads_ref)
    print(f"{dtype} cin {cin} classes {ncls} fml {fml} {n} x {hh} x {ww}: loss gpu {l4[0]:.6f} oracle32 {refs[torch.float32][0]:.6f} oracle64 {refs[torch.float64][0]:.6f}; "
          f"k {int(l4[3])} n_pos {int(l4[7])} of {n * (hh // 4) * (ww // 4)} pixels")
    off = 0
    for i, (nm, _) in enumerate(onet.weight_shapes(cin, ncls)):
        a, b = refs[torch.float32][1][i].reshape(-1).astype(np.float64), refs[torch.float64][1][i].reshape(-1)
        k = a.size
        e32 = np.linalg.norm(g[off:off + k] - a) / max(np.linalg.norm(a), 1e-30)
        e64 = np.linalg.norm(g[off:off + k] - b) / max(np.linalg.norm(b), 1e-30)
        spread = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
        print(f"  {nm:10s} size {k:5d}  vs fp32-oracle {e32:.2e}  vs fp64-oracle {e64:.2e}  oracle fp32-vs-fp64 {spread:.2e}")
        off += k


if __name__ == "__main__":
    a = sys.argv[1:]
    run(a[0], int(a[1]), int(a[2]), a[3] == "1", int(a[4]), int(a[5]), int(a[6]))
