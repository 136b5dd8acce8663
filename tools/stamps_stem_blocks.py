"""Time line of the blocks of stem123_kernel with the postprocess of k earlier maps inside (diagnostic build, tools/build_diag.sh):
s_memrealtime (100 MHz, shared by all CUs) at kernel entry, after the postprocess job, at the start of every strip and at the end."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ubdvss_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_ab", "libubd_hip_diag.so")
from ubdvss_amd import NetConfig, Model, synthetic
torch.cuda.set_device(0)
lib = _lib.load()
cfg = NetConfig(grey=False)
m = Model(cfg, seed=1)
nimg = int(os.environ.get("N", 32))
labs = synthetic.rectangle_maps(3, nimg, 128, 128)
x = torch.from_numpy(synthetic.textured_images(4, labs, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
out = torch.empty((nimg, 128, 128, 1), device="cuda")
prev = m.predict_on_device(x).clone()
nblk = m.num_cus
lib.ubd_debug_set_stamps.argtypes = [ctypes.c_void_p]; lib.ubd_debug_set_stamps.restype = None
for k in (0, 1, 32):
    job = None
    if k:
        lg = prev[:k].contiguous()
        outs = m.alloc_postprocess_outputs(k, 128, 128, 1024)
        job = {"logits": lg, "logit_threshold": 0.0, "scale": 4, "min_area": 5, "cap": 1024, "outputs": outs}
    for _ in range(100): m.predict_on_device(x, out=out, postprocess=job)
    st = torch.zeros(nblk * 8 * 16 * 8 + nblk * 32, dtype=torch.int64, device="cuda")
    lib.ubd_debug_set_stamps(st.data_ptr())
    m.predict_on_device(x, out=out, postprocess=job)
    torch.cuda.synchronize()
    lib.ubd_debug_set_stamps(None)
    b = st.cpu().numpy().astype(np.int64)[nblk * 8 * 16 * 8:].reshape(nblk, 32)
    t0 = b[:, 0].min()
    us = lambda v: (v - t0) / 100.0
    nstr = (b[:, 4:28] > 0).sum(1)
    print(f"--- {nimg} images, postprocess of {k} maps inside: kernel span {us(b[:, 2].max()):.1f} us; strips per block: "
          + ", ".join(f"{c} x {int((nstr == c).sum())}" for c in sorted(set(nstr))))
    for name, sel in (("postprocess blocks", np.arange(nblk) < k), ("other blocks", np.arange(nblk) >= k)):
        if not sel.any(): continue
        print(f"  {name:18s} entry {us(b[sel, 0]).mean():6.1f}  job done {us(b[sel, 1]).mean():6.1f}  first strip {us(b[sel, 4]).mean():6.1f}  "
              f"end mean {us(b[sel, 2]).mean():6.1f} max {us(b[sel, 2]).max():6.1f}  strips mean {nstr[sel].mean():.2f}")
    o = (np.arange(nblk) >= k) & (nstr > 0)
    print("  prologue of the blocks without a job (us after entry): ticket + ring barrier %.2f, patch requested + tables filled %.2f, patch landed + barrier %.2f, first L1 phase + barrier %.2f"
          % tuple(((b[o, j] - b[o, 0]) / 100.0).mean() for j in (3, 28, 29, 30)))
    d = np.diff(b[:, 4:28], axis=1)
    d = d[(b[:, 5:28] > 0)]
    print(f"  strip period: median {np.median(d) / 100:.1f} us, p10 {np.percentile(d, 10) / 100:.1f}, p90 {np.percentile(d, 90) / 100:.1f}")
    last = np.array([b[i, 4 + nstr[i] - 1] for i in range(nblk) if nstr[i] > 0])
    print(f"  last strip starts: min {us(last.min()):.1f} median {us(np.median(last)):.1f} max {us(last.max()):.1f}")
