"""In-kernel phase timing of dilconv16s_kernel (diagnostic build, tools/build_diag.sh): s_memtime of every wave at the phase boundaries of
its first 8 items; bf16 train step at batch 64 (TRAIN=1, default) or the cfg5 forward (8 x 1024 x 1024 fp16).  Usage: stamps_d16s.py [dilation ...]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ubdvss_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_ab", os.environ.get("DIAG_LIB", "libubd_hip_diag.so"))
from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic
torch.cuda.set_device(0)
lib = _lib.load()
if os.environ.get("TRAIN", "1") == "1":
    m = Model(NetConfig(grey=False), dtype="bfloat16", seed=1)
    tr = Trainer(m, Adam())
    lab = synthetic.rectangle_maps(30, 64, 128, 128)
    x = torch.from_numpy(synthetic.textured_images(31, lab, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
    y = torch.from_numpy(lab).cuda()
    run = lambda: tr.train_step_on_device(x, y)
    what = "bf16 train step, 64 x 128 x 128 maps"
else:
    m = Model(NetConfig(grey=False), dtype="float16", seed=1)
    x = torch.from_numpy(synthetic.noise_images(2, 8, 1024, 1024, 3)).cuda()
    run = lambda: m.predict_on_device(x)
    what = "cfg5 fp16 forward, 8 x 256 x 256 maps"
for _ in range(100): run()
lib.ubd_debug_set_stamps_d16s.argtypes = [ctypes.c_void_p, ctypes.c_int]; lib.ubd_debug_set_stamps_d16s.restype = None
names = ["wait for the tile (vmcnt)", "barrier", "decode + issue the next tile's 4 DMA pieces", "four rows: 28 reads, 56 MFMAs, epilogues, 8 stores"]
for d in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
    st = torch.zeros((768, 4, 8, 8), dtype=torch.int64, device="cuda")
    lib.ubd_debug_set_stamps_d16s(st.data_ptr(), d)
    run(); torch.cuda.synchronize()
    lib.ubd_debug_set_stamps_d16s(None, 0)
    s = st.cpu().numpy()
    used = s[:, 0, 1, 0] > 0
    s = s[used]
    seg = np.diff(s[:, :, 1:5, :5], axis=-1)
    period = s[:, 0, 2:5, 0] - s[:, 0, 1:4, 0]
    loop = s[:, :, 2:5, 0] - s[:, :, 1:4, 4]
    print(f"{what}, dilation {d}: blocks with >= 2 items {used.sum()}, item period median {np.median(period):.0f} cycles")
    t0 = s[:, :, 1, 5].min()                                    # 100-MHz clock shared by all CUs: the first block's entry
    ent, pre, post = (s[:, 0, 1, 5] - t0) / 100.0, (s[:, 0, 1, 6] - t0) / 100.0, (s[:, 0, 1, 7] - t0) / 100.0
    print(f"  block time line (us after the first block's entry): entry median {np.median(ent):.1f} p90 {np.percentile(ent, 90):.1f} max {ent.max():.1f};  item loop starts {np.median(pre):.1f} "
          f"(prologue {np.median(s[:, 0, 0, 6] - s[:, 0, 0, 5]):.0f} cycles);  loop ends median {np.median(post):.1f} p10 {np.percentile(post, 10):.1f} max {post.max():.1f}")
    print(f"  first item: wait for the tile {np.median(s[:, 0, 0, 1] - s[:, 0, 0, 0]):.0f} cycles, barrier {np.median(s[:, 0, 0, 2] - s[:, 0, 0, 1]):.0f}, rows {np.median(s[:, 0, 0, 4] - s[:, 0, 0, 3]):.0f}")
    for w in range(4):
        print(f"  wave {w}: " + "  ".join(f"{names[k]}: {np.median(seg[:, w, :, k]):.0f}" for k in range(4)) + f"  loop: {np.median(loop[:, w]):.0f}")
