"""PCIe-inclusive rate of the reference's real seam (model_runner.py:60-67: ModelRunner.predict batch after batch, numpy in, object lists
out): ModelRunner.predict per batch vs ModelRunner.predict_stream over the same batches (pinned staging ring, copy streams)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, ModelRunner, PreprocessingType, synthetic
torch.cuda.set_device(0)
labels = synthetic.rectangle_maps(3, 32, 128, 128)
imgs = [synthetic.textured_images(4 + k, labels, 4, 3) for k in range(4)]
for name in ("uint8", "float32"):
    cfg = NetConfig(grey=False, preprocessing=PreprocessingType.MOBILENET_LIKE) if name == "uint8" else NetConfig(grey=False)
    model = Model(cfg, seed=1)
    arrs = imgs if name == "uint8" else [a.astype(np.float32) / 127.5 - 1.0 for a in imgs]
    runner = ModelRunner(cfg, pixel_threshold=0.5, max_objects_per_image=1024)
    for _ in range(3): runner.predict(model, arrs[0])
    nb = 24
    t0 = time.perf_counter()
    for k in range(nb): runner.predict(model, arrs[k % 4])
    t_seq = (time.perf_counter() - t0) / nb
    for threads in (2, 4, 8):
        # steady state of a streaming job: a fresh process needs ~100 batches to settle (pinned-allocator cache, clocks), so four untimed
        # 24-batch runs first (bench.py's protocol), then 48 timed batches in one pipeline
        for _ in range(4):
            list(runner.predict_stream(model, [arrs[k % 4] for k in range(24)], copy_threads=threads))
        nb = 48
        t0 = time.perf_counter()
        n = sum(1 for _ in runner.predict_stream(model, (arrs[k % 4] for k in range(nb)), copy_threads=threads))
        t_str = (time.perf_counter() - t0) / nb
        print(f"{name}: predict {t_seq * 1e3:.3f} ms / batch ({32 / t_seq:.0f} img/s)   predict_stream[{threads} copy threads] {t_str * 1e3:.3f} ms / batch ({32 / t_str:.0f} img/s)  "
              f"[{arrs[0].nbytes / 1e6:.1f} MB per batch: {arrs[0].nbytes / t_str / 1e9:.1f} GB/s host -> device]  consumer thread per batch: "
              + ", ".join(f"{k[:-2]} {v / nb * 1e3:.3f} ms" for k, v in runner.last_stream_stats.items() if k.endswith("_s")), flush=True)
