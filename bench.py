"""bench.py -- headline benchmark of the ubdvss hot path on MI355X.

Metric (BASELINE.json): images/sec at 512x512 for forward + CCL postprocess.
Workload at every N: BASELINE.json configs[1] per GPU -- batch=32 512x512x3 fp32 forward + device
postprocess (threshold -> external components -> quads).  A "step" = one pass of that path over one
synthetic batch already resident in HBM.  Multi-GPU: images shard by rank with no data-path
collective (replicas), weak scaling (32 images per GPU), value = all ranks' images / max-over-ranks
time.

Extra objects on the JSON line:
  roofline     -- the dominant kernel (dense dilated 3x3 conv, fp32 MFMA), timed live with HIP events
  cpu_baseline -- the torch-CPU restatement (oracle/, kind "port") on a bounded sample, rank 0, N=1
  parts        -- net-only / postprocess-only timings of the same step
Launch: python bench.py [--gpus N --steps K --warmup W]; for N>1 under torch.distributed.run.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH, SIDE, C_IN = 32, 512, 3
BYTES_PER_IMAGE_FP32 = 50.40e6            # layer-wise compulsory traffic, 512x512x3, n_cls = 0
FLOP_PER_IMAGE = 1.1627e9
PEAK_MFMA_F32 = 157.3                     # TFLOP/s, MI355X_MICROARCH.md (f32-input MFMA = vector peak)
PEAK_HBM = 8000.0                         # GB/s
SETTLE_STEPS = 300                        # untimed steps (settle + warm-up) before any timed region: clock ramp after idle


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-train", action="store_true", help="skip the extra train-step measurement")
    ap.add_argument("--train-batch", type=int, default=64)
    return ap.parse_args()


def cpu_baseline(seconds):
    """torch-CPU restatement of the same path (forward fp32 + C restatement of the OpenCV
    postprocess) on a bounded sample: batches of 4 images 512x512x3, all host threads."""
    from oracle import net_numpy as onet, net_torch as otorch, cv_post as ocv
    from ubdvss_amd import synthetic
    w = onet.init_weights(1, C_IN, 0)
    tw = otorch.to_torch_weights(w, torch.float32)
    nb = 4
    labels = synthetic.rectangle_maps(3, nb, SIDE // 4, SIDE // 4)
    x = torch.from_numpy(synthetic.textured_images(4, labels, 4, C_IN).astype(np.float32) / 127.5 - 1.0)

    def step():
        with torch.no_grad():
            lg = otorch.forward(x, tw).numpy()
        det = (lg[..., 0] > -0.0).astype(np.uint8)
        return [ocv.postprocess(det[i], None, 4, 5) for i in range(nb)]

    # thread count: oneDNN on these small convolutions does not scale to hundreds of host threads, so
    # probe a few counts briefly and time the sample with the fastest (reported as "cores")
    ncpu = os.cpu_count() or 1
    best = (None, 1e30)
    for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64, ncpu)}):
        torch.set_num_threads(nt)
        step()
        t0 = time.perf_counter(); step(); dt = time.perf_counter() - t0
        if dt < best[1]:
            best = (nt, dt)
        if dt > 4.0:
            break
    torch.set_num_threads(best[0])
    step()
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 2000:
            break
    cpu_model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(nb * n / el, 2), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model, "host_cpus": ncpu,
            "sample": f"{n} batches of {nb} textured 512x512x3 images, torch-CPU fp32 forward (oneDNN) + C "
                      f"restatement of the OpenCV postprocess, {el:.1f} s"}


def cpu_baseline_train(seconds, threads):
    """torch-CPU restatement of the train step (fp32 forward + loss + autograd backward + Keras Adam) at the reference's
    default batch of 8 (train.py:31), bounded sample."""
    from oracle import net_numpy as onet, net_torch as otorch
    from ubdvss_amd import synthetic
    nb = 8
    w = onet.init_weights(1, C_IN, 0)
    labels = synthetic.rectangle_maps(30, nb, SIDE // 4, SIDE // 4)
    x = synthetic.textured_images(31, labels, 4, C_IN).astype(np.float32) / 127.5 - 1.0
    flat = np.concatenate([a.reshape(-1) for a in w]).astype(np.float64)
    m = np.zeros_like(flat); v = np.zeros_like(flat)
    torch.set_num_threads(threads)

    def step(t):
        nonlocal flat, m, v
        _, _, _, grads = otorch.loss_and_grads(x, labels[..., None], onet.unflatten_weights(flat.astype(np.float32), C_IN, 0), False, True,
                                               dtype=torch.float32)
        g = np.concatenate([a.reshape(-1) for a in grads]).astype(np.float64)
        flat, m, v = otorch.adam_step(flat, g, m, v, t)

    step(1)
    t0 = time.perf_counter()
    n = 0
    while True:
        n += 1
        step(1 + n)
        el = time.perf_counter() - t0
        if el >= seconds or n >= 200:
            break
    return {"value": round(nb * n / el, 2), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{n} steps of batch {nb} (512x512x3), torch-CPU fp32 forward + loss + autograd backward + Adam, {el:.1f} s"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("UBD_BENCH_FORCE_DIST") == "1":      # the second: one-rank check of the RCCL path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device(f"cuda:{torch.cuda.current_device()}")

    from ubdvss_amd import NetConfig, Model, ModelRunner, Trainer, Adam, synthetic, _lib
    cfg = NetConfig(grey=False)
    model = Model(cfg, seed=1)                               # glorot-uniform random init, zero biases
    runner = ModelRunner(cfg, pixel_threshold=0.5, max_objects_per_image=1024, pipelined=True)

    # synthetic batch resident in HBM: stripe-textured rectangles on noise
    labels = synthetic.rectangle_maps(3 + rank, BATCH, SIDE // 4, SIDE // 4)
    x = torch.from_numpy(synthetic.textured_images(4 + rank, labels, 4, C_IN).astype(np.float32) / 127.5 - 1.0).to(dev)
    rect_logits = torch.from_numpy(synthetic.logits_from_maps(labels, 0, seed=5)).to(dev)

    def step():
        return runner.predict_on_device(model, x)

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The chip needs a few hundred back-to-back steps after an idle gap before its clocks settle (20 steps: 0.54-0.60 ms,
    # 500: 0.50 ms per step): untimed settle steps in front of the W warm-up steps keep `value` independent
    # of the K / W the caller picks.  Reported as "clock_settle_steps".
    settle = max(0, SETTLE_STEPS - args.warmup)
    for _ in range(settle):
        step()
    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * BATCH * args.steps / elapsed

    # ---- extra: train step (fwd + loss + bwd + gradient all-reduce + Adam), batch/GPU = --train-batch:
    #      configs[2] / configs[3] name bf16 activations (fp32 master weights, fp32 accumulation); the fp32
    #      variant is measured beside it
    def time_train(dtype, n_cls=0):
        tb = args.train_batch
        tlabels = synthetic.rectangle_maps(30 + rank, tb, SIDE // 4, SIDE // 4, n_classes=n_cls)
        tx = torch.from_numpy(synthetic.textured_images(31 + rank, tlabels, 4, C_IN).astype(np.float32) / 127.5 - 1.0).to(dev)
        ty = torch.from_numpy(tlabels).to(dev)
        tcfg = cfg if n_cls == 0 else NetConfig(class_names=[f"class{i}" for i in range(n_cls)], grey=False)
        tmodel = Model(tcfg, dtype=dtype, seed=1)
        trainer = Trainer(tmodel, Adam(lr=1e-3))
        trainer.broadcast_weights()
        for _ in range(max(1, args.warmup) + max(0, SETTLE_STEPS - args.warmup)):
            trainer.train_step_on_device(tx, ty)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            trainer.train_step_on_device(tx, ty)
        sync_all()
        tel = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([tel], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tel = float(t.item())
        bpe = 4.0 if dtype == "float32" else 2.0
        e_fwd = SIDE * SIDE * (C_IN + 45 + (1 + n_cls) / 16.0)                    # E_fwd elements per image
        train_bytes_per_image = (3 * e_fwd - SIDE * SIDE * C_IN) * bpe             # E_train = 3 E_fwd - H W C_in
        res = {"metric": "images/sec (512x512) train step", "value": round(world * tb * args.steps / tel, 1),
               "unit": "images/s", "ms_per_step": round(tel / args.steps * 1e3, 4), "batch_per_gpu": tb,
               "dtype": {"float32": "f32", "bfloat16": "bf16"}[dtype], "n_classes": n_cls,
               "parallelism": f"dp{world}: per-replica loss, one flat-gradient all-reduce (RCCL) per step" if world > 1 else "single GPU",
               "loss_last": round(float(trainer.loss[0]), 5),
               "hbm_frac_algorithmic": round(tb * train_bytes_per_image / (tel / args.steps) / 1e9 / PEAK_HBM, 4)}
        del trainer, tmodel, tx, ty
        torch.cuda.empty_cache()
        return res

    train = train_f32 = train_cls8 = None
    if not args.no_train:
        train = time_train("bfloat16")
        train_cls8 = time_train("bfloat16", n_cls=8)       # configs[2], second run: 8 classes (labels 1..8), detection + classification loss
        train_f32 = time_train("float32")

    # ---- extra: configs[4], batch 8 of 1024x1024x3, fp16 activations, forward only ("HBM-bound roofline run")
    def time_cfg5():
        n5, side5 = 8, 1024
        m5 = Model(cfg, dtype="float16", seed=1)
        x5 = torch.from_numpy(synthetic.noise_images(7 + rank, n5, side5, side5, C_IN)).to(dev)
        for _ in range(max(2, args.warmup)):
            m5.predict_on_device(x5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps5 = max(5, min(args.steps, 200))
        e0.record()
        for _ in range(reps5):
            m5.predict_on_device(x5)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps5
        bytes_img = side5 * side5 * C_IN * 4.0 + side5 * side5 * 45.0 * 2.0 + (side5 // 4) ** 2 * 4.0   # fp32 image in, fp16 activations (SURVEY 8(d) element count), fp32 logits out
        res = {"workload": "configs[4]: batch=8 1024x1024x3 fp16 forward, dilations {1,2,4,8,16,1}", "ms_per_batch": round(ms, 4),
               "images_per_s": round(n5 / ms * 1e3, 1), "dtype": "f16",
               "roofline": {"bound": "hbm", "achieved": round(n5 * bytes_img / (ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM, "unit": "GB/s",
                            "frac": round(n5 * bytes_img / (ms * 1e-3) / 1e9 / PEAK_HBM, 4),
                            "note": "algorithmic bytes: fp32 image read + every 16-bit activation written once and read once + fp32 logits"}}
        del m5, x5
        torch.cuda.empty_cache()
        return res

    cfg5 = None
    if not args.no_train and rank == 0:
        cfg5 = time_cfg5()

    line = None
    if rank == 0:
        # ---- parts: net only / postprocess only (rectangle maps) -- HIP events on the launch stream
        def timed(fn, reps):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        reps = max(5, min(args.steps, 200))
        net_ms = timed(lambda: model.predict_on_device(x), reps)
        logits = model.predict_on_device(x)
        post_ms = timed(lambda: model.postprocess_on_device(logits, runner.logit_threshold, 4, 5, cap=1024), reps)
        post_rect_ms = timed(lambda: model.postprocess_on_device(rect_logits, runner.logit_threshold, 4, 5, cap=1024), reps)
        counts = out[4].cpu().numpy()

        # ---- roofline of the dominant kernel: one dense dilated layer on the real L3 activations
        lib = _lib.load()
        n, mh, mw = BATCH, SIDE // 4, SIDE // 4
        act_in = torch.rand((n, mh, mw, 24), device=dev) - 0.3
        act_out = torch.empty_like(act_in)
        ws = torch.empty(int(lib.ubd_forward_workspace_bytes(model._h, 1, 4, 4)), dtype=torch.uint8, device=dev)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.ubd_pack_weights(model._h, model.params.data_ptr(), ws.data_ptr(), ws.numel(), stream), "pack")
        layer_ms = []
        for layer in range(6):
            layer_ms.append(timed(lambda: _lib.check(lib.ubd_dilated_layer(
                model._h, model.params.data_ptr(), layer, act_in.data_ptr(), act_out.data_ptr(), n, mh, mw,
                ws.data_ptr(), stream), "dil"), reps))
        t_layer = float(np.mean(layer_ms)) * 1e-3
        flop_layer = 2.0 * 216 * 24 * n * mh * mw
        bytes_layer = 2.0 * n * mh * mw * 24 * 4
        # HBM traffic of one launch from the PMC run committed under profiles/ (separate --pmc passes of
        # tools/gpu_pmc.sh; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 wide reads)
        traffic = None
        try:
            vals = {}
            for ln in open(os.path.join(ROOT, "profiles", "r01_pmc_dilconv_wino.txt")):
                parts = ln.split()
                if len(parts) >= 2:
                    vals[parts[0]] = float(parts[1])
            traffic = round((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0 / 1e6, 1)     # MB per launch
        except Exception:
            traffic = None
        roofline = {"bound": "mfma", "kernel": "dilconv_wino_kernel<0> (Winograd F(2x2,3x3) fp32 MFMA; FLOPs counted as direct conv)", "achieved": round(flop_layer / t_layer / 1e12, 3),
                    "peak": PEAK_MFMA_F32, "unit": "TFLOP/s", "frac": round(flop_layer / t_layer / 1e12 / PEAK_MFMA_F32, 4),
                    "traffic": traffic, "traffic_unit": "MB/launch (PMC, profiles/r01_pmc_dilconv_wino.txt; algorithmic 100.7 MB)", "avg_launch_us": round(t_layer * 1e6, 2),
                    "per_dilation_us": [round(v * 1e3, 2) for v in layer_ms],
                    "algorithmic_gbps": round(bytes_layer / t_layer / 1e9, 1)}
        fwd_hbm = {"bound": "hbm", "achieved": round(BATCH * BYTES_PER_IMAGE_FP32 / (net_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM,
                   "unit": "GB/s", "frac": round(BATCH * BYTES_PER_IMAGE_FP32 / (net_ms * 1e-3) / 1e9 / PEAK_HBM, 4),
                   "note": "whole forward pass, algorithmic bytes 50.40 MB/image (SURVEY 8(d))",
                   "mfma_frac": round(BATCH * FLOP_PER_IMAGE / (net_ms * 1e-3) / 1e12 / PEAK_MFMA_F32, 4)}

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.cpu_seconds)
            if train is not None:
                train["cpu_baseline"] = cpu_baseline_train(max(4.0, args.cpu_seconds * 0.75), cpu["cores"])

        line = {
            "metric": "images/sec (512x512) fwd+CCL", "value": round(value, 1), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "clock_settle_steps": max(0, SETTLE_STEPS - args.warmup),
            "config": {"workload": "configs[1]: batch=32 512x512x3 fp32 forward + CCL postprocess per GPU "
                                   "(stripe-textured rectangle images, random-init weights)",
                       "batch_per_gpu": BATCH, "image": [SIDE, SIDE, C_IN], "parallelism": f"replicas x{world}, no collective"},
            "roofline": roofline, "roofline_forward_pass": fwd_hbm, "cpu_baseline": cpu, "train_step": train, "train_step_8_classes": train_cls8, "train_step_f32": train_f32, "forward_fp16_cfg5": cfg5,
            "parts": {"net_ms": round(net_ms, 4), "postprocess_ms_on_net_maps": round(post_ms, 4),
                      "postprocess_ms_on_rectangle_maps": round(post_rect_ms, 4),
                      "objects_found_mean": float(counts.mean()), "objects_found_max": int(counts.max())},
        }
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
