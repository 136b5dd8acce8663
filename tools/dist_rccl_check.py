"""THE multi-GPU readiness command for a node with 2..8 MI355X (train.py:176-188 under data parallelism):

    python tools/dist_rccl_check.py                  # runs itself at 2, 4 and 8 ranks (as many as the node has GPUs), one process per GPU
    python tools/dist_rccl_check.py --ranks 8        # one world size
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/dist_rccl_check.py     # under a launcher

Per world size, configs[3]'s per-GPU shape (bf16 activations, 64 x 512 x 512 x 3 per rank; DIST_CHECK_BATCH / DIST_CHECK_SIDE shrink it):
  1. CORRECTNESS of the cross-stream ordering over REAL RCCL.  The dilated + head segment of the gradient vector is all-reduced on the
     handle's communication stream under the stem backward (comm.hip, backward.hip: ubd_comm_begin_tail behind the kernel that makes the
     segment final, ubd_comm_finish before Adam).  With the bf16 step's fixed-order reductions the summed gradients must be BIT-EQUAL between
       (a) fused communication + chained partial-sum reduction (the default),
       (b) fused communication + UBD_REDUCE=batched,
       (c) no communicator in the handle: local gradients, then ONE torch.distributed all-reduce of the whole vector,
     on EVERY one of DIST_CHECK_STEPS (default 200) steps -- a missing event wait shows up as a stale or half-summed segment on some step --
     and the replicas' parameters must be identical on all ranks after the steps (checksum min == max over ranks, every mode).
  2. TIMING (HIP events on the compute stream, MAX over ranks, median of 5 blocks of 40 steps after 100 settle steps):
       step_ms_fused      the product path: ubd_train_step with its communicator
       step_ms_explicit   local gradients + one torch.distributed all-reduce + Adam (the collective fully exposed)
       step_ms_local      no collective at all (what one GPU does alone on its shard; parameters of the ranks diverge, timing only)
       allreduce_us       one 33 028-float all-reduce by itself, back to back (latency-bound: 132 KB)
     exposed = fused - local (what the collective still costs the step), hidden = allreduce_us - exposed.
Rank 0 prints one JSON object per world size and "DIST_RCCL_CHECK OK" / "FAILED"; exit code 0 only if every world size passed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worlds_for(n_gpus, asked=None):
    """world sizes to run on a node with n_gpus devices: the one asked for, else 2, 4, 8 as far as the devices go"""
    if asked:
        if asked < 2: raise SystemExit("dist_rccl_check: --ranks must be >= 2")
        if asked > n_gpus: raise SystemExit(f"dist_rccl_check: --ranks {asked} asked for, {n_gpus} GPU(s) visible")
        return [asked]
    ws = [w for w in (2, 4, 8) if w <= n_gpus]
    if not ws: raise SystemExit(f"dist_rccl_check: needs >= 2 GPUs on the node, {n_gpus} visible (the one-GPU stand-in is tests/test_gpu_comm_loopback.py)")
    return ws


def launch(argv):
    import argparse
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=None)
    ap.add_argument("--port", type=int, default=29533)
    args = ap.parse_args(argv)
    rc_all = 0
    for k, w in enumerate(worlds_for(torch.cuda.device_count(), args.ranks)):          # device_count does not initialise the GPU
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={w}", "--master-addr", "127.0.0.1",
               "--master-port", str(args.port + k), os.path.abspath(__file__)]
        print("dist_rccl_check:", " ".join(cmd), flush=True)
        rc = subprocess.call(cmd, env=env)                                             # a child process: nothing here has touched the GPU
        rc_all = rc_all or rc
    print("DIST_RCCL_CHECK", "OK (all world sizes)" if rc_all == 0 else "FAILED", flush=True)
    return rc_all


def worker():
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, ROOT)
    from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic, distributed
    rank, world, local = distributed.init_from_env("nccl")
    assert world >= 2, "run under torch.distributed.run with --nproc-per-node >= 2"
    dev = torch.device(f"cuda:{local}")
    steps = int(os.environ.get("DIST_CHECK_STEPS", "200"))
    batch, side = int(os.environ.get("DIST_CHECK_BATCH", "64")), int(os.environ.get("DIST_CHECK_SIDE", "512"))
    cfg = NetConfig(grey=False)
    labels = synthetic.rectangle_maps(100 + rank, batch, side // 4, side // 4)           # every rank its own shard
    x = torch.from_numpy(synthetic.textured_images(200 + rank, labels, 4, 3).astype(np.float32) / 127.5 - 1.0).to(dev)
    y = torch.from_numpy(labels).to(dev)

    def make(mode):
        os.environ.pop("UBD_REDUCE", None)
        if mode == "batched": os.environ["UBD_REDUCE"] = "batched"       # read when the handle is created
        m = Model(cfg, dtype="bfloat16", seed=7)
        if mode in ("chained", "batched"): distributed.attach_native_comm(m, fused=True)
        tr = Trainer(m, Adam(lr=1e-3))
        tr.broadcast_weights()
        return tr

    trs = {mode: make(mode) for mode in ("chained", "batched", "unfused")}
    os.environ.pop("UBD_REDUCE", None)
    bad = 0
    for s in range(steps):
        grads = {}
        for mode, tr in trs.items():
            tr.train_step_on_device(x, y)                                # unfused: Trainer all-reduces through torch.distributed
            grads[mode] = tr.grads.clone()
        torch.cuda.synchronize()
        if not (torch.equal(grads["chained"], grads["batched"]) and torch.equal(grads["chained"], grads["unfused"])):
            bad += 1
            if bad <= 3:
                d = (grads["chained"] - grads["unfused"]).abs()
                print(f"rank {rank} step {s}: gradients differ (max |chained - unfused| = {float(d.max()):.3e} at {int(d.argmax())}, "
                      f"chained == batched: {torch.equal(grads['chained'], grads['batched'])})", flush=True)
    # the ranks must also agree with each other: parameter checksum min == max
    for mode, tr in trs.items():
        cs = tr.model.params.double().sum().reshape(1)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if float(lo) != float(hi):
            bad += 1
            print(f"rank {rank}: parameters of mode {mode} differ between ranks ({float(lo)} vs {float(hi)})", flush=True)

    # ---- timing
    def timed(fn, settle=100, blocks=5, per=40):
        for _ in range(settle): fn()
        out = []
        for _ in range(blocks):
            dist.barrier(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(per): fn()
            e1.record(); torch.cuda.synchronize()
            out.append(distributed.max_over_ranks(e0.elapsed_time(e1) / per, device=dev))
        return float(np.median(out))

    local_model = Model(cfg, dtype="bfloat16", seed=7)
    local_tr = Trainer(local_model, Adam(lr=1e-3), process_group=False)      # False: no collective (Trainer treats it as a single process)
    res = {"world": world, "steps": steps, "batch_per_gpu": batch, "global_batch": batch * world, "side": side,
           "bit_equal_fused_vs_explicit_on_every_step": bad == 0,
           "step_ms_fused": round(timed(lambda: trs["chained"].train_step_on_device(x, y)), 4),
           "step_ms_explicit": round(timed(lambda: trs["unfused"].train_step_on_device(x, y)), 4),
           "step_ms_local": round(timed(lambda: local_tr.train_step_on_device(x, y)), 4)}
    g = trs["unfused"].grads.clone()
    res["allreduce_us"] = round(timed(lambda: dist.all_reduce(g), settle=200, per=200) * 1e3, 1)
    res["exposed_comm_us"] = round((res["step_ms_fused"] - res["step_ms_local"]) * 1e3, 1)
    res["hidden_comm_us"] = round(res["allreduce_us"] - res["exposed_comm_us"], 1)
    res["images_per_s_fused"] = round(batch * world / (res["step_ms_fused"] * 1e-3), 1)
    t = torch.tensor([bad], device=dev)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps(res), flush=True)
        print("DIST_RCCL_CHECK", "OK" if int(t) == 0 else f"FAILED ({int(t)})", f"world {world} steps {steps}", flush=True)
    dist.destroy_process_group()
    return 0 if int(t) == 0 else 1


if __name__ == "__main__":
    sys.exit(worker() if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1 else launch(sys.argv[1:]))
