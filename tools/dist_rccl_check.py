"""Real-RCCL check of the data-parallel train step's cross-stream ordering (needs >= 2 GPUs, run under torch.distributed.run):
the dilated + head segment of the gradient vector is all-reduced on the handle's communication stream under the stem backward
(comm.hip, backward.hip: ubd_comm_begin_tail behind the kernel that makes the segment final, ubd_comm_finish before Adam).  With the
bf16 step's fixed-order reductions the summed gradients must be BIT-EQUAL between
  (a) fused communication + chained partial-sum reduction (the default),
  (b) fused communication + UBD_REDUCE=batched,
  (c) no communicator in the handle: local gradients, then ONE torch.distributed all-reduce of the whole vector,
every step, for DIST_CHECK_STEPS (default 200) steps -- a missing event wait shows up as a stale or half-summed segment on some step.
Exit code 0 and "DIST_RCCL_CHECK OK" on rank 0 when all ranks agree."""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic, distributed

rank, world, local = distributed.init_from_env("nccl")
assert world >= 2, "run under torch.distributed.run with --nproc-per-node >= 2"
dev = torch.device(f"cuda:{local}")
steps = int(os.environ.get("DIST_CHECK_STEPS", "200"))
cfg = NetConfig(grey=False)
labels = synthetic.rectangle_maps(100 + rank, 8, 32, 32)                   # every rank its own shard
x = torch.from_numpy(synthetic.textured_images(200 + rank, labels, 4, 3).astype(np.float32) / 127.5 - 1.0).to(dev)
y = torch.from_numpy(labels).to(dev)


def make(mode):
    os.environ.pop("UBD_REDUCE", None)
    if mode == "batched": os.environ["UBD_REDUCE"] = "batched"           # read when the handle is created
    m = Model(cfg, dtype="bfloat16", seed=7)
    if mode != "unfused": distributed.attach_native_comm(m, fused=True)
    tr = Trainer(m, Adam(lr=1e-3))
    tr.broadcast_weights()
    return tr

trs = {mode: make(mode) for mode in ("chained", "batched", "unfused")}
os.environ.pop("UBD_REDUCE", None)
bad = 0
for s in range(steps):
    grads = {}
    for mode, tr in trs.items():
        tr.train_step_on_device(x, y)                                      # unfused: Trainer all-reduces through torch.distributed
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
t = torch.tensor([bad], device=dev)
dist.all_reduce(t)
if rank == 0: print("DIST_RCCL_CHECK", "OK" if int(t) == 0 else f"FAILED ({int(t)})", f"world {world} steps {steps}", flush=True)
dist.destroy_process_group()
sys.exit(0 if int(t) == 0 else 1)
