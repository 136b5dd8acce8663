"""One-rank check of the RCCL plumbing the multi-GPU bench relies on (run under torch.distributed.run on a GPU box):
process-group init with device_id, barrier, SUM / MAX all-reduce and broadcast on device tensors, then one bf16 train
step through Trainer with the process group attached."""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
dev = torch.device(f"cuda:{local}")
t = torch.arange(33028, dtype=torch.float32, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.SUM)
dist.broadcast(t, src=0)
m = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(m, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert float(t[-1]) == 33027.0 * dist.get_world_size() and float(m) == 1.5
from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic
cfg = NetConfig(grey=False)
model = Model(cfg, dtype="bfloat16", seed=1)
tr = Trainer(model, Adam(1e-3))
labels = synthetic.rectangle_maps(3, 4, 64, 64)
x = torch.from_numpy(synthetic.textured_images(4, labels, 4, 3).astype(np.float32) / 127.5 - 1.0).to(dev)
y = torch.from_numpy(labels.astype(np.int32)).to(dev)
l0 = float(tr.train_step_on_device(x, y).flatten()[0])
for _ in range(5): l1 = float(tr.train_step_on_device(x, y).flatten()[0])
print(f"rank {dist.get_rank()}/{dist.get_world_size()} backend {dist.get_backend()} loss {l0:.4f} -> {l1:.4f}")
dist.destroy_process_group()
