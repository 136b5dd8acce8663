"""CPU, world_size 2 over gloo: the data-parallel plumbing (gradient sum all-reduce + 1/world scale,
parameter broadcast, image sharding, max-over-ranks timing) -- the same functions the GPU trainer and
bench.py call with backend nccl (= RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import net_numpy as onet, net_torch as otorch
from ubdvss_amd import distributed as ud, synthetic


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rk, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rk), WORLD_SIZE=str(world), LOCAL_RANK=str(rk))
    r, w, _ = ud.init_from_env(backend="gloo")
    assert (r, w) == (rk, world) and ud.world_size() == world and ud.rank() == rk
    # per-replica loss/gradients on this rank's shard (oracle), exactly what each GPU rank computes
    torch.set_num_threads(2)
    n_total = 4
    lo, hi = ud.shard_range(n_total, rk, world)
    wts = onet.init_weights(1, 3, 0, bias_scale=0.1)
    x = synthetic.noise_images(5, n_total, 32, 32, 3)[lo:hi]
    yt = synthetic.rectangle_maps(6, n_total, 8, 8)[lo:hi, ..., None]
    _, _, _, grads = otorch.loss_and_grads(x, yt, wts, False)
    flat = torch.from_numpy(onet.flatten_weights(grads).astype(np.float32))
    scale = ud.allreduce_gradients(flat)
    params = torch.from_numpy(onet.flatten_weights(wts).astype(np.float32)) + (rk * 1.0)
    ud.broadcast_parameters(params, src=0)
    tmax = ud.max_over_ranks(1.0 + rk)
    np.savez(os.path.join(out_dir, f"r{rk}.npz"), g=flat.numpy() * scale, p=params.numpy(), tmax=tmax, lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_two_ranks(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 2, 2, 4)
    assert np.array_equal(r0["g"], r1["g"])                       # every rank holds the same mean gradient
    assert np.array_equal(r0["p"], r1["p"])                       # broadcast from rank 0
    assert float(r0["tmax"]) == float(r1["tmax"]) == 2.0
    # mean of the per-shard gradients, computed here without any collective
    wts = onet.init_weights(1, 3, 0, bias_scale=0.1)
    acc = 0
    for lo, hi in ((0, 2), (2, 4)):
        x = synthetic.noise_images(5, 4, 32, 32, 3)[lo:hi]
        yt = synthetic.rectangle_maps(6, 4, 8, 8)[lo:hi, ..., None]
        acc = acc + onet.flatten_weights(otorch.loss_and_grads(x, yt, wts, False)[3])
    assert np.abs(r0["g"] - acc / 2).max() < 1e-6 * max(1.0, np.abs(acc).max())


def test_shard_range_covers_everything():
    for n in (1, 7, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [ud.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_readiness_command_picks_its_world_sizes():
    """tools/dist_rccl_check.py (the one command for an 8-GPU node): world sizes 2, 4, 8 as far as the GPUs go, a clear refusal otherwise"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("dist_rccl_check", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "dist_rccl_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.worlds_for(8) == [2, 4, 8] and mod.worlds_for(4) == [2, 4] and mod.worlds_for(3) == [2] and mod.worlds_for(8, 8) == [8]
    for bad in ((1, None), (8, 1), (2, 4)):
        with pytest.raises(SystemExit):
            mod.worlds_for(*bad)
