"""GPU: the data-parallel exchange step inside the C ABI (include/ubd.h, ubd_comm_*; no reference counterpart --
the reference is single-device).  The GPU box has one MI355X, so the communicator has one rank: what is checked here is that
the RCCL calls really run on the handle's communicator / streams (fused into the train step under the stem backward, or as
an explicit ubd_allreduce_grads) and leave gradients, loss and the Adam update bit-identical to the run without one.  The
N-rank arithmetic (sum, 1/world scale, broadcast) is covered on CPU by tests/test_dist_gloo.py and by bench.py --gpus N."""
import ctypes

import numpy as np
import pytest
import torch

from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic, distributed, _lib

pytestmark = pytest.mark.gpu


def _batch(n=4, side=64):
    labels = synthetic.rectangle_maps(51, n, side // 4, side // 4)
    x = synthetic.textured_images(52, labels, 4, 3).astype(np.float32) / 127.5 - 1.0
    return torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda()


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
@pytest.mark.parametrize("fused", [True, False])
def test_one_rank_communicator_leaves_the_step_unchanged(dtype, fused):
    cfg = NetConfig(grey=False)
    x, y = _batch()
    ref = Trainer(Model(cfg, dtype=dtype, seed=3), Adam(lr=1e-3))
    for _ in range(3):
        ref.train_step_on_device(x, y)
    m = Model(cfg, dtype=dtype, seed=3)
    assert distributed.attach_native_comm(m, fused=fused) == 1
    assert _lib.load().ubd_comm_world(m._h) == 1
    tr = Trainer(m, Adam(lr=1e-3))
    tr.broadcast_weights()                                    # ubd_broadcast_params on the handle's communicator
    for _ in range(3):
        tr.train_step_on_device(x, y)
    torch.cuda.synchronize()
    if dtype == "bfloat16":                                   # fixed-order reductions: bit-identical
        assert torch.equal(tr.grads, ref.grads) and torch.equal(m.params, ref.model.params)
    else:
        assert torch.allclose(tr.grads, ref.grads, rtol=1e-4, atol=1e-7) and torch.allclose(m.params, ref.model.params, rtol=1e-5, atol=1e-7)
    assert float(tr.loss[0]) == pytest.approx(float(ref.loss[0]), rel=1e-5)
    # explicit collective on a raw buffer: in place, sum over one rank = identity
    buf = torch.arange(33028, dtype=torch.float32, device="cuda")
    _lib.check(_lib.load().ubd_allreduce_grads(m._h, buf.data_ptr(), buf.numel(), m._stream()), "ubd_allreduce_grads")
    torch.cuda.synchronize()
    assert torch.equal(buf, torch.arange(33028, dtype=torch.float32, device="cuda"))
    _lib.check(_lib.load().ubd_comm_destroy(m._h), "ubd_comm_destroy")
    assert _lib.load().ubd_comm_world(m._h) == 1


def test_collectives_without_a_communicator_fail_loudly():
    m = Model(NetConfig(grey=False), seed=0)
    lib = _lib.load()
    buf = torch.zeros(8, device="cuda")
    assert lib.ubd_allreduce_grads(m._h, buf.data_ptr(), 8, m._stream()) != 0
    assert b"no communicator" in lib.ubd_last_error()
    assert lib.ubd_broadcast_params(m._h, buf.data_ptr(), 8, 0, m._stream()) != 0
    bad = (ctypes.c_char * 128)()
    assert lib.ubd_comm_init(m._h, bad, 3, 2, 0) != 0 and b"out of range" in lib.ubd_last_error()


@pytest.mark.parametrize("n_cls", [0, 3])
def test_batch_global_loss_mode_one_rank_equals_local(n_cls):
    """UBD_COMM_GLOBAL_LOSS with one rank: the eight collectives of the sharded loss run (sums, counters, three histograms, tie
    counts, final sums) and must leave loss, metrics, gradients and the update identical to the per-replica path -- with one
    replica the two semantics coincide.  (The sharded selection arithmetic itself is checked against the oracle's global
    top-k on CPU: tests/test_oracle_loss.py::test_sharded_radix_select_equals_global_topk.)"""
    cfg = NetConfig(class_names=[f"c{i}" for i in range(n_cls)] if n_cls else None, grey=False)
    labels = synthetic.rectangle_maps(61, 4, 16, 16, n_classes=n_cls)
    x = torch.from_numpy(synthetic.textured_images(62, labels, 4, 3).astype(np.float32) / 127.5 - 1.0).cuda()
    y = torch.from_numpy(labels).cuda()
    ref = Trainer(Model(cfg, seed=3), Adam(lr=1e-3))
    ref.train_step_on_device(x, y)
    m = Model(cfg, seed=3)
    distributed.attach_native_comm(m, fused=True, global_loss=True)
    tr = Trainer(m, Adam(lr=1e-3))
    tr.train_step_on_device(x, y)
    torch.cuda.synchronize()
    assert torch.equal(tr.loss, ref.loss)
    assert torch.allclose(tr.grads, ref.grads, rtol=1e-4, atol=1e-7) and torch.allclose(m.params, ref.model.params, rtol=1e-5, atol=1e-7)



@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: a real two-rank RCCL exchange (the GPU box of the test run has one)")
def test_two_rank_real_rccl_fused_step_is_bit_equal_to_the_unfused_all_reduce():
    """the fused data-parallel step puts the dilated + head segment's all-reduce on the communication stream under the stem
    backward; chained reduction moved the point where that segment becomes final.  On real RCCL with two ranks (one process per GPU,
    tools/dist_rccl_check.py): gradients of the fused + chained, fused + batched and unfused (one torch all-reduce) steps are BIT-equal
    on every one of 200 steps, and the ranks' parameters agree afterwards.  Skipped on a one-GPU box (there the N-rank paths run as
    threads over the stream-ordered in-process stand-in: tests/test_gpu_comm_loopback.py)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29577", os.path.join(root, "tools", "dist_rccl_check.py")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "DIST_RCCL_CHECK OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
