"""GPU: the N > 1 paths of the data-parallel exchange step, EXECUTED on the one-GPU box (BASELINE.json
configs[3]; no reference counterpart -- the reference is single-device).  The ranks are host threads of this process, each
with its own C-ABI handle, stream and workspace on the same MI355X; the collectives go through tests/loopback's in-process
stand-in for the six RCCL entry points comm.hip resolves (UBD_RCCL_LIB).  What is checked is everything except the wire:
  * per-replica loss, fused two-segment all-reduce inside ubd_train_step: both ranks end with the SUM of the two shards'
    gradients (bit-equal to g0 + g1 of two stand-alone runs), identical parameters after Adam with 1 / world;
  * UBD_COMM_GLOBAL_LOSS: the loss of losses.py:86-126 over the GLOBAL batch (top-k of the flattened batch of both ranks, global
    n_pos / n_neg / means) equals the oracle's loss on the concatenated batch, d loss / d logits of each shard equals the
    oracle's gradient slice, the summed parameter gradients equal a single-handle run on the whole batch;
  * explicit ubd_allreduce_grads / ubd_broadcast_params with two ranks;
  * round 4: the same three paths with EIGHT ranks (configs[3]'s world size) on tiny shapes -- fused all-reduce = the rank-order
    sum of eight shard gradients, the batch-global top-k with its k-th value repeated in several shards, broadcast from a
    non-zero root.
Round 5: the stand-in is STREAM-ORDERED (no host synchronisation of any stream: a rank's contribution is read by a device copy on
the stream it passed, the rank-order sum runs on the group's stream behind the ranks' events, the result is copied out on each
rank's stream behind the sum's event -- NCCL's visibility contract), so these tests now also check the cross-stream ORDERING of
the fused step: test_fused_step_ordering_* goes red on a build of the library with one event wait dropped
(tools/prove_comm_ordering.sh, profiles/r05_comm_ordering_power.log).
"""
import ctypes
import os
import threading

import numpy as np
import pytest
import torch

from oracle import loss_numpy as oloss
from ubdvss_amd import NetConfig, Model, Trainer, Adam, synthetic, _lib

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOOPBACK = os.path.join(ROOT, "tests", "loopback", "libloopback_nccl.so")


@pytest.fixture()
def loopback(monkeypatch):
    if not os.path.exists(LOOPBACK):
        import subprocess
        subprocess.check_call(["bash", os.path.join(ROOT, "tests", "loopback", "build.sh")])
    monkeypatch.setenv("UBD_RCCL_LIB", LOOPBACK)


def _ranks(world, body):
    """Runs body(rank) on `world` threads (ctypes releases the GIL inside the C calls, so the ranks meet in the collectives);
    re-raises the first failure."""
    errs = [None] * world
    outs = [None] * world

    def run(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                outs[r] = body(r)
                torch.cuda.synchronize()
        except BaseException as e:                      # noqa: BLE001 -- re-raised in the main thread
            errs[r] = e
    ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ts), "a rank is stuck in a collective"
    for e in errs:
        if e is not None:
            raise e
    return outs


def _attach(model, uid, rank, world, flags):
    lib = _lib.load()
    with torch.cuda.device(model.device):
        _lib.check(lib.ubd_comm_init(model._h, uid, rank, world, flags), "ubd_comm_init")
    model._native_comm = "fused" if flags & _lib.UBD_COMM_FUSED else "explicit"
    model._global_loss = bool(flags & _lib.UBD_COMM_GLOBAL_LOSS)


def _uid():
    buf = (ctypes.c_char * _lib.UBD_UNIQUE_ID_BYTES)()
    _lib.check(_lib.load().ubd_comm_unique_id(buf), "ubd_comm_unique_id")
    return buf


def _batch(n, side, seed, n_cls=0):
    labels = synthetic.rectangle_maps(seed, n, side // 4, side // 4, n_classes=n_cls)
    x = synthetic.textured_images(seed + 1, labels, 4, 3).astype(np.float32) / 127.5 - 1.0
    return torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda()


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
@pytest.mark.parametrize("fused", [True, False])
def test_two_ranks_per_replica_loss(loopback, dtype, fused):
    world, cfg = 2, NetConfig(grey=False)
    x, y = _batch(8, 96, 61)
    shards = [(x[:4], y[:4]), (x[4:], y[4:])]
    alone = []
    for r in range(world):                                       # the two shards without any communicator
        tr = Trainer(Model(cfg, dtype=dtype, seed=3), Adam(lr=1e-3))
        tr.backward_on_device(*shards[r])
        torch.cuda.synchronize()
        alone.append((tr.grads.clone(), tr.loss.clone()))
    want = alone[0][0] + alone[1][0]                             # rank order, fp32: what a sum all-reduce returns
    uid = _uid()

    def body(r):
        m = Model(cfg, dtype=dtype, seed=3 + 10 * r)             # different initial weights: the broadcast must fix that
        _attach(m, uid, r, world, _lib.UBD_COMM_FUSED if fused else 0)
        assert _lib.load().ubd_comm_world(m._h) == 2
        tr = Trainer(m, Adam(lr=1e-3))
        tr.broadcast_weights(src=0)
        tr.backward_on_device(*shards[r])
        if not fused:
            _lib.check(_lib.load().ubd_allreduce_grads(m._h, tr.grads.data_ptr(), tr.grads.numel(), m._stream()), "ubd_allreduce_grads")
        torch.cuda.current_stream().synchronize()
        g = tr.grads.clone()
        loss = tr.loss.clone()
        if fused:
            tr.apply_gradients()                                 # 1 / world inside Adam; no second collective
        torch.cuda.current_stream().synchronize()
        return g, loss, m.params.clone()
    outs = _ranks(world, body)
    assert torch.equal(outs[0][0], outs[1][0])                   # one all-reduce result on both ranks, bit for bit
    for r in range(world):
        if dtype == "bfloat16":                                  # fixed-order reductions: the shard gradients repeat bit for bit
            assert torch.equal(outs[r][0], want), (r, float((outs[r][0] - want).abs().max()))
            assert torch.equal(outs[r][1], alone[r][1])          # per-replica loss: each rank's own
        else:                                                    # fp32 weight gradients are summed with float atomics: run-to-run rounding
            assert torch.allclose(outs[r][0], want, rtol=1e-4, atol=1e-7)
            assert torch.allclose(outs[r][1], alone[r][1], rtol=1e-5)
    assert torch.equal(outs[0][2], outs[1][2])                   # same update everywhere
    if fused:
        ref = Trainer(Model(cfg, dtype=dtype, seed=3), Adam(lr=1e-3))
        ref.grads.copy_(want)
        ref.iterations = 1
        o = ref.opt
        _lib.check(_lib.load().ubd_adam_step(ref.model.params.data_ptr(), ref.grads.data_ptr(), ref.m.data_ptr(), ref.v.data_ptr(),
                                             ref.grads.numel(), 1, o.lr, o.beta_1, o.beta_2, o.epsilon, 0.5, ref.model._stream()), "adam")
        torch.cuda.synchronize()
        if dtype == "bfloat16":
            assert torch.equal(outs[0][2], ref.model.params)
        else:
            assert torch.allclose(outs[0][2], ref.model.params, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("n_cls", [0, 3])
def test_two_ranks_global_loss_equals_the_oracle_on_the_whole_batch(loopback, n_cls):
    """losses.py:99-116 at the GLOBAL batch: n_pos / n_neg, the means and the top-k of the flattened batch run over both ranks'
    pixels.  ubd_loss on two shards vs the numpy oracle on the concatenated batch; ties at the threshold are forced."""
    world = 2
    rng = np.random.default_rng(8)
    n, h, w = 6, 24, 40
    labels = synthetic.rectangle_maps(71, n, h, w, n_classes=n_cls)
    lg = rng.normal(0, 2.0, (n, h, w, 1 + n_cls)).astype(np.float32)
    lg[..., 0] = np.round(lg[..., 0] * 2) / 2                    # many equal negatives: the tie rule crosses the rank boundary
    uid = _uid()
    cfg = NetConfig(class_names=[f"c{i}" for i in range(n_cls)] if n_cls else None, grey=False)
    lib = _lib.load()

    def body(r):
        m = Model(cfg, seed=1)
        _attach(m, uid, r, world, _lib.UBD_COMM_GLOBAL_LOSS)
        lo, hi = r * n // 2, (r + 1) * n // 2
        lt = torch.from_numpy(lg[lo:hi]).cuda().contiguous()
        yt = torch.from_numpy(labels[lo:hi]).cuda().contiguous()
        loss = torch.zeros(16, device="cuda")
        grad = torch.empty_like(lt)
        ws = torch.empty(int(lib.ubd_loss_workspace_bytes(m._h, n // 2, h, w)), dtype=torch.uint8, device="cuda")
        _lib.check(lib.ubd_loss(m._h, lt.data_ptr(), yt.data_ptr(), n // 2, h, w, loss.data_ptr(), grad.data_ptr(), ws.data_ptr(),
                                ws.numel(), m._stream()), "ubd_loss")
        torch.cuda.current_stream().synchronize()
        return loss.cpu().numpy(), grad.cpu().numpy()
    outs = _ranks(world, body)
    # single handle on the whole batch: the same kernels without any collective
    from ubdvss_amd import losses
    whole_loss, whole_grad = losses.loss_and_grad(labels, lg)
    whole_loss, whole_grad = whole_loss.cpu().numpy(), whole_grad.cpu().numpy()
    got_grad = np.concatenate([outs[0][1], outs[1][1]], axis=0)
    assert np.array_equal(outs[0][0], outs[1][0])                # the global loss, identical on every rank
    assert np.allclose(outs[0][0][:8], whole_loss[:8], rtol=2e-6, atol=1e-7), (outs[0][0], whole_loss)
    assert np.array_equal(outs[0][0][7:14], whole_loss[7:14])    # counters: n_pos, tp, tn, fp, fn, class hits, n_pixels
    assert np.allclose(got_grad, whole_grad, rtol=2e-6, atol=1e-12)
    assert np.array_equal(got_grad != 0, whole_grad != 0)        # the same pixels were selected as hard negatives (tie rule incl.)
    # and the oracle itself on the concatenated batch
    ref_loss, ref_grad = oloss.total_loss(labels[..., None], lg.astype(np.float64), n_cls > 0)
    assert abs(float(outs[0][0][0]) - float(ref_loss)) <= 2e-5 * abs(float(ref_loss))
    assert np.abs(got_grad - ref_grad).max() <= 2e-5 * np.abs(ref_grad).max()
    assert np.array_equal(got_grad[..., 0] != 0, ref_grad[..., 0] != 0)


def test_two_ranks_global_loss_train_step_equals_one_handle_on_the_whole_batch(loopback):
    """ubd_train_step with UBD_COMM_FUSED | UBD_COMM_GLOBAL_LOSS on two shards: summed gradients = the gradients of ONE handle on
    the whole batch (the reference's objective at the global batch; grad_scale 1)."""
    world, cfg = 2, NetConfig(grey=False)
    x, y = _batch(8, 96, 81)
    one = Trainer(Model(cfg, seed=3), Adam(lr=1e-3))
    one.backward_on_device(x, y)
    torch.cuda.synchronize()
    uid = _uid()

    def body(r):
        m = Model(cfg, seed=3)
        _attach(m, uid, r, world, _lib.UBD_COMM_FUSED | _lib.UBD_COMM_GLOBAL_LOSS)
        tr = Trainer(m, Adam(lr=1e-3))
        tr.backward_on_device(x[4 * r:4 * r + 4], y[4 * r:4 * r + 4])
        torch.cuda.current_stream().synchronize()
        return tr.grads.clone(), tr.loss.clone()
    outs = _ranks(world, body)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][1][0]) == pytest.approx(float(one.loss[0]), rel=1e-5)
    g, ref = outs[0][0].double(), one.grads.double()
    assert float((g - ref).norm() / ref.norm()) < 2e-5


def test_two_ranks_broadcast(loopback):
    world, cfg = 2, NetConfig(grey=False)
    uid = _uid()

    def body(r):
        m = Model(cfg, seed=20 + r)
        _attach(m, uid, r, world, 0)
        before = m.params.clone()
        Trainer(m).broadcast_weights(src=1)
        torch.cuda.current_stream().synchronize()
        return before, m.params.clone()
    outs = _ranks(world, body)
    assert not torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][0]) and torch.equal(outs[1][1], outs[1][0])


# ---------------------------------------------------------------------------------------------- eight ranks (configs[3])
def test_eight_ranks_fused_allreduce_is_the_rank_order_sum(loopback):
    """world = 8 as in configs[3] (64 images per GPU there; 1 image of 64 x 64 per rank here): ubd_train_step with the fused
    two-segment all-reduce leaves, on every rank, g0 + g1 + ... + g7 added in rank order (fp32), and Adam with 1 / 8 gives the
    same parameters everywhere."""
    world, cfg = 8, NetConfig(grey=False)
    x, y = _batch(world, 64, 91)
    alone = []
    for r in range(world):
        tr = Trainer(Model(cfg, dtype="bfloat16", seed=3), Adam(lr=1e-3))
        tr.backward_on_device(x[r:r + 1], y[r:r + 1])
        torch.cuda.synchronize()
        alone.append(tr.grads.clone())
    want = alone[0].clone()
    for r in range(1, world):
        want = want + alone[r]
    uid = _uid()

    def body(r):
        m = Model(cfg, dtype="bfloat16", seed=3 + r)             # eight different initialisations: rank 0's must win
        _attach(m, uid, r, world, _lib.UBD_COMM_FUSED)
        assert _lib.load().ubd_comm_world(m._h) == 8
        tr = Trainer(m, Adam(lr=1e-3))
        tr.broadcast_weights(src=0)
        tr.backward_on_device(x[r:r + 1], y[r:r + 1])
        torch.cuda.current_stream().synchronize()
        g = tr.grads.clone()
        tr.apply_gradients()
        torch.cuda.current_stream().synchronize()
        return g, m.params.clone()
    outs = _ranks(world, body)
    for r in range(world):
        assert torch.equal(outs[r][0], want), (r, float((outs[r][0] - want).abs().max()))
        assert torch.equal(outs[r][1], outs[0][1])


def test_eight_ranks_global_topk_with_the_kth_value_in_several_shards(loopback):
    """losses.py:99-116 over eight shards: logits quantised to halves, so the k-th largest masked-negative loss value repeats in
    EVERY shard and the tf.nn.top_k tie rule (lower flat index first) has to cut through the rank order; vs the oracle on the
    concatenated batch and vs one handle on the whole batch."""
    world = 8
    rng = np.random.default_rng(18)
    n, h, w = 8, 16, 24
    labels = synthetic.rectangle_maps(73, n, h, w)
    lg = rng.normal(0, 1.5, (n, h, w, 1)).astype(np.float32)
    lg[..., 0] = np.round(lg[..., 0] * 2) / 2
    uid = _uid()
    cfg = NetConfig(grey=False)
    lib = _lib.load()

    def body(r):
        m = Model(cfg, seed=1)
        _attach(m, uid, r, world, _lib.UBD_COMM_GLOBAL_LOSS)
        lt = torch.from_numpy(lg[r:r + 1]).cuda().contiguous()
        yt = torch.from_numpy(labels[r:r + 1]).cuda().contiguous()
        loss = torch.zeros(16, device="cuda")
        grad = torch.empty_like(lt)
        ws = torch.empty(int(lib.ubd_loss_workspace_bytes(m._h, 1, h, w)), dtype=torch.uint8, device="cuda")
        _lib.check(lib.ubd_loss(m._h, lt.data_ptr(), yt.data_ptr(), 1, h, w, loss.data_ptr(), grad.data_ptr(), ws.data_ptr(),
                                ws.numel(), m._stream()), "ubd_loss")
        torch.cuda.current_stream().synchronize()
        return loss.cpu().numpy(), grad.cpu().numpy()
    outs = _ranks(world, body)
    got_grad = np.concatenate([o[1] for o in outs], axis=0)
    for r in range(1, world):
        assert np.array_equal(outs[r][0], outs[0][0])
    ref_loss, ref_grad = oloss.total_loss(labels[..., None], lg.astype(np.float64), False)
    assert abs(float(outs[0][0][0]) - float(ref_loss)) <= 2e-5 * abs(float(ref_loss))
    assert np.abs(got_grad - ref_grad).max() <= 2e-5 * np.abs(ref_grad).max()
    assert np.array_equal(got_grad[..., 0] != 0, ref_grad[..., 0] != 0)          # the same hard negatives, tie rule across ranks incl.
    # the k-th value really does repeat across shards: the hard-negative weight appears on SOME but not ALL elements that carry it
    from ubdvss_amd import losses
    whole_loss, whole_grad = losses.loss_and_grad(labels, lg)
    assert np.array_equal(got_grad != 0, whole_grad.cpu().numpy() != 0)
    assert np.allclose(outs[0][0][:8], whole_loss.cpu().numpy()[:8], rtol=2e-6, atol=1e-7)
    k = int(outs[0][0][3])
    neg = (labels == 0)
    x0 = lg[..., 0]
    ce = np.maximum(x0, 0) + np.log1p(np.exp(-np.abs(x0)))
    kth = np.sort(ce[neg])[::-1][k - 1]
    shards_with_kth = sum(int(np.any(np.isclose(ce[r][neg[r]], kth, rtol=0, atol=1e-7))) for r in range(world))
    assert shards_with_kth >= 2, shards_with_kth


def test_eight_ranks_broadcast_from_a_non_zero_root(loopback):
    world, cfg = 8, NetConfig(grey=False)
    uid = _uid()

    def body(r):
        m = Model(cfg, seed=40 + r)
        _attach(m, uid, r, world, 0)
        before = m.params.clone()
        Trainer(m).broadcast_weights(src=5)
        torch.cuda.current_stream().synchronize()
        return before, m.params.clone()
    outs = _ranks(world, body)
    assert not torch.equal(outs[0][0], outs[5][0])
    for r in range(world):
        assert torch.equal(outs[r][1], outs[5][0])



@pytest.mark.parametrize("world", [2, 4])
def test_fused_step_ordering_matches_the_explicit_all_reduce_step_by_step(loopback, monkeypatch, world):
    """Cross-stream ordering of the fused data-parallel step (comm.hip / backward.hip): the dilated + head segment is all-reduced on
    the handle's communication stream under the stem backward (event `ready`: compute -> communication stream; event `done`:
    communication stream -> the caller's stream before Adam).  Reference without any cross-stream edge: the same step with an
    unfused communicator and ONE explicit ubd_allreduce_grads on the caller's stream.  bf16 gradients are bit-reproducible, the
    stand-in sums in rank order: gradients and parameters must be BIT-equal on every one of 12 steps, with the sum deliberately
    300 us late (LOOPBACK_DELAY_US; only the big segment: the small one that follows it on the caller's stream is on time, and the
    stand-in does not order two collectives with each other) so that a consumer that does not wait reads stale bytes for certain."""
    monkeypatch.setenv("LOOPBACK_DELAY_US", "300")
    monkeypatch.setenv("LOOPBACK_DELAY_MIN_COUNT", "10000")          # the dilated + head segment (31 273 floats) is late, the stem segment (1 755) is not
    cfg = NetConfig(grey=False)
    steps = 12
    batches = [_batch(4 * world, 96, 300 + 7 * k) for k in range(3)]

    def run(fused):
        uid = _uid()

        def body(r):
            m = Model(cfg, dtype="bfloat16", seed=5)
            _attach(m, uid, r, world, _lib.UBD_COMM_FUSED if fused else 0)
            tr = Trainer(m, Adam(lr=1e-3))
            tr.broadcast_weights(src=0)
            hist = []
            for s in range(steps):
                x, y = batches[s % 3]
                tr.backward_on_device(x[4 * r:4 * r + 4], y[4 * r:4 * r + 4])
                tr.apply_gradients()                             # explicit mode: ubd_allreduce_grads on the caller's stream, then Adam
                hist.append((tr.grads.clone(), m.params.clone()))    # clones are enqueued on the rank's stream: ordered behind the step
            torch.cuda.current_stream().synchronize()
            return hist
        return _ranks(world, body)
    fused, explicit = run(True), run(False)
    for r in range(world):
        for s in range(steps):
            assert torch.equal(fused[r][s][0], explicit[r][s][0]), f"rank {r} step {s}: summed gradients differ (max {float((fused[r][s][0] - explicit[r][s][0]).abs().max()):.3e})"
            assert torch.equal(fused[r][s][1], explicit[r][s][1]), f"rank {r} step {s}: parameters differ"
            assert torch.equal(fused[r][s][1], fused[0][s][1]), f"rank {r} step {s}: replicas drifted apart"
