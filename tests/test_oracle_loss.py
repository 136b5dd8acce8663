"""CPU: numpy loss restatement with analytic gradients vs torch autograd, KATs and golden case."""
import os

import numpy as np
import torch

from oracle import loss_numpy as oloss, net_torch as otorch
from ubdvss_amd import synthetic


def _case(seed, n_cls, n=2, h=16, w=24):
    rng = np.random.default_rng(seed)
    yt = synthetic.rectangle_maps(seed, n, h, w, n_classes=n_cls)[..., None]
    yp = rng.normal(0, 2.5, (n, h, w, 1 + n_cls))
    return yt, yp


def test_matches_torch_autograd():
    for seed, n_cls in [(0, 0), (1, 3), (2, 1)]:
        yt, yp = _case(seed, n_cls)
        t = torch.tensor(yp, dtype=torch.float64, requires_grad=True)
        loss = otorch.total_loss(torch.tensor(yt, dtype=torch.float64), t, n_cls > 0)
        loss.backward()
        l2, g2 = oloss.total_loss(yt, yp, n_cls > 0)
        assert abs(float(loss.detach()) - l2) < 1e-10
        assert np.abs(t.grad.numpy() - g2).max() < 1e-12


def test_clip_points_and_zero_grad_outside():
    assert abs(oloss.LOGIT_LO_F32 - (-16.118095)) < 1e-5 and abs(oloss.LOGIT_HI_F32 - 15.942385) < 1e-5
    yt = np.zeros((1, 2, 2, 1)); yt[0, 0, 0, 0] = 1
    yp = np.array([-40.0, 40.0, 0.3, -0.2]).reshape(1, 2, 2, 1)
    _, g = oloss.total_loss(yt, yp, False)
    assert g[0, 0, 0, 0] == 0 and g[0, 0, 1, 0] == 0 and g[0, 1, 0, 0] != 0


def test_degenerate_batches():
    yp = np.random.default_rng(0).normal(size=(1, 4, 4, 1))
    for fill in (0, 1):                                           # no positives / no negatives
        yt = np.full((1, 4, 4, 1), fill)
        loss, g = oloss.total_loss(yt, yp, False)
        assert np.isfinite(loss) and np.isfinite(g).all()


def test_topk_tie_rule_lower_index_first():
    m = oloss.topk_mask(np.array([1.0, 3.0, 3.0, 3.0, 0.5]), 2)
    assert m.tolist() == [False, True, True, False, False]


def test_golden_loss(golden_dir, manifest):
    d = np.load(os.path.join(golden_dir, "loss_case.npz"))
    yt = d["y_true"].astype(np.int64)[..., None]
    l_det, g_det = oloss.total_loss(yt, d["y_pred"][..., :1], False)
    l_all, g_all = oloss.total_loss(yt, d["y_pred"], True)
    assert abs(l_det - manifest["loss_case"]["det"]) < 1e-9 and abs(l_all - manifest["loss_case"]["total"]) < 1e-9
    assert np.abs(g_det - d["g_det"]).max() < 1e-7 and np.abs(g_all - d["g_all"]).max() < 1e-7


def test_adam_matches_torch():
    rng = np.random.default_rng(0)
    p = rng.normal(size=50); m = np.zeros(50); v = np.zeros(50)
    tp = torch.tensor(p.copy(), requires_grad=True)
    opt = torch.optim.Adam([tp], lr=1e-3, betas=(0.9, 0.999), eps=1e-7)
    for t in range(1, 4):
        g = rng.normal(size=50)
        p, m, v = otorch.adam_step(p, g, m, v, t)
        tp.grad = torch.tensor(g.copy())
        opt.step()
    # Keras folds the bias correction into lr_t and adds eps outside the corrected sqrt -- equal to
    # torch's form up to the eps placement: differences are O(eps)
    assert np.abs(p - tp.detach().numpy()).max() < 1e-6


def test_sharded_radix_select_equals_global_topk():
    """The batch-global loss mode (UBD_COMM_GLOBAL_LOSS, loss.hip) selects the hard negatives of the GLOBAL flattened batch
    (losses.py:111) from per-rank shards with: summed counters -> k; three levels of summed 11/11/10-bit histograms of the
    fp32 bit pattern -> the threshold T and the number of elements equal to T that are still selected; per-rank tie counts in
    rank order -> each rank's first tie rank.  This is that arithmetic in numpy on 2-5 shards (many forced ties), checked
    against the oracle's global top-k -- the multi-rank path cannot run on the one-GPU test box."""
    rng = np.random.default_rng(17)
    for world in (2, 3, 5):
        for trial in range(6):
            n_loc = int(rng.integers(200, 900))
            z = (rng.random((world, n_loc)) < rng.choice([0.05, 0.3, 0.6])).astype(np.float32)
            ce = np.abs(rng.normal(0, 1, (world, n_loc))).astype(np.float32)
            if trial % 2:
                ce = np.round(ce * 4) / 4                     # heavy ties
            cn = (ce * (1 - z)).astype(np.float32)            # masked-negative BCE per rank (positives -> 0)
            bits = cn.view(np.uint32)
            # stage 1: counters + level-0 histogram, "all-reduced"
            n_pos = int(z.sum()); n_tot = world * n_loc
            k = int(min(max(n_pos, 1), max(n_tot - n_pos, 1)))

            def select(hist, k_rem):                          # bin of the k_rem-th largest, elements strictly above it
                acc = 0
                for b in range(len(hist) - 1, -1, -1):
                    if acc + hist[b] >= k_rem:
                        return b, k_rem - acc
                    acc += hist[b]
                return 0, k_rem - acc
            h0 = sum(np.bincount(bits[r] >> 21, minlength=2048) for r in range(world))
            p0, k1 = select(h0, k)
            h1 = sum(np.bincount((bits[r][(bits[r] >> 21) == p0] >> 10) & 2047, minlength=2048) for r in range(world))
            b1, k2 = select(h1, k1)
            p1 = (p0 << 11) | b1
            h2 = sum(np.bincount(bits[r][(bits[r] >> 10) == p1] & 1023, minlength=1024) for r in range(world))
            b2, need_eq = select(h2, k2)
            T = np.uint32((p1 << 10) | b2)
            # stage 2: tie counts gathered in rank order
            ties = [int((bits[r] == T).sum()) for r in range(world)]
            sel = np.zeros((world, n_loc), bool)
            for r in range(world):
                base = sum(ties[:r])
                tie_rank = base + np.cumsum(bits[r] == T) - 1
                sel[r] = (bits[r] > T) | ((bits[r] == T) & (tie_rank < need_eq))
            ref = oloss.topk_mask(cn.reshape(-1).astype(np.float64), k).reshape(world, n_loc)
            assert sel.sum() == k and np.array_equal(sel, ref), (world, trial)
