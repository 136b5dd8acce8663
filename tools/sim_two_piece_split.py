"""Numpy simulation (round 6): would a TWO-piece bf16 split of the Winograd-domain samples V (round-to-nearest, five MFMAs instead of six) hold the
fp32 parity gate of tests/test_gpu_forward.py (2e-5 * max|logit| + 1e-6)?  Six dilated layers + head in float64 with V rounded to two bf16 pieces
(net.py:298-311 arithmetic, random weights / inputs).  Result: 0.6e-5 .. 1.95e-5 of max|logit| -- at the gate; the product keeps the exact
three-piece split.  Usage: python tools/sim_two_piece_split.py"""
import numpy as np, sys
sys.path.insert(0, "/root/repo")
from oracle import net_numpy as onet
rng = np.random.default_rng(0)
def bf16_rne(x):
    x = np.asarray(x, np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
def split2(v):  # two-piece RNE
    p1 = bf16_rne(v); p2 = bf16_rne((v - p1).astype(np.float32))
    return (p1.astype(np.float64) + p2.astype(np.float64))
def conv_dil(x, k, b, d, vmode):
    # x (H,W,C) float64 ; direct conv in fp64, but with the Winograd V perturbation emulated: y = A^T[(G g G^T) .* V']A
    H, W, C = x.shape
    Bt = np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], np.float64)
    G = np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], np.float64)
    At = np.array([[1,1,1,0],[0,1,-1,-1]], np.float64)
    U = np.einsum('ak,klio,bl->abio', G, k.astype(np.float64), G)   # (4,4,ci,co)
    xp = np.zeros((H + 3*d + 2*d, W + 3*d + 2*d, C)); xp[d:d+H, d:d+W] = x
    y = np.zeros((H + 2*d, W + 2*d, k.shape[3]))
    # tiles: for each sub-grid residue and each tile origin
    for y0 in range(0, H, 1):
        pass
    # vectorised: gather D[a][b] planes for all output positions (tile origin = every pixel with (y//d)%2==0 and (x//d)%2==0)
    ys = np.array([yy for yy in range(H) if (yy // d) % 2 == 0]); xs = np.array([xx for xx in range(W) if (xx // d) % 2 == 0])
    D = np.zeros((4, 4, len(ys), len(xs), C))
    for a in range(4):
        for bb in range(4):
            D[a, bb] = xp[np.ix_(ys + a*d, xs + bb*d)]      # xp offset d => sample at y-d+a*d
    V = np.einsum('as,stijc,bt->abijc', Bt, D, Bt)
    if vmode == 'f32': V = V.astype(np.float32).astype(np.float64)
    if vmode == '2p': V = split2(V.astype(np.float32))
    M = np.einsum('abijc,abco->abijo', V, U)
    Y = np.einsum('ra,abijo,cb->rcijo', At, M, At)
    out = np.zeros((H + 2*d, W + 2*d, k.shape[3]))
    for r in range(2):
        for c in range(2):
            out[np.ix_(ys + r*d, xs + c*d)] = Y[r, c]
    return np.maximum(out[:H, :W] + b, 0)
worst = {}
for trial in range(6):
    w = onet.init_weights(100 + trial, 3, 0, bias_scale=0.25)
    x0 = rng.random((64, 64, 24)) * rng.choice([1.0, 3.0])
    res = {}
    for mode in ('exact', 'f32', '2p'):
        x = x0.copy()
        for L, d in enumerate([1, 2, 4, 8, 16, 1]):
            x = conv_dil(x, w[9 + 2*L], w[10 + 2*L].astype(np.float64), d, mode)
        res[mode] = x @ w[21][0, 0].astype(np.float64) + w[22]
    mx = np.abs(res['exact']).max()
    for mode in ('f32', '2p'):
        e = np.abs(res[mode] - res['exact']).max() / mx
        worst[mode] = max(worst.get(mode, 0), e)
        print(trial, mode, f"max err / max|logit| = {e:.2e}  (gate 2e-5)")
print(worst)
