"""Numerics of Winograd F(4x4,3x3) against F(2x2,3x3) for the 64 residual 3x3 layers, BEFORE any kernel exists (review r3 item 7):
the whole encoder / decoder evaluated on the CPU in float32 with the 3x3 128 -> 128 convs replaced by an emulated Winograd form
(transforms and the channel contraction in float32, filter transform G g G^T in float64 rounded once, as a pack kernel would),
compared with the float64 oracle -- the same comparison the parity tests make for the device path (tests/util.py: error
relative to the tensor scale; bound 5e-5 for a whole network, 1e-4 end to end).

  python tools/wino_f4_numerics.py [H W]       (default 256 256; 512 768 = a Kodak image, ~2 min on 8 cores)

Transform matrices by the Toom-Cook construction for arbitrary points (checked against direct correlation at start-up):
  y = A^T [(G g) . (B^T d)],  A^T = V_m^T, G = D V_r, B^T = D^-1 V^-T, V = Vandermonde of the points (+ infinity), D a row scaling.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
import torch.nn.functional as F

from imgcomp_cvpr_amd import config_parser as cp, weights as W
from oracle import oracle as O


def toom_cook(m, r, points, scale='lavin'):
    """matrices (AT m x n, G n x r, BT n x n), n = m + r - 1, finite `points` (n - 1 of them) + infinity"""
    n = m + r - 1
    assert len(points) == n - 1
    a = np.asarray(points, np.float64)
    V = np.zeros((n, n))
    for j in range(n - 1):
        V[j] = a[j] ** np.arange(n)
    V[n - 1, n - 1] = 1.0
    AT = V[:, :m].T.copy()
    AT[:, n - 1] = 0.0
    AT[m - 1, n - 1] = 1.0
    G = V[:, :r].copy()
    G[n - 1] = 0.0
    G[n - 1, r - 1] = 1.0
    BT = np.linalg.inv(V).T
    if scale == 'lavin':                        # G rows divided by N_j = prod_{l != j} (a_j - a_l): B^T gets small integer-like entries
        D = np.ones(n)
        for j in range(n - 1):
            D[j] = 1.0 / np.prod([a[j] - a[l] for l in range(n - 1) if l != j])
        G = D[:, None] * G
        BT = BT / D[:, None]
    return AT, G, BT


def check(m, r, AT, G, BT):
    rs = np.random.RandomState(0)
    d, g = rs.randn(m + r - 1), rs.randn(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    assert np.allclose(y, ref, atol=1e-10), (y, ref)


class WinoConv(object):
    def __init__(self, m, points, scale):
        self.m = m
        AT, G, BT = toom_cook(m, 3, points, scale)
        check(m, 3, AT, G, BT)
        self.AT64, self.G64, self.BT64 = AT, G, BT
        self.AT, self.BT = torch.tensor(AT, dtype=torch.float32), torch.tensor(BT, dtype=torch.float32)
        self.cache = {}

    def __call__(self, x, w_tf):
        """x (N, 128, H, W) float32, w_tf [3,3,cin,cout]: SAME stride-1 conv through F(m x m, 3 x 3) in float32"""
        m, n = self.m, self.m + 2
        key = id(w_tf)
        if key not in self.cache:
            g = np.asarray(w_tf, np.float64).transpose(3, 2, 0, 1)                      # co, ci, 3, 3
            U = np.einsum('ai,ocij,bj->ocab', self.G64, g, self.G64)                    # float64, rounded once
            self.cache[key] = torch.tensor(U, dtype=torch.float32)
        U = self.cache[key]
        N, C, H, Wd = x.shape
        th, tw = -(-H // m), -(-Wd // m)
        xp = F.pad(x, (1, tw * m - Wd + 1, 1, th * m - H + 1))
        t = xp.unfold(2, n, m).unfold(3, n, m)                                          # N, C, th, tw, n, n
        V = torch.einsum('ai,nctuij,bj->nctuab', self.BT, t, self.BT)
        M = torch.einsum('ocab,nctuab->notuab', U, V)
        Y = torch.einsum('ia,notuab,jb->notuij', self.AT, M, self.AT)                   # N, co, th, tw, m, m
        Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, U.shape[0], th * m, tw * m)
        return Y[:, :, :H, :Wd].contiguous()


def run(x_np, wts, cfg, conv):
    orig = O.conv2d_same

    def patched(x, w_tf, stride):
        if conv is not None and x.dtype == torch.float32 and stride == 1 and tuple(np.asarray(w_tf).shape) == (3, 3, 128, 128):
            return conv(x, w_tf)
        return orig(x, w_tf, stride)
    O.conv2d_same = patched
    try:
        with torch.no_grad():
            x = torch.as_tensor(x_np).float()
            enc = O.encode(x, wts, cfg)
            xo = O.decode(enc.qhard, wts, cfg)
    finally:
        O.conv2d_same = orig
    return enc, xo


def rel(a, b):
    b = b.double()
    return float((a.double() - b).abs().max()) / max(1.0, float(b.abs().max()))


def main():
    H, Wd = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 256)
    torch.set_num_threads(os.cpu_count() or 8)
    ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
    pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
    wts = W.synthetic_weights(ae_cfg, pc_cfg)
    cfg = ae_cfg.as_dict()
    x = W.synthetic_image((1, 3, H, Wd), 'natural', seed=0)
    with torch.no_grad():
        ref = O.encode(torch.as_tensor(x).double(), wts, cfg)
        ref_xo = O.decode(ref.qhard, wts, cfg)
    forms = [('direct fp32', None),
             ('F(2x2) {0,1,-1}', WinoConv(2, [0, 1, -1], 'lavin')),
             ('F(4x4) {0,1,-1,2,-2}', WinoConv(4, [0, 1, -1, 2, -2], 'lavin')),
             ('F(4x4) {0,1,-1,1/2,-1/2}', WinoConv(4, [0, 1, -1, 0.5, -0.5], 'lavin')),
             ('F(4x4) {0,1,-1,1/2,-2}', WinoConv(4, [0, 1, -1, 0.5, -2], 'lavin')),
             ('F(4x4) {0,1/2,-1/2,3/2,-3/2}', WinoConv(4, [0, 0.5, -0.5, 1.5, -1.5], 'lavin'))]
    print('image {} x {}; errors relative to the tensor scale max(1, max|ref|) against the float64 oracle (bounds: network 5e-5, end to end 1e-4)'.format(H, Wd))
    for name, conv in forms:
        enc, xo = run(x, wts, cfg, conv)
        # decode from the REFERENCE symbols so that encoder flips do not hide the decoder's own error
        xo_same = run.__globals__['O'].decode  # noqa
        flips = int((enc.symbols != ref.symbols).sum())
        print('{:32s} z {:.2e}   x_out (own symbols) {:.2e}   symbol flips {} of {}'.format(
            name, rel(enc.z, ref.z), rel(xo, ref_xo), flips, ref.symbols.numel()), flush=True)
        if conv is not None:
            # decoder alone on the reference's qhard
            orig = O.conv2d_same
            O.conv2d_same = lambda x_, w_, s_, c=conv, o=orig: (c(x_, w_) if (x_.dtype == torch.float32 and s_ == 1 and tuple(np.asarray(w_).shape) == (3, 3, 128, 128)) else o(x_, w_, s_))
            with torch.no_grad():
                xo2 = O.decode(ref.qhard.float(), wts, cfg)
            O.conv2d_same = orig
            print('{:32s} decoder alone on the reference symbols: x_out {:.2e} (abs {:.2e} grey levels)'.format(
                '', rel(xo2, ref_xo), float((xo2.double() - ref_xo).abs().max())), flush=True)


if __name__ == '__main__':
    main()
