"""A 5x5 / stride-2 SAME convolution is ONE 3x3 / stride-1 SAME convolution over the four phases of its input stacked as channels
(space-to-depth), and the 5x5 / stride-2 transposed convolution is ONE 3x3 convolution to four phase planes per output channel
(depth-to-space afterwards) -- the algebra csrc/conv5_phase.* rests on, checked against the oracle's own conv in float64, and the
float32 error of running those 3x3 convolutions in Winograd F(4x4,3x3) form (emulated as tools/wino_f4_numerics.py does).

  h2 :  Y[i] = sum_a W[a] X[2i + a - 1]     (TF SAME, even input: pad 1 before, 2 after)
        odd rows  X1[m] = X[2m+1]: taps (W[0], W[2], W[4]) at m = i-1, i, i+1
        even rows X0[m] = X[2m]  : taps (0,    W[1], W[3]) at m = i-1, i, i+1
  h12:  Y[2m+p] (adjoint of the above): p = 1: (W[4], W[2], W[0]) at i = m-1, m, m+1;  p = 0: (W[3], W[1], 0)
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from oracle import oracle as O

SEL = {1: (0, 2, 4), 0: (None, 1, 3)}            # forward: 3-tap filter of input phase p (None = zero)
SELT = {1: (4, 2, 0), 0: (3, 1, None)}           # transposed: 3-tap filter of output phase p


def conv_filter_3x3(w5):
    """[5,5,cin,cout] -> [3,3,4*cin,cout]; input channel of the stacked tensor = (2p+q) * cin + c"""
    kh, kw, cin, cout = w5.shape
    w3 = np.zeros((3, 3, 4 * cin, cout), w5.dtype)
    for p in (0, 1):
        for q in (0, 1):
            for a in range(3):
                for b in range(3):
                    if SEL[p][a] is not None and SEL[q][b] is not None:
                        w3[a, b, (2 * p + q) * cin:(2 * p + q + 1) * cin] = w5[SEL[p][a], SEL[q][b]]
    return w3


def deconv_filter_3x3(w5):
    """[5,5,cout,cin] (TF transposed layout) -> [3,3,cin,4*cout]; output channel of the 3x3 conv = co * 4 + 2p + q"""
    kh, kw, cout, cin = w5.shape
    w3 = np.zeros((3, 3, cin, 4 * cout), w5.dtype)
    for p in (0, 1):
        for q in (0, 1):
            for a in range(3):
                for b in range(3):
                    if SELT[p][a] is not None and SELT[q][b] is not None:
                        w3[a, b, :, 2 * p + q::4] = w5[SELT[p][a], SELT[q][b]].T
    return w3


def space_to_depth(x):
    N, C, H, W = x.shape
    return x.reshape(N, C, H // 2, 2, W // 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(N, 4 * C, H // 2, W // 2)


def depth_to_space(y, cout):
    N, _, H, W = y.shape
    return y.reshape(N, cout, 2, 2, H, W).permute(0, 1, 4, 2, 5, 3).reshape(N, cout, 2 * H, 2 * W)


if __name__ == '__main__':
    from tools.wino_f4_numerics import WinoConv
    rs = np.random.RandomState(0)
    x = torch.tensor(rs.normal(0, 1, (1, 64, 48, 64)))
    w = rs.normal(0, 0.03, (5, 5, 64, 128))
    ref = O.conv2d_same(x, w, 2)
    got = O.conv2d_same(space_to_depth(x), conv_filter_3x3(w), 1)
    print('h2  as one 3x3 conv over 256 phase channels: max |diff| %.2e (float64)' % float((ref - got).abs().max()))
    xt = torch.tensor(rs.normal(0, 1, (1, 128, 24, 32)))
    wt = rs.normal(0, 0.03, (5, 5, 64, 128))
    reft = O.conv2d_transpose_same(xt, wt, 2)
    gott = depth_to_space(O.conv2d_same(xt, deconv_filter_3x3(wt), 1), 64)
    print('h12 as one 3x3 conv to 256 phase channels:   max |diff| %.2e (float64)' % float((reft - gott).abs().max()))
    f4 = WinoConv(4, [0, 1, -1, 2, -2], 'lavin')
    f4c = lambda t, w3: f4(t, w3)
    w3 = conv_filter_3x3(w).astype(np.float32)
    y32 = F = None
    y_direct = O.conv2d_same(x.float(), w.astype(np.float32), 2)
    y_f4 = f4(space_to_depth(x).float().contiguous(), w3)
    sc = max(1.0, float(ref.abs().max()))
    print('h2  float32: direct 5x5 %.2e   F(4x4) over phases %.2e   (of the tensor scale, against float64)' % (
        float((y_direct.double() - ref).abs().max()) / sc, float((y_f4.double() - ref).abs().max()) / sc))
    w3t = deconv_filter_3x3(wt).astype(np.float32)
    yt_direct = O.conv2d_transpose_same(xt.float(), wt.astype(np.float32), 2)
    yt_f4 = depth_to_space(f4(xt.float().contiguous(), w3t), 64)
    sct = max(1.0, float(reft.abs().max()))
    print('h12 float32: direct 5x5 %.2e   F(4x4) over phases %.2e' % (
        float((yt_direct.double() - reft).abs().max()) / sct, float((yt_f4.double() - reft).abs().max()) / sct))
