"""Loop-level float64 definitions of the conv primitives (TEST INFRASTRUCTURE ONLY).

These restate TensorFlow's *documented definitions* directly as index arithmetic, with no
library convolution underneath, so that ``oracle/oracle.py`` (which leans on torch's CPU
conv kernels) can itself be pinned on small cases:

* ``conv2d_same``           -- tf.nn.conv2d, padding='SAME' (cross-correlation,
                               pad_before = total//2, extra pixel goes to bottom/right)
* ``conv2d_transpose_same`` -- tf.nn.conv2d_transpose defined, as TF defines it, as the
                               gradient of conv2d wrt its input (scatter form)
* ``conv3d_valid``          -- tf.nn.conv3d, padding='VALID'

Pure numpy/python loops: use on tiny tensors only.
"""
import numpy as np


def same_pads(in_size, k, stride):
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    return total // 2, total - total // 2, out


def conv2d_same(x, w, stride):
    """x: (N,Cin,H,W); w: (kh,kw,Cin,Cout) -> (N,Cout,ceil(H/s),ceil(W/s))."""
    N, Cin, H, W = x.shape
    kh, kw, _, Cout = w.shape
    pt, _, OH = same_pads(H, kh, stride)
    pl, _, OW = same_pads(W, kw, stride)
    y = np.zeros((N, Cout, OH, OW), np.float64)
    for oy in range(OH):
        for ox in range(OW):
            for ky in range(kh):
                iy = oy * stride + ky - pt
                if iy < 0 or iy >= H:
                    continue
                for kx in range(kw):
                    ix = ox * stride + kx - pl
                    if ix < 0 or ix >= W:
                        continue
                    # (N,Cin) @ (Cin,Cout)
                    y[:, :, oy, ox] += x[:, :, iy, ix].astype(np.float64) @ w[ky, kx].astype(np.float64)
    return y


def conv2d_transpose_same(x, w, stride):
    """x: (N,Cin,h,w); w: (kh,kw,Cout,Cin) -> (N,Cout,h*s,w*s).

    Gradient of the SAME forward conv F: (N,Cout,h*s,w*s) -> (N,Cin,h,w) with filter
    (kh,kw,Cout,Cin) wrt its input: every forward read  F_in[iy,ix] * w[ky,kx] -> out[oy,ox]
    becomes a scatter  y[iy,ix] += x[oy,ox] * w[ky,kx]."""
    N, Cin, h, wd = x.shape
    kh, kw, Cout, _ = w.shape
    H, W = h * stride, wd * stride
    pt, _, oh = same_pads(H, kh, stride)
    pl, _, ow = same_pads(W, kw, stride)
    assert (oh, ow) == (h, wd)
    y = np.zeros((N, Cout, H, W), np.float64)
    for oy in range(h):
        for ox in range(wd):
            for ky in range(kh):
                iy = oy * stride + ky - pt
                if iy < 0 or iy >= H:
                    continue
                for kx in range(kw):
                    ix = ox * stride + kx - pl
                    if ix < 0 or ix >= W:
                        continue
                    # (N,Cin) @ (Cin,Cout)
                    y[:, :, iy, ix] += x[:, :, oy, ox].astype(np.float64) @ w[ky, kx].astype(np.float64).T
    return y


def conv3d_valid(x, w):
    """x: (N,Cin,D,H,W); w: (kd,kh,kw,Cin,Cout) -> (N,Cout,D-kd+1,H-kh+1,W-kw+1)."""
    N, Cin, D, H, W = x.shape
    kd, kh, kw, _, Cout = w.shape
    OD, OH, OW = D - kd + 1, H - kh + 1, W - kw + 1
    y = np.zeros((N, Cout, OD, OH, OW), np.float64)
    for a in range(kd):
        for b in range(kh):
            for c in range(kw):
                patch = x[:, :, a:a + OD, b:b + OH, c:c + OW].astype(np.float64)
                y += np.einsum('nidhw,io->nodhw', patch, w[a, b, c].astype(np.float64))
    return y
