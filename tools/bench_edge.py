"""Time the edge layers of the autoencoder (h1, from_bn, h13) at a given image size through the C-ABI.

    python tools/bench_edge.py [--H 512 --W 768 --N 1 --C 32]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcomp_cvpr_amd import _lib as L  # noqa: E402


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--H', type=int, default=512)
    ap.add_argument('--W', type=int, default=768)
    ap.add_argument('--N', type=int, default=1)
    ap.add_argument('--C', type=int, default=32)
    ap.add_argument('--tpw', type=int, default=0, help='h13: tiles per work-group (0 = automatic)')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    if a.tpw and hasattr(L.lib, 'ic_edge_set_tuning'):
        L.lib.ic_edge_set_tuning(0, a.tpw)
    N, H, W, C = a.N, a.H, a.W, a.C
    st = L.current_stream()
    g = torch.Generator(device='cpu').manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    out = {}
    # h1
    x = torch.rand(N, 3, H, W, generator=g).mul(255).to(dev)
    w = r(5, 5, 3, 64) * 0.1
    sc, sh = r(64).abs() + 0.5, r(64)
    y = torch.empty(N, 64, H // 2, W // 2, device=dev)
    flops = 2.0 * N * 64 * (H // 2) * (W // 2) * 75
    t = timeit(lambda: L.check(L.lib.ic_conv2d_bn_act_f32(L.ptr(x), L.ptr(w), L.ptr(sc), L.ptr(sh), None, None, L.ptr(y),
                                                          N, 3, H, W, 64, 5, 5, 2, 1, None, None, st)))
    out['h1'] = (t, flops / t * 1e-6)
    # from_bn
    q = r(N, C, H // 8, W // 8)
    w = r(3, 3, 128, C) * 0.1
    sc, sh = r(128).abs() + 0.5, r(128)
    y = torch.empty(N, 128, H // 4, W // 4, device=dev)
    flops = 2.0 * N * 128 * (H // 4) * (W // 4) * C * 9 / 4
    t = timeit(lambda: L.check(L.lib.ic_deconv2d_bn_act_f32(L.ptr(q), L.ptr(w), L.ptr(sc), L.ptr(sh), L.ptr(y),
                                                            N, C, H // 8, W // 8, 128, 3, 3, 1, None, None, st)))
    out['from_bn'] = (t, flops / t * 1e-6)
    # h13
    x = r(N, 64, H // 2, W // 2)
    w = r(5, 5, 3, 64) * 0.1
    sc, sh = r(3).abs() + 0.5, r(3)
    y = torch.empty(N, 3, H, W, device=dev)
    flops = 2.0 * N * 3 * H * W * 64 * 25 / 4
    t = timeit(lambda: L.check(L.lib.ic_deconv2d_bn_act_f32(L.ptr(x), L.ptr(w), L.ptr(sc), L.ptr(sh), L.ptr(y),
                                                            N, 64, H // 2, W // 2, 3, 5, 5, 0, None, None, st)))
    out['h13'] = (t, flops / t * 1e-6)
    for k, (t, tf) in out.items():
        print('{:8s} {:8.1f} us  {:6.1f} TFLOP/s (direct-form)'.format(k, t, tf))


if __name__ == '__main__':
    main()
