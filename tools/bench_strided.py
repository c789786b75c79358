#!/usr/bin/env python
"""Time the strided 5x5 matrix-core layers (h2, to_bn, h12) at a given image size through the C-ABI."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcomp_cvpr_amd import _lib as L


def timeit(fn, iters=40):
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
    ap.add_argument('--C', type=int, default=33)
    a = ap.parse_args()
    dev = torch.device('cuda:0'); st = L.current_stream()
    g = torch.Generator(device='cpu').manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    N, H, W = a.N, a.H, a.W
    for name, cin, cout, hin, win, tr in (('h2', 64, 128, H // 2, W // 2, 0), ('to_bn', 128, a.C, H // 4, W // 4, 0),
                                          ('h12', 128, 64, H // 4, W // 4, 1), ('3x3s2 (bwd of from_bn)', 128, 32, H // 4, W // 4, 0)):
        k = 3 if name.startswith('3x3') else 5
        w = (r(k, k, cout, cin) if tr else r(k, k, cin, cout)) * 0.05
        n_pk = L.lib.ic_conv2d_mfma_packed_floats(k, k, cin, cout, 2, tr)
        wp = torch.empty(n_pk, device=dev)
        L.check(L.lib.ic_pack_conv2d_mfma_f32(L.ptr(w), L.ptr(wp), k, k, cin, cout, 2, tr, st))
        x = r(N, cin, hin, win)
        oh, ow = (2 * hin, 2 * win) if tr else (hin // 2, win // 2)
        y = torch.empty(N, cout, oh, ow, device=dev)
        sc, sh = r(cout).abs() + 0.5, r(cout)
        t = timeit(lambda: L.check(L.lib.ic_conv2d_mfma_bn_act_f32(L.ptr(x), L.ptr(wp), L.ptr(sc), L.ptr(sh), L.ptr(y), N, cin, hin, win,
                                                                    cout, k, k, 2, tr, 1, st)))
        flops = 2.0 * N * cout * cin * k * k * (hin * win if tr else oh * ow)
        print('%-24s %8.1f us  %6.1f TFLOP/s' % (name, t, flops / t * 1e-6))


if __name__ == '__main__':
    main()
