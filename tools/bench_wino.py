#!/usr/bin/env python
"""Micro-benchmark: Winograd vs direct form of the 3x3 128->128 layer at one feature-map shape (HIP events)."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--n', type=int, default=1)
    p.add_argument('--h', type=int, default=128)
    p.add_argument('--w', type=int, default=192)
    p.add_argument('--reps', type=int, default=30)
    a = p.parse_args()
    lib = _lib.lib
    dev = torch.device('cuda:0')
    st = _lib.current_stream(dev)
    x = torch.randn((a.n, 128, a.h, a.w), device=dev)
    r = torch.randn_like(x)
    y = torch.empty_like(x)
    y2 = torch.empty_like(x)
    w = torch.randn((3, 3, 128, 128), device=dev) * 0.05
    wp = torch.empty(lib.ic_conv3x3_c128_packed_floats(), device=dev)
    ww = torch.empty(lib.ic_wino3x3_c128_packed_floats(), device=dev)
    _lib.check(lib.ic_pack_conv3x3_c128_f32(_lib.ptr(w), _lib.ptr(wp), st))
    _lib.check(lib.ic_pack_wino3x3_c128_f32(_lib.ptr(w), _lib.ptr(ww), 0, st))
    sc = torch.rand(128, device=dev) + 0.5
    sh = torch.randn(128, device=dev)
    ev = [ctypes.c_void_p(), ctypes.c_void_p()]
    for e in ev:
        _lib.check(lib.ic_event_create(ctypes.byref(e)))
    flop = 2.0 * 9 * 128 * 128 * a.n * a.h * a.w

    def direct():
        _lib.check(lib.ic_conv3x3_c128_bn_act_f32(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(r),
                                                  None, _lib.ptr(y), a.n, a.h, a.w, 1, st))

    def wino():
        _lib.check(lib.ic_wino3x3_c128_bn_act_f32(_lib.ptr(x), _lib.ptr(ww), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(r),
                                                  None, _lib.ptr(y2), a.n, a.h, a.w, 1, st))

    def timeit(f):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        lib.ic_event_record(ev[0], st)
        for _ in range(a.reps):
            f()
        lib.ic_event_record(ev[1], st)
        ms = ctypes.c_float()
        _lib.check(lib.ic_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms)))
        return ms.value / a.reps * 1e3
    print('shape N={} {}x{}  direct-form flop/launch {:.3e}'.format(a.n, a.h, a.w, flop))
    us = timeit(direct)
    print('direct  : {:8.2f} us  {:6.1f} TFLOP/s'.format(us, flop / us / 1e6))
    for ksplit in (0, 1):
        lib.ic_wino3x3_c128_set_tuning(2, ksplit)
        us = timeit(wino)
        print('winograd {}: {:8.2f} us  {:6.1f} direct-equivalent TFLOP/s'.format('K-split (4 quarters per group)' if ksplit else 'whole-K waves', us, flop / us / 1e6))
    lib.ic_wino3x3_c128_set_tuning(2, -1)
    print('max |direct - winograd| = {:.3e}  (max |y| {:.3e})'.format(float((y - y2).abs().max()), float(y.abs().max())))


if __name__ == '__main__':
    main()
