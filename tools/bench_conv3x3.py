#!/usr/bin/env python
"""Micro-benchmark of the dominant kernel (ic_conv3x3_c128_bn_act_f32): every tile variant x inner-loop
schedule x work-groups-per-CU limit at one feature-map shape.  HIP events on the launch stream."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib

VARIANTS = [(4, 8, 16), (4, 4, 32), (3, 8, 12), (3, 6, 16), (3, 3, 32), (2, 4, 16), (2, 2, 32), (2, 8, 8),
            (1, 2, 16), (1, 4, 8)]


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--n', type=int, default=1)
    p.add_argument('--h', type=int, default=128)
    p.add_argument('--w', type=int, default=192)
    p.add_argument('--reps', type=int, default=30)
    p.add_argument('--variants', default='all')
    p.add_argument('--pads', default='0,86016')
    p.add_argument('--abl', default='0', help='ablation masks for the pipelined kernel (tuning)')
    p.add_argument('--phases', action='store_true', help='print in-kernel phase timing (shader clocks)')
    a = p.parse_args()
    lib = _lib.lib
    dev = torch.device('cuda:0')
    st = _lib.current_stream(dev)
    x = torch.randn((a.n, 128, a.h, a.w), device=dev)
    r = torch.randn_like(x)
    y = torch.empty_like(x)
    w = torch.randn((3, 3, 128, 128), device=dev) * 0.05
    wp = torch.empty(lib.ic_conv3x3_c128_packed_floats(), device=dev)
    _lib.check(lib.ic_pack_conv3x3_c128_f32(_lib.ptr(w), _lib.ptr(wp), st))
    sc = torch.rand(128, device=dev) + 0.5
    sh = torch.randn(128, device=dev)
    ev = [ctypes.c_void_p(), ctypes.c_void_p()]
    for e in ev:
        _lib.check(lib.ic_event_create(ctypes.byref(e)))
    flop = 2.0 * 9 * 128 * 128 * a.n * a.h * a.w

    def run():
        _lib.check(lib.ic_conv3x3_c128_bn_act_f32(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(r),
                                                  None, _lib.ptr(y), a.n, a.h, a.w, 0, st))

    vs = range(len(VARIANTS)) if a.variants == 'all' else [int(v) for v in a.variants.split(',')]
    print('shape N={} {}x{}  flop/launch {:.3e}'.format(a.n, a.h, a.w, flop))
    for v in vs:
        pt, tr, tc = VARIANTS[v]
        nwg = a.n * -(-a.h // tr) * -(-a.w // tc)
        for pipe, abl in [(pp, int(m)) for pp in (0, 1) for m in a.abl.split(',')]:
            lib.ic_conv3x3_c128_set_tuning(3, abl)
            for pad in [int(s) for s in a.pads.split(',')]:
                lib.ic_conv3x3_c128_set_tuning(0, v)
                lib.ic_conv3x3_c128_set_tuning(1, pad)
                lib.ic_conv3x3_c128_set_tuning(2, pipe)
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                lib.ic_event_record(ev[0], st)
                for _ in range(a.reps):
                    run()
                lib.ic_event_record(ev[1], st)
                ms = ctypes.c_float()
                _lib.check(lib.ic_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms)))
                us = ms.value / a.reps * 1e3
                print('variant {:2d} PT={} {}x{:<2d} nwg={:5d} pipe={} abl={} ldspad={:6d}: {:8.2f} us  {:6.1f} TFLOP/s'.format(
                    v, pt, tr, tc, nwg, pipe, abl, pad, us, flop / us / 1e6))
                if a.phases:
                    dbg = torch.zeros(4 * nwg, dtype=torch.int64, device=dev)
                    lib.ic_conv3x3_c128_set_debug_buffer(_lib.ptr(dbg))
                    run()
                    torch.cuda.synchronize()
                    lib.ic_conv3x3_c128_set_debug_buffer(None)
                    d = dbg.cpu().view(nwg, 4).double()
                    import numpy as np
                    raw = dbg.cpu().numpy().reshape(nwg, 4).astype(np.uint64)
                    pro = torch.as_tensor((raw[:, 1] & np.uint64(0xFFFFFF)).astype(np.float64))
                    main, epi = d[:, 2] - d[:, 0] - pro, d[:, 3] - d[:, 2]
                    print('    phases [shader clocks] prologue mean {:.0f} max {:.0f} | main mean {:.0f} max {:.0f} | '
                          'epilogue mean {:.0f} max {:.0f}'.format(
                              pro.mean(), pro.max(), main.mean(), main.max(), epi.mean(), epi.max()))
    lib.ic_conv3x3_c128_set_tuning(0, -1)
    lib.ic_conv3x3_c128_set_tuning(1, 0)
    lib.ic_conv3x3_c128_set_tuning(2, -1)
    lib.ic_conv3x3_c128_set_tuning(3, 0)


if __name__ == '__main__':
    main()
