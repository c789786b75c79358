#!/usr/bin/env python
"""Micro-benchmark of the 3x3 filter-gradient kernel at the training shape (HIP events)."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--n', type=int, default=32)
    p.add_argument('--h', type=int, default=32)
    p.add_argument('--w', type=int, default=32)
    p.add_argument('--reps', type=int, default=20)
    a = p.parse_args()
    lib = _lib.lib
    dev = torch.device('cuda:0'); st = _lib.current_stream(dev)
    x = torch.randn((a.n, 128, a.h, a.w), device=dev); dy = torch.randn_like(x)
    w = torch.randn((3, 3, 128, 128), device=dev); dw = torch.empty_like(w)
    need = lib.ic_conv2d_wgrad_workspace_bytes(a.n, 128, 128, a.h, a.w, 3, 3)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    ev = [ctypes.c_void_p(), ctypes.c_void_p()]
    for e in ev:
        _lib.check(lib.ic_event_create(ctypes.byref(e)))

    def run():
        _lib.check(lib.ic_conv2d_wgrad_f32(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), a.n, 128, a.h, a.w, 128, 3, 3, 1,
                                           _lib.ptr(w), 0.0, _lib.ptr(ws), need, st))
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
    flop = 2.0 * 9 * 128 * 128 * a.n * a.h * a.w
    print('wgrad N={} {}x{}: {:.1f} us per call (kernel + reduce)  {:.1f} TFLOP/s  workspace {:.1f} MB'.format(
        a.n, a.h, a.w, us, flop / us / 1e6, need / 1e6))


if __name__ == '__main__':
    main()
