#!/usr/bin/env python
"""3x3 128 -> 128 filter gradient: Winograd-domain kernel against the direct form, training shape (32 x 128 x 32 x 32) by default.
   python tools/wgrad_bench.py [N H W]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib
lib = _lib.lib
dev = torch.device('cuda:0'); st = _lib.current_stream(dev)
N, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 32, 32)))
x = torch.relu(torch.randn((N, 128, H, W), device=dev)); dy = torch.randn((N, 128, H, W), device=dev) * 0.1
w = torch.randn((3, 3, 128, 128), device=dev) * 0.05
dw = torch.empty_like(w); dw2 = torch.empty_like(w)
n1 = lib.ic_conv3x3_c128_wgrad_workspace_bytes(N, H, W); n2 = lib.ic_conv2d_wgrad_workspace_bytes(N, 128, 128, H, W, 3, 3)
ws1 = torch.empty(max(n1, 16), dtype=torch.uint8, device=dev); ws2 = torch.empty(n2, dtype=torch.uint8, device=dev)
ev = [ctypes.c_void_p(), ctypes.c_void_p()]
for e in ev: lib.ic_event_create(ctypes.byref(e))
def wino(): _lib.check(lib.ic_conv3x3_c128_wgrad_f32(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), N, H, W, _lib.ptr(w), 1e-4, _lib.ptr(ws1), n1, st))
def direct(): _lib.check(lib.ic_conv2d_wgrad_f32(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw2), N, 128, H, W, 128, 3, 3, 1, _lib.ptr(w), 1e-4, _lib.ptr(ws2), n2, st))
flop = 2.0 * 9 * 128 * 128 * N * H * W
for rnd in range(3):
    for name, f in (('winograd', wino), ('direct', direct)):
        for _ in range(20): f()
        lib.ic_event_record(ev[0], st)
        for _ in range(50): f()
        lib.ic_event_record(ev[1], st)
        ms = ctypes.c_float(); lib.ic_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
        us = ms.value / 50 * 1e3
        print('{:9s} {:7.1f} us per call (kernel + slice reduction) = {:6.1f} TFLOP/s direct-form, executed {:5.1f}'.format(
            name, us, flop / us / 1e6, flop / us / 1e6 * (16.0 / 36.0 if name == 'winograd' else 1.0)), flush=True)
torch.cuda.synchronize()
print('max |winograd - direct| =', float((dw - dw2).abs().max()), 'scale', float(dw2.abs().max()))
