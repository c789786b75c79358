#!/usr/bin/env python
"""Experiment: one 3x3 layer (Kodak map) launched back to back on CU-masked streams."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib
from tools.bench_cumask import masked_stream

lib = _lib.lib
hip = ctypes.CDLL('libamdhip64.so')
dev = torch.device('cuda:0')
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 192)
x = torch.randn((1, 128, h, w), device=dev); y = torch.empty_like(x); r = torch.randn_like(x)
wt = torch.randn((3, 3, 128, 128), device=dev) * 0.05
ww = torch.empty(lib.ic_wino3x3_c128_packed_floats(), device=dev)
_lib.check(lib.ic_pack_wino3x3_c128_f32(_lib.ptr(wt), _lib.ptr(ww), 0, _lib.current_stream(dev)))
sc = torch.ones(128, device=dev); sh = torch.zeros(128, device=dev)
torch.cuda.synchronize()

def run_on(stream, n=64):
    with torch.cuda.stream(stream):
        st = _lib.current_stream(dev)
        for _ in range(8):
            _lib.check(lib.ic_wino3x3_c128_bn_act_f32(_lib.ptr(x), _lib.ptr(ww), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(r), None, _lib.ptr(y), 1, h, w, 1, st))
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            _lib.check(lib.ic_wino3x3_c128_bn_act_f32(_lib.ptr(x), _lib.ptr(ww), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(r), None, _lib.ptr(y), 1, h, w, 1, st))
        stream.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

print('default stream      %.1f us' % run_on(torch.cuda.current_stream(dev)))
print('torch side stream   %.1f us' % run_on(torch.cuda.Stream(device=dev)))
for n in (256, 224, 200, 192, 160, 128):
    print('masked %3d CUs      %.1f us' % (n, run_on(masked_stream(hip, list(range(n))))))
