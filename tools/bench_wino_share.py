#!/usr/bin/env python
"""A/B: whole-K Winograd with per-wave input transform vs transform shared through LDS (tuning key 4)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib
lib = _lib.lib
dev = torch.device('cuda:0'); st = _lib.current_stream(dev)
n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1, 128, 192)
x = torch.randn((n, 128, h, w), device=dev); r = torch.randn_like(x)
wt = torch.randn((3, 3, 128, 128), device=dev) * 0.05
ww = torch.empty(lib.ic_wino3x3_c128_packed_floats(), device=dev)
_lib.check(lib.ic_pack_wino3x3_c128_f32(_lib.ptr(wt), _lib.ptr(ww), 0, st))
sc = torch.rand(128, device=dev) + 0.5; sh = torch.randn(128, device=dev)
lib.ic_wino3x3_c128_set_tuning(2, 0)
outs = []
for share in (0, 2, 3, 2, 3):
    lib.ic_wino3x3_c128_set_tuning(4, 1 if share else 0)
    lib.ic_wino3x3_c128_set_tuning(5, 1 if share >= 2 else 0)
    lib.ic_wino3x3_c128_set_tuning(6, 1 if share == 3 else 0)
    y = torch.full_like(x, float('nan'))
    def run():
        _lib.check(lib.ic_wino3x3_c128_bn_act_f32(_lib.ptr(x), _lib.ptr(ww), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(r), None, _lib.ptr(y), n, h, w, 1, st))
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(64): run()
    e1.record(); torch.cuda.synchronize()
    outs.append(y)
    print('form %d (0 per-wave, 2 shared + XCD runs, 3 = 16x16 jobs): %.2f us' % (share, e0.elapsed_time(e1) / 64 * 1e3))
print('max |diff| 0 vs 2 = %.3e, 2 vs 3 = %.3e  nan %d' % ((outs[0] - outs[1]).abs().max().item(), (outs[1] - outs[2]).abs().max().item(), int(torch.isnan(outs[2]).sum())))
