#!/usr/bin/env python
"""Tuning: per-region shader-clock means of the K-split Winograd work-groups (needs a -DWN_PROF build via IMGCOMP_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib
lib = _lib.lib
n, h, w = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device('cuda:0'); st = _lib.current_stream(dev)
x = torch.randn((n, 128, h, w), device=dev); y = torch.empty_like(x); r = torch.randn_like(x)
wt = torch.randn((3, 3, 128, 128), device=dev) * 0.05
ww = torch.empty(lib.ic_wino3x3_c128_packed_floats(), device=dev)
_lib.check(lib.ic_pack_wino3x3_c128_f32(_lib.ptr(wt), _lib.ptr(ww), 0, st))
sc = torch.ones(128, device=dev); sh = torch.zeros(128, device=dev)
nwg = 2 * n * -(-h // 2) * -(-w // 32)
prof = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
lib.ic_wino3x3_c128_set_tuning(6, 1)
def run():
    _lib.check(lib.ic_wino3x3_c128_bn_act_f32(_lib.ptr(x), _lib.ptr(ww), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(r), None, _lib.ptr(y), n, h, w, 1, st))
for _ in range(300): run()
torch.cuda.synchronize()
a = prof.data_ptr()
lo = a & 0xffffffff
lib.ic_wino3x3_c128_set_tuning(0, lo if lo < 2**31 else lo - 2**32)
lib.ic_wino3x3_c128_set_tuning(1, a >> 32)
run(); torch.cuda.synchronize()
d = prof.cpu().view(nwg * 4, 4).double()
d = d[d[:, 1] > 0]
print('t16 waves %d  clocks: prologue %.0f  k-loop %.0f (%.0f per k-step x 2)  epilogue %.0f  entry %.0f | wave total %.0f'
      % (d.shape[0], d[:, 0].mean(), d[:, 1].mean(), d[:, 1].mean() / 16, d[:, 2].mean(), 0.0, d[:, :3].sum(1).mean()))
