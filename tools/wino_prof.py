#!/usr/bin/env python
"""Tuning: per-region shader-clock totals of the Winograd main loop (needs a -DWN_PROF build via IMGCOMP_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib
lib = _lib.lib
n, h, w = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device('cuda:0'); st = _lib.current_stream(dev)
x = torch.randn((n, 128, h, w), device=dev); y = torch.empty_like(x)
wt = torch.randn((3, 3, 128, 128), device=dev) * 0.05
ww = torch.empty(lib.ic_wino3x3_c128_packed_floats(), device=dev)
_lib.check(lib.ic_pack_wino3x3_c128_f32(_lib.ptr(wt), _lib.ptr(ww), 0, st))
sc = torch.ones(128, device=dev); sh = torch.zeros(128, device=dev)
nwg = n * -(-h // 4) * -(-w // 32)
prof = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
def run():
    _lib.check(lib.ic_wino3x3_c128_bn_act_f32(_lib.ptr(x), _lib.ptr(ww), _lib.ptr(sc), _lib.ptr(sh), None, None, _lib.ptr(y), n, h, w, 1, st))
for _ in range(3): run()
torch.cuda.synchronize()
a = prof.data_ptr()
lib.ic_wino3x3_c128_set_tuning(0, a & 0xffffffff if (a & 0xffffffff) < 2**31 else (a & 0xffffffff) - 2**32)
lib.ic_wino3x3_c128_set_tuning(1, (a >> 32))
run(); torch.cuda.synchronize()
d = prof.cpu().view(nwg * 4, 4).double()
live = d[:, 1] > 0
d = d[live]
span = (d[:, 3] + d[:, :3].sum(1)).max() - d[:, 3].min()
print('waves %d  clocks: prologue %.0f  loop %.0f (%.0f per k-step)  epilogue %.0f  | wave total %.0f  kernel span %.0f  start skew max %.0f'
      % (d.shape[0], d[:, 0].mean(), d[:, 1].mean(), d[:, 1].mean() / 64, d[:, 2].mean(), d[:, :3].sum(1).mean(), span, (d[:, 3] - d[:, 3].min()).max()))
