#!/usr/bin/env python
"""Prologue of the NB-segment 3x3 kernel split into phases (a -DWN_PROF -DWN_PROF2 build: tools/build_variants.sh for both
conv3x3_wino.hip and conv3x3_wino_tn.hip, IMGCOMP_HIP_LIB=.../lib_prof2.so): address set-up | first patch data transformed |
remaining transforms + ring writes | barrier + first operands, per wave (median, max) in shader clocks."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib
lib = _lib.lib
dev = torch.device('cuda:0'); st = _lib.current_stream(dev)
N, H, W = 1, 128, 192
w = torch.randn((3, 3, 128, 128), device=dev) * 0.05
wp = torch.empty(lib.ic_conv3x3_c128_both_packed_floats(), device=dev)
_lib.check(lib.ic_pack_conv3x3_c128_both_f32(_lib.ptr(w), _lib.ptr(wp), 0, st))
sc, sh = torch.ones(128, device=dev), torch.zeros(128, device=dev)
x = [torch.relu(torch.randn((N, 128, H, W), device=dev)) for _ in range(2)]
r = torch.randn((N, 128, H, W), device=dev)
setp = lib.ic_wino3x3_c128_debug_set_prof_buffer; setp.argtypes = [ctypes.c_void_p]; setp.restype = None
for name, f in (('seg3', _lib.CONV3_WINO_SEG3), ('seg1', _lib.CONV3_WINO_SEG1)):
    buf = torch.zeros(8192 * 16, dtype=torch.int64, device=dev)
    for i in range(10):
        _lib.check(lib.ic_conv3x3_c128_auto_f32(_lib.ptr(x[i & 1]), _lib.ptr(wp), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(r), None, _lib.ptr(x[(i + 1) & 1]), N, H, W, 1, f, st))
    torch.cuda.synchronize()
    setp(ctypes.c_void_p(buf.data_ptr()))
    _lib.check(lib.ic_conv3x3_c128_auto_f32(_lib.ptr(x[0]), _lib.ptr(wp), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(r), None, _lib.ptr(x[1]), N, H, W, 1, f, st))
    torch.cuda.synchronize(); setp(None)
    d = buf.view(-1, 4).cpu(); d = d[d.abs().sum(1) != 0].float()
    print(name, 'waves', d.shape[0], '| addr set-up med {:.0f} max {:.0f} | first data transformed med {:.0f} max {:.0f} | rest of transforms+puts med {:.0f} max {:.0f} | barrier+first read med {:.0f} max {:.0f}'.format(
        d[:, 0].median(), d[:, 0].max(), d[:, 1].median(), d[:, 1].max(), d[:, 2].median(), d[:, 2].max(), d[:, 3].median(), d[:, 3].max()), flush=True)
