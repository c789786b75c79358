#!/usr/bin/env python
"""One 3x3 layer, one form, back to back on the residual-stack shape -- the target of the rocprofv3 counter passes.
   python tools/run_layer.py [--form seg3] [--n 1 --h 128 --w 192] [--launches 20] [--calib]
--calib adds a device-to-device copy of a known size (64 Mi floats read + written) so that FETCH_SIZE / WRITE_SIZE of this
access pattern can be calibrated in the same pass."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib

FORMS = {'auto': 0, 'wholek': _lib.CONV3_WINO_WHOLEK, 'ksplit': _lib.CONV3_WINO_KSPLIT, 't16': _lib.CONV3_WINO_T16, 'seg1': _lib.CONV3_WINO_SEG1,
         'seg2': _lib.CONV3_WINO_SEG2, 'seg3': _lib.CONV3_WINO_SEG3, 'direct': _lib.CONV3_DIRECT,
         'w4': _lib.CONV3_WINO4, 'w4wg8': _lib.CONV3_WINO4 | _lib.CONV3_WINO4_WG8}
p = argparse.ArgumentParser()
p.add_argument('--form', default='auto')
p.add_argument('--n', type=int, default=1)
p.add_argument('--h', type=int, default=128)
p.add_argument('--w', type=int, default=192)
p.add_argument('--launches', type=int, default=20)
p.add_argument('--calib', action='store_true')
a = p.parse_args()
lib = _lib.lib
dev = torch.device('cuda:0')
st = _lib.current_stream(dev)
ws = []
for i in range(8):
    w = torch.randn((3, 3, 128, 128), device=dev) * 0.05
    wp = torch.empty(lib.ic_conv3x3_c128_both_packed_floats(), device=dev)
    _lib.check(lib.ic_pack_conv3x3_c128_both_f32(_lib.ptr(w), _lib.ptr(wp), 0, st))
    ws.append(wp)
x = [torch.relu(torch.randn((a.n, 128, a.h, a.w), device=dev)) for _ in range(2)]
r = torch.randn((a.n, 128, a.h, a.w), device=dev)
sc, sh = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev) * 0.1
for i in range(a.launches):
    _lib.check(lib.ic_conv3x3_c128_auto_f32(_lib.ptr(x[i & 1]), _lib.ptr(ws[i % 8]), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(r), None,
                                            _lib.ptr(x[(i + 1) & 1]), a.n, a.h, a.w, 1, FORMS[a.form], st))
torch.cuda.synchronize()
if a.calib:
    src = torch.randn(64 * 1024 * 1024, device=dev)
    dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)
    torch.cuda.synchronize()
