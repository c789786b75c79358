#!/usr/bin/env python
"""Tuning: shader-clock timeline of the h13 matrix-core kernel (needs a -DH13_PROF build via IMGCOMP_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib as L
lib = L.lib
n, h, w, tpw = (int(v) for v in sys.argv[1:5])
dev = torch.device('cuda:0'); st = L.current_stream()
x = torch.randn((n, 64, h, w), device=dev); y = torch.empty((n, 3, 2 * h, 2 * w), device=dev)
wt = torch.randn((5, 5, 3, 64), device=dev) * 0.05
sc = torch.ones(3, device=dev); sh = torch.zeros(3, device=dev)
lib.ic_edge_set_tuning(0, tpw)
def run():
    L.check(lib.ic_deconv2d_bn_act_f32(L.ptr(x), L.ptr(wt), L.ptr(sc), L.ptr(sh), L.ptr(y), n, 64, h, w, 3, 5, 5, 0, None, None, st))
for _ in range(3): run()
torch.cuda.synchronize()
prof = torch.zeros(65536 * 32, dtype=torch.int64, device=dev)
a = prof.data_ptr()
lo = a & 0xffffffff
lib.ic_edge_set_tuning(1, lo if lo < 2**31 else lo - 2**32)
lib.ic_edge_set_tuning(2, a >> 32)
run(); torch.cuda.synchronize()
raw = prof.cpu().view(-1, 8)
raw = raw[raw[:, 0] > 0]
pk = raw[:, 7]
print('  prologue split: issue tile loads %.0f  filter gather+store %.0f  tile store %.0f' % ((pk & 0xfffff).double().mean(), ((pk >> 20) & 0xfffff).double().mean(), ((pk >> 40) & 0xfffff).double().mean()))
print('  kernel args ready after %.0f' % raw[:, 6].double().mean())
d = raw.double()
d[:, 6] = d[:, 5]
d[:, 7] = d[:, 6]
d = d[d[:, 0] > 0]
t0 = d[:, 0].min()
names = ['prologue', 'chunk0', 'chunk1', 'chunk2', 'chunk3', 'epilogue', 'rest']
diffs = d[:, 1:] - d[:, :-1]
print('waves %d  kernel span %.0f clk  start skew mean %.0f max %.0f' % (d.shape[0], d[:, 7].max() - t0, (d[:, 0] - t0).mean(), (d[:, 0] - t0).max()))
for i, nm in enumerate(names):
    print('  %-9s mean %8.0f  min %8.0f  max %8.0f' % (nm, diffs[:, i].mean(), diffs[:, i].min(), diffs[:, i].max()))
print('  wave total mean %.0f' % (d[:, 7] - d[:, 0]).mean())
