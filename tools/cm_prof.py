#!/usr/bin/env python
"""Tuning: shader-clock timeline of the strided matrix-core layers (needs a -DCM_PROF build of conv_mfma.hip via IMGCOMP_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib as L
lib = L.lib
dev = torch.device('cuda:0'); st = L.current_stream()
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 768)
g = torch.Generator(device='cpu').manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
prof = torch.zeros(1 << 22, dtype=torch.int64, device=dev)
for name, cin, cout, hin, win, tr in (('h2', 64, 128, H // 2, W // 2, 0), ('to_bn', 128, 33, H // 4, W // 4, 0), ('h12', 128, 64, H // 4, W // 4, 1)):
    w = (r(5, 5, cout, cin) if tr else r(5, 5, cin, cout)) * 0.05
    wp = torch.empty(lib.ic_conv2d_mfma_packed_floats(5, 5, cin, cout, 2, tr), device=dev)
    L.check(lib.ic_pack_conv2d_mfma_f32(L.ptr(w), L.ptr(wp), 5, 5, cin, cout, 2, tr, st))
    x = r(1, cin, hin, win)
    oh, ow = (2 * hin, 2 * win) if tr else (hin // 2, win // 2)
    y = torch.empty(1, cout, oh, ow, device=dev)
    sc, sh = r(cout).abs() + 0.5, r(cout)
    run = lambda: L.check(lib.ic_conv2d_mfma_bn_act_f32(L.ptr(x), L.ptr(wp), L.ptr(sc), L.ptr(sh), L.ptr(y), 1, cin, hin, win, cout, 5, 5, 2, tr, 1, st))
    lib.ic_conv2d_mfma_set_prof(0, 0)
    for _ in range(3): run()
    torch.cuda.synchronize()
    prof.zero_()
    a = prof.data_ptr()
    lib.ic_conv2d_mfma_set_prof(a & 0xffffffff, a >> 32)
    run(); torch.cuda.synchronize()
    lib.ic_conv2d_mfma_set_prof(0, 0)
    d = prof.cpu().view(-1, 4)
    d = d[d[:, 0] > 0].double()
    t0 = d[:, 0].min()
    print('%-6s waves %5d  prologue %6.0f  loop %7.0f  epilogue %6.0f  wave total %7.0f | launch span %7.0f  start skew mean %6.0f max %6.0f  end mean %7.0f'
          % (name, d.shape[0], (d[:, 1] - d[:, 0]).mean(), (d[:, 2] - d[:, 1]).mean(), (d[:, 3] - d[:, 2]).mean(), (d[:, 3] - d[:, 0]).mean(),
             d[:, 3].max() - t0, (d[:, 0] - t0).mean(), (d[:, 0] - t0).max(), (d[:, 3] - t0).mean()))
