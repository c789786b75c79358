"""F(4x4) against the F(2x2) plan over map sizes (time per launch, one launch at a time): where ic_conv3x3_c128_pick_form should switch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib as L
lib, dev = L.lib, torch.device('cuda:0')
w = torch.randn((3, 3, 128, 128), device=dev) * 0.03
sc, sh = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev) * 0.1
wp = torch.empty(lib.ic_conv3x3_c128_both_packed_floats(), device=dev)
L.check(lib.ic_pack_conv3x3_c128_both_f32(L.ptr(w), L.ptr(wp), 0, L.current_stream(dev)))


def timed(x, r, y, flags, reps=100):
    N, _, H, W = x.shape
    def go():
        L.check(lib.ic_conv3x3_c128_auto_f32(L.ptr(x), L.ptr(wp), L.ptr(sc), L.ptr(sh), L.ptr(r), None, L.ptr(y), N, H, W, 1, flags, L.current_stream(dev)))
    for _ in range(10):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for shape in ((1, 32, 32), (1, 64, 64), (2, 64, 64), (4, 64, 64), (8, 64, 64), (1, 64, 96), (1, 96, 128), (1, 128, 128), (1, 128, 192), (2, 128, 192),
              (3, 128, 192), (1, 192, 256), (1, 256, 384), (30, 20, 20), (1, 20, 20), (32, 32, 32), (30, 40, 40), (8, 32, 32)):
    x = torch.relu(torch.randn((shape[0], 128) + shape[1:], device=dev))
    r, y = torch.randn_like(x), torch.empty_like(x)
    t4, t2 = timed(x, r, y, L.CONV3_WINO4), timed(x, r, y, L.CONV3_NO_WINO4)
    print('%2d x %3d x %3d: F4 %6.1f us (%4d work-groups)   F2 plan %6.1f us   auto picks %d' % (
        shape + (t4, lib.ic_wino4_3x3_c128_workgroups(*shape), t2, lib.ic_conv3x3_c128_pick_form(shape[0], shape[1], shape[2], 0))), flush=True)
