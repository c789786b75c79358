"""h2 / h12: the direct MFMA kernels against the F(4x4)-over-phases form, time per launch (torch events, 50 launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib as L
lib, dev = L.lib, torch.device('cuda:0')
st = L.current_stream(dev)


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for N, H, W in ((1, 128, 192), (4, 128, 192), (1, 64, 64), (8, 64, 64), (1, 540, 960)):
    sc, sh = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev) * 0.1
    w = torch.randn((5, 5, 64, 128), device=dev) * 0.03
    wp4 = torch.empty(lib.ic_wino4_conv5s2_packed_floats(), device=dev)
    out = []
    for tr in (0, 1):
        n_m = lib.ic_conv2d_mfma_packed_floats(5, 5, 128 if tr else 64, 64 if tr else 128, 2, tr)
        wpm = torch.empty(n_m, device=dev)
        L.check(lib.ic_pack_conv2d_mfma_f32(L.ptr(w), L.ptr(wpm), 5, 5, 128 if tr else 64, 64 if tr else 128, 2, tr, st))
        L.check(lib.ic_pack_wino4_conv5s2_f32(L.ptr(w), L.ptr(wp4), tr, st))
        if tr:
            x = torch.relu(torch.randn((N, 128, H, W), device=dev)); y = torch.empty((N, 64, 2 * H, 2 * W), device=dev)
            t_d = timed(lambda: L.check(lib.ic_conv2d_mfma_bn_act_f32(L.ptr(x), L.ptr(wpm), L.ptr(sc), L.ptr(sh), L.ptr(y), N, 128, H, W, 64, 5, 5, 2, 1, 1, st)))
            t_w = timed(lambda: L.check(lib.ic_wino4_deconv5s2_c128_c64_bn_act_f32(L.ptr(x), L.ptr(wp4), L.ptr(sc), L.ptr(sh), L.ptr(y), N, H, W, 1, 0, st)))
        else:
            x = torch.relu(torch.randn((N, 64, 2 * H, 2 * W), device=dev)); y = torch.empty((N, 128, H, W), device=dev)
            xs = torch.empty((N, 256, H, W), device=dev)
            L.check(lib.ic_space_to_depth2_f32(L.ptr(x), L.ptr(xs), N, 64, 2 * H, 2 * W, st))
            t_d = timed(lambda: L.check(lib.ic_conv2d_mfma_bn_act_f32(L.ptr(x), L.ptr(wpm), L.ptr(sc), L.ptr(sh), L.ptr(y), N, 64, 2 * H, 2 * W, 128, 5, 5, 2, 0, 1, st)))
            t_w = timed(lambda: L.check(lib.ic_wino4_conv5s2_c64_c128_bn_act_f32(L.ptr(xs), L.ptr(wp4), L.ptr(sc), L.ptr(sh), L.ptr(y), N, H, W, 1, 0, st)))
        out.append('%s direct %.1f us, F(4x4) over phases %.1f us (%d work-groups)' % ('h12' if tr else 'h2', t_d, t_w, lib.ic_wino4_conv5s2_workgroups(N, H, W, tr)))
    print('%d x %d x %d: ' % (N, H, W) + '   '.join(out), flush=True)
