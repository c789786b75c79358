"""F(4x4,3x3) kernel (csrc/conv3x3_wino4.hip) against float64 and against the F(2x2) forms: error and time per launch.
   python tools/wino4_check.py [N H W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from imgcomp_cvpr_amd import _lib as L
lib = L.lib
dev = torch.device('cuda:0')
N, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1, 128, 192)
g = torch.Generator().manual_seed(1)
x = torch.relu(torch.randn((N, 128, H, W), generator=g)) * 1.5
w = torch.randn((3, 3, 128, 128), generator=g) * 0.03
sc = torch.rand(128, generator=g) * 0.6 + 0.5
sh = torch.randn(128, generator=g) * 0.1
r1 = torch.randn((N, 128, H, W), generator=g)
r2 = torch.randn((N, 128, H, W), generator=g)
xd, wd, scd, shd, r1d, r2d = (t.to(dev) for t in (x, w, sc, sh, r1, r2))
st = L.current_stream(dev)
wp4 = torch.empty(lib.ic_wino4_3x3_c128_packed_floats(), device=dev)
L.check(lib.ic_pack_wino4_3x3_c128_f32(L.ptr(wd), L.ptr(wp4), 0, st))
wp2 = torch.empty(lib.ic_wino3x3_c128_packed_floats(), device=dev)
L.check(lib.ic_pack_wino3x3_c128_f32(L.ptr(wd), L.ptr(wp2), 0, st))


def ref64(relu, res):
    y = F.conv2d(F.pad(x.double(), (1, 1, 1, 1)), w.double().permute(3, 2, 0, 1))
    y = y * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    if relu:
        y = torch.relu(y)
    for r in res:
        y = y + r.double()
    return y


def run4(relu, res, flags=0):
    y = torch.full((N, 128, H, W), float('nan'), device=dev)
    L.check(lib.ic_wino4_3x3_c128_bn_act_f32(L.ptr(xd), L.ptr(wp4), L.ptr(scd), L.ptr(shd), L.ptr(res[0]) if len(res) > 0 else None,
                                             L.ptr(res[1]) if len(res) > 1 else None, L.ptr(y), N, H, W, relu, flags, L.current_stream(dev)))
    return y


def run2(relu, res, flags=0):
    y = torch.full((N, 128, H, W), float('nan'), device=dev)
    L.check(lib.ic_wino3x3_c128_bn_act_f32(L.ptr(xd), L.ptr(wp2), L.ptr(scd), L.ptr(shd), L.ptr(res[0]) if len(res) > 0 else None,
                                           L.ptr(res[1]) if len(res) > 1 else None, L.ptr(y), N, H, W, relu, flags, L.current_stream(dev)))
    return y


torch.set_num_threads(16)
for relu, res_h, res_d in ((1, (), ()), (0, (r1,), (r1d,)), (0, (r1, r2), (r1d, r2d))):
    ref = ref64(relu, res_h)
    scale = max(1.0, float(ref.abs().max()))
    y4, y2 = run4(relu, res_d), run2(relu, res_d)
    torch.cuda.synchronize()
    e4 = float((y4.double().cpu() - ref).abs().max()) / scale
    e2 = float((y2.double().cpu() - ref).abs().max()) / scale
    print('relu {} residuals {}: F(4x4) rel err {:.2e}   F(2x2) rel err {:.2e}   nan in F(4x4): {}'.format(relu, len(res_d), e4, e2, bool(torch.isnan(y4).any())), flush=True)


def bench(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


print('shape {}x128x{}x{}: F(4x4) {:.1f} us   F(2x2) auto {:.1f} us   F(2x2) whole-K {:.1f} us   work-groups F4 {}'.format(
    N, H, W, bench(lambda: run4(1, (r1d,))), bench(lambda: run2(1, (r1d,))), bench(lambda: run2(1, (r1d,), L.CONV3_WINO_WHOLEK)),
    lib.ic_wino4_3x3_c128_workgroups(N, H, W)))
