"""A/B of F(4x4) kernel builds (IMGCOMP_HIP_LIB=<variant>): error against float64 on one image of a batch, bit-identity of repeated
launches at full load, time per launch at three sizes (torch events around 50 launches into one output buffer)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from imgcomp_cvpr_amd import _lib as L
lib, dev = L.lib, torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
w = torch.randn((3, 3, 128, 128), generator=g) * 0.03
sc, sh = torch.rand(128, generator=g) * 0.6 + 0.5, torch.randn(128, generator=g) * 0.1
wd, scd, shd = w.to(dev), sc.to(dev), sh.to(dev)
wp4 = torch.empty(lib.ic_wino4_3x3_c128_packed_floats(), device=dev)
L.check(lib.ic_pack_wino4_3x3_c128_f32(L.ptr(wd), L.ptr(wp4), 0, L.current_stream(dev)))
wp2 = torch.empty(lib.ic_wino3x3_c128_packed_floats(), device=dev)
L.check(lib.ic_pack_wino3x3_c128_f32(L.ptr(wd), L.ptr(wp2), 0, L.current_stream(dev)))


def launch(fn, wp, x, r, y, flags=0):
    N, _, H, W = x.shape
    L.check(fn(L.ptr(x), L.ptr(wp), L.ptr(scd), L.ptr(shd), L.ptr(r), None, L.ptr(y), N, H, W, 1, flags, L.current_stream(dev)))


def timed(fn, wp, x, r, y, flags=0, reps=50):
    for _ in range(5):
        launch(fn, wp, x, r, y, flags)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch(fn, wp, x, r, y, flags)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


torch.set_num_threads(16)
x = torch.relu(torch.randn((6, 128, 128, 192), generator=g)) * 1.5
r = torch.randn((6, 128, 128, 192), generator=g)
xd, rd = x.to(dev), r.to(dev)
ys = []
for _ in range(10):
    y = torch.full_like(xd, float('nan'))
    launch(lib.ic_wino4_3x3_c128_bn_act_f32, wp4, xd, rd, y)
    ys.append(y)
torch.cuda.synchronize()
same = all(torch.equal(ys[0], t) for t in ys[1:])
ref = torch.relu(F.conv2d(F.pad(x[5:6].double(), (1, 1, 1, 1)), w.double().permute(3, 2, 0, 1)) * sc.double().view(1, -1, 1, 1)
                 + sh.double().view(1, -1, 1, 1)) + r[5:6].double()
err = float((ys[0][5:6].double().cpu() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
out = ['err %.2e' % err, 'deterministic' if same else 'NOT DETERMINISTIC']
# 0 / 1 / 2 residuals, ReLU on / off, a ragged shape: against the F(2x2) kernel (itself within 5e-7 of float64)
worst = 0.0
for shape in ((2, 37, 68), (1, 128, 192)):
    xs = torch.randn((shape[0], 128) + shape[1:], device=dev)
    r1, r2 = torch.randn_like(xs), torch.randn_like(xs)
    for relu in (0, 1):
        for res in ((None, None), (r1, None), (r1, r2)):
            y4, y2 = torch.full_like(xs, float('nan')), torch.full_like(xs, float('nan'))
            for fn, wp, y in ((lib.ic_wino4_3x3_c128_bn_act_f32, wp4, y4), (lib.ic_wino3x3_c128_bn_act_f32, wp2, y2)):
                L.check(fn(L.ptr(xs), L.ptr(wp), L.ptr(scd), L.ptr(shd), L.ptr(res[0]), L.ptr(res[1]), L.ptr(y), shape[0], shape[1], shape[2], relu, 0,
                           L.current_stream(dev)))
            torch.cuda.synchronize()
            worst = max(worst, float((y4 - y2).abs().max()) / max(1.0, float(y2.abs().max())))
out.append('vs F2 (6 cases x 2 shapes) %.2e' % worst)
for shape in ((1, 128, 192), (8, 128, 192), (1, 540, 960)):
    xs = torch.relu(torch.randn((shape[0], 128) + shape[1:], device=dev))
    rs, y = torch.randn_like(xs), torch.empty_like(xs)
    out.append('%dx%dx%d: F4 %.1f us, F2 %.1f us' % (shape + (timed(lib.ic_wino4_3x3_c128_bn_act_f32, wp4, xs, rs, y),
                                                      timed(lib.ic_wino3x3_c128_bn_act_f32, wp2, xs, rs, y))))
print(os.path.basename(os.environ.get('IMGCOMP_HIP_LIB', 'shipped')), ' | '.join(out), flush=True)
