"""F(4x4) 3x3 kernel, 8-wave work-groups (IC_CONV3_WINO4_WG8: all 128 output channels of a segment in one work-group, the input
transform made once per segment) against the 4-wave form: bit-identity over shapes / residual counts / ReLU, then time per launch
alone on the stream and with n launches in flight on n streams (HIP events on the main stream around all of them)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib as L
lib, dev = L.lib, torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
w = torch.randn((3, 3, 128, 128), generator=g) * 0.03
sc, sh = torch.rand(128, generator=g) * 0.6 + 0.5, torch.randn(128, generator=g) * 0.1
wd, scd, shd = w.to(dev), sc.to(dev), sh.to(dev)
wp4 = torch.empty(lib.ic_wino4_3x3_c128_packed_floats(), device=dev)
L.check(lib.ic_pack_wino4_3x3_c128_f32(L.ptr(wd), L.ptr(wp4), 0, L.current_stream(dev)))
WG8 = L.CONV3_WINO4_WG8


def launch(x, r1, r2, y, relu, flags, st=None):
    N, _, H, W = x.shape
    L.check(lib.ic_wino4_3x3_c128_bn_act_f32(L.ptr(x), L.ptr(wp4), L.ptr(scd), L.ptr(shd), L.ptr(r1), L.ptr(r2), L.ptr(y), N, H, W, relu, flags,
                                             L.current_stream(dev) if st is None else st))


out = {'identical': True, 'cases': 0}
for shape in ((1, 128, 192), (2, 37, 68), (3, 32, 32), (1, 64, 64), (2, 100, 36), (1, 540, 960)):
    xs = torch.randn((shape[0], 128) + shape[1:], device=dev)
    r1, r2 = torch.randn_like(xs), torch.randn_like(xs)
    for relu in (0, 1):
        for res in ((None, None), (r1, None), (r1, r2)):
            ya, yb = torch.full_like(xs, float('nan')), torch.full_like(xs, float('nan'))
            launch(xs, res[0], res[1], ya, relu, 0)
            launch(xs, res[0], res[1], yb, relu, WG8)
            torch.cuda.synchronize()
            out['cases'] += 1
            if not torch.equal(ya, yb):
                out['identical'] = False
                out.setdefault('bad', []).append([list(shape), relu, sum(r is not None for r in res), float((ya - yb).abs().max())])
print(json.dumps(out), flush=True)


def timed_alone(x, r, y, flags, reps=40):
    for _ in range(5):
        launch(x, r, None, y, 1, flags)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch(x, r, None, y, 1, flags)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def timed_flight(xs, rs, ys, flags, reps=30):
    """len(xs) independent chains of launches (y -> next x: ping-pong so that every launch depends on its predecessor), one per stream"""
    main = torch.cuda.current_stream(dev)
    strs = [torch.cuda.Stream(device=dev) for _ in xs]
    import ctypes
    hs = [ctypes.c_void_p(s_.cuda_stream) for s_ in strs]

    def burst(n):
        for s_ in strs:
            s_.wait_stream(main)
        for i in range(n):
            for k in range(len(xs)):
                a, b = (xs[k], ys[k]) if i % 2 == 0 else (ys[k], xs[k])
                launch(a, rs[k], None, b, 1, flags, hs[k])
        for s_ in strs:
            main.wait_stream(s_)
    burst(4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    burst(reps)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(xs)) * 1e3


res = {}
for shape in ((1, 128, 192), (8, 128, 192), (1, 540, 960)):
    x = torch.relu(torch.randn((shape[0], 128) + shape[1:], device=dev)) * 0.5
    r, y = torch.randn_like(x) * 0.1, torch.empty_like(x)
    for rnd in range(2):
        for name, fl in (('wg4', 0), ('wg8', WG8)):
            res.setdefault('%dx%dx%d alone us' % shape, {}).setdefault(name, []).append(round(timed_alone(x, r, y, fl), 2))
for nf in (2, 4, 6, 8):
    xs = [torch.relu(torch.randn((1, 128, 128, 192), device=dev)) * 0.5 for _ in range(nf)]
    rs = [torch.randn_like(t) * 0.1 for t in xs]
    ys = [torch.empty_like(t) for t in xs]
    for rnd in range(2):
        for name, fl in (('wg4', 0), ('wg8', WG8)):
            res.setdefault('kodak map, %d in flight, us per launch' % nf, {}).setdefault(name, []).append(round(timed_flight(xs, rs, ys, fl), 2))
print(json.dumps(res, indent=1), flush=True)
