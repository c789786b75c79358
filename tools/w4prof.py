"""in-kernel stamps of the F(4x4) kernel (needs a build with -DW4_STAMPS: make -C imgcomp_cvpr_amd/csrc clean all CXXFLAGS+=-DW4_STAMPS): prologue / loop / epilogue shader clocks per wave"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'wino4_check.py')).read().split("torch.set_num_threads(16)")[0])
wgs = int(lib.ic_wino4_3x3_c128_workgroups(N, H, W))
waves = 4
raw = ctypes.CDLL(L.LIB_PATH)
buf = torch.zeros((wgs * waves, 4), dtype=torch.int64, device=dev)
raw.ic_wino4_debug_set_buffer(ctypes.c_void_p(buf.data_ptr()))
for _ in range(5):
    run4(1, (r1d,))
torch.cuda.synchronize()
b = buf.cpu().double()
print('waves', b.shape[0], ' prologue %.0f  loop %.0f  epilogue %.0f clocks (means);  loop min %.0f max %.0f; ideal MFMA issue 36864 per wave' % (
    b[:, 0].mean(), b[:, 1].mean(), b[:, 2].mean(), b[:, 1].min(), b[:, 1].max()))
bi = buf.cpu()
ok = bi[:, 3] > 0
end = (bi[:, 3] + bi[:, 0] + bi[:, 1] + bi[:, 2])[ok]
# (every XCD counts its own shader clock: spans are taken per XCD -- entries that lie within 2^32 ticks of each other -- and the longest is reported)
start = bi[:, 3][ok]
order = torch.argsort(start)
start, end = start[order], end[order]
cuts = [0] + [i + 1 for i in range(len(start) - 1) if int(start[i + 1] - start[i]) > (1 << 32)] + [len(start)]
span = max(float(end[a:b_].max() - start[a:b_].min()) for a, b_ in zip(cuts[:-1], cuts[1:]))
print('clock domains seen:', len(cuts) - 1)
print('rows without a stamp:', int((~ok).sum()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
yy = torch.empty((N, 128, H, W), device=dev)
def go():
    L.check(lib.ic_wino4_3x3_c128_bn_act_f32(L.ptr(xd), L.ptr(wp4), L.ptr(scd), L.ptr(shd), L.ptr(r1d), None, L.ptr(yy), N, H, W, 1, 0, L.current_stream(dev)))
for _ in range(5):
    go()
e0.record()
for _ in range(20):
    go()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print('launch span ~%.0f clocks; %.1f us per launch (events) -> >= %.2f GHz effective shader clock' % (span, us, span / us / 1e3))
