"""in-kernel stamps of the F(4x4) kernel (needs a build with -DW4_STAMPS: tools/build_variants.sh conv3x3_wino4.hip st="-fno-slp-vectorize -DW4_STAMPS"):
prologue / loop / epilogue shader clocks per wave, and the TURNOVER of a wave slot: the clocks between the end of one work-group's wave and
the first instruction of the next wave on the same slot (same XCC, SE, CU, SIMD, wave id -- they share a clock).
   IMGCOMP_HIP_LIB=.../variants/lib_st.so [W4_FLAGS=0x8000000] python tools/w4prof.py [N H W]        (W4_FLAGS: per-call plan flags, e.g. IC_CONV3_WINO4_WG8)"""
import os, sys, ctypes
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'wino4_check.py')).read().split("torch.set_num_threads(16)")[0])
FLAGS = int(os.environ.get('W4_FLAGS', '0'), 0)
wgs = int(lib.ic_wino4_3x3_c128_workgroups(N, H, W))
raw = ctypes.CDLL(L.LIB_PATH)
buf = torch.zeros((wgs * 4, 6), dtype=torch.int64, device=dev)
raw.ic_wino4_debug_set_buffer(ctypes.c_void_p(buf.data_ptr()))
yy = torch.empty((N, 128, H, W), device=dev)


def go():
    L.check(lib.ic_wino4_3x3_c128_bn_act_f32(L.ptr(xd), L.ptr(wp4), L.ptr(scd), L.ptr(shd), L.ptr(r1d), None, L.ptr(yy), N, H, W, 1, FLAGS, L.current_stream(dev)))


for _ in range(5):
    go()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    go()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
b = buf.cpu().numpy()
print('flags %#x' % FLAGS)
print('%d waves: prologue %.0f  loop %.0f  epilogue %.0f clocks (means); loop min %d max %d; ideal MFMA issue 36864 per wave; %.1f us per launch (events)' % (
    b.shape[0], b[:, 0].mean(), b[:, 1].mean(), b[:, 2].mean(), b[:, 1].min(), b[:, 1].max(), us))
slots = defaultdict(list)
for row in b:
    hw, xcc = int(row[5]) & 0xffffffff, (int(row[5]) >> 32) & 0xf
    key = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15)          # XCC, SE, SH, CU, SIMD, wave slot
    slots[key].append((int(row[3]), int(row[4])))
gaps, chain = [], []
for k, v in slots.items():
    v.sort()
    chain.append(v[-1][1] - v[0][0])
    for (s0, e0_), (s1, e1_) in zip(v[:-1], v[1:]):
        gaps.append(s1 - e0_)
import numpy as np
gaps = np.array(gaps) if gaps else np.array([0])
print('%d wave slots used, %.2f waves per slot; span of a slot first start -> last end: mean %.0f max %.0f clocks -> %.2f GHz if the longest slot spans the launch' % (
    len(slots), b.shape[0] / max(1, len(slots)), np.mean(chain), np.max(chain), np.max(chain) / us / 1e3))
print('turnover of a slot (end of a wave -> first instruction of the next one): median %.0f  mean %.0f  p90 %.0f clocks over %d hand-overs; negative: %d' % (
    np.median(gaps), gaps.mean(), np.percentile(gaps, 90), len(gaps), int((gaps < 0).sum())))
