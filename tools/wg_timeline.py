"""Per-work-group timeline of one conv3x3 launch: which CU/XCD ran each group, when, for how long."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from imgcomp_cvpr_amd import _lib
lib = _lib.lib
dev = torch.device('cuda:0'); st = _lib.current_stream(dev)
n, h, w = 1, 128, 192
x = torch.randn((n, 128, h, w), device=dev); r = torch.randn_like(x); y = torch.empty_like(x)
wt = torch.randn((3, 3, 128, 128), device=dev) * 0.05
wp = torch.empty(lib.ic_conv3x3_c128_packed_floats(), device=dev)
lib.ic_pack_conv3x3_c128_f32(_lib.ptr(wt), _lib.ptr(wp), st)
sc = torch.rand(128, device=dev) + 0.5; sh = torch.randn(128, device=dev)
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib.ic_conv3x3_c128_set_tuning(0, variant)
def run():
    _lib.check(lib.ic_conv3x3_c128_bn_act_f32(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(r), None,
                                              _lib.ptr(y), n, h, w, 0, st))
for _ in range(5): run()
nwg = 4096
dbg = torch.zeros(4 * nwg, dtype=torch.int64, device=dev)
lib.ic_conv3x3_c128_set_debug_buffer(_lib.ptr(dbg)); run(); torch.cuda.synchronize(); lib.ic_conv3x3_c128_set_debug_buffer(None)
d = dbg.cpu().numpy().reshape(nwg, 4).astype(np.uint64)
d = d[d[:, 3] != 0]
start = d[:, 0].astype(np.float64); pro = (d[:, 1] & np.uint64(0xFFFFFF)).astype(np.float64)
hw = (d[:, 1] >> np.uint64(24)); main_end = d[:, 2].astype(np.float64); end = d[:, 3].astype(np.float64)
hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.uint64); xcc = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(int)
cu = ((hwid >> np.uint64(8)) & np.uint64(0xF)).astype(int); sh_ = ((hwid >> np.uint64(12)) & np.uint64(1)).astype(int)
se = ((hwid >> np.uint64(13)) & np.uint64(7)).astype(int)
t0 = start.min()
print('groups', len(d), 'span', end.max() - t0)
print('start: min %.0f max %.0f' % (0, start.max() - t0))
dur = end - start; main = main_end - start - pro
print('dur  mean %.0f min %.0f max %.0f' % (dur.mean(), dur.min(), dur.max()))
print('main mean %.0f min %.0f max %.0f; pro mean %.0f; epi mean %.0f' % (main.mean(), main.min(), main.max(), pro.mean(), (end - main_end).mean()))
key = xcc * 1000 + se * 100 + sh_ * 16 + cu
print('distinct CUs', len(set(key)), 'groups per CU: min', min(np.bincount(np.unique(key, return_inverse=True)[1])), 'max', max(np.bincount(np.unique(key, return_inverse=True)[1])))
for xc in range(8):
    m = xcc == xc
    if m.any(): print('XCC %d: groups %d  CUs %d  dur mean %.0f max %.0f  end max %.0f' % (xc, m.sum(), len(set(key[m])), dur[m].mean(), dur[m].max(), (end[m] - t0).max()))
# per CU: number of groups and last end
ids, inv = np.unique(key, return_inverse=True)
cnt = np.bincount(inv); last = np.array([(end[inv == i] - t0).max() for i in range(len(ids))])
for c in sorted(set(cnt)):
    print('CUs with %d groups: %d, last end mean %.0f max %.0f' % (c, (cnt == c).sum(), last[cnt == c].mean(), last[cnt == c].max()))
m = xcc == 0
t00 = start[m].min()
print('XCC0 start spread', start[m].max() - t00, ' end spread', (end[m] - t00).min(), (end[m] - t00).max())
rows = []
for k in sorted(set(key[m])):
    mm = m & (key == k)
    rows.append((k, np.sort(dur[mm]).astype(int).tolist(), np.sort(start[mm] - t00).astype(int).tolist(), int((end[mm] - t00).max())))
rows.sort(key=lambda r: r[3])
for r in rows[:6] + rows[-6:]: print(r)
per_cu_mean = np.array([np.mean(r[1]) for r in rows]); per_cu_spread = np.array([max(r[1]) - min(r[1]) for r in rows])
print('per-CU mean dur: min %.0f max %.0f ; within-CU spread mean %.0f max %.0f' % (per_cu_mean.min(), per_cu_mean.max(), per_cu_spread.mean(), per_cu_spread.max()))
bidx = np.nonzero(dbg.cpu().numpy().reshape(nwg, 4)[:, 3] != 0)[0]
for k in sorted(set(key[m]))[:8]:
    mm = m & (key == k)
    o = np.argsort(dur[mm])
    print('CU', k, 'blocks (fast->slow)', bidx[mm][o].tolist(), 'dur', dur[mm][o].astype(int).tolist())
