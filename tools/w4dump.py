"""debug: raw accumulators of the F(4x4) kernel, two launches compared (needs a -DW4_DBG=256 build)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'wino4_check.py')).read().split("torch.set_num_threads(16)")[0])
wgs = int(lib.ic_wino4_3x3_c128_workgroups(N, H, W))
raw = ctypes.CDLL(L.LIB_PATH)
bufs = [torch.zeros((wgs * 4, 36, 64, 4), device=dev) for _ in range(4)]
ys = []
for b in bufs:
    raw.ic_wino4_debug_set_buffer(ctypes.c_void_p(b.data_ptr()))
    ys.append(run4(1, ()))
torch.cuda.synchronize()
for k in range(1, 4):
    same_y = bool(torch.equal(ys[0], ys[k]))
    d = (bufs[0] != bufs[k])
    print('run', k, 'outputs equal run 0:', same_y, ' accumulator words that differ:', int(d.sum()))
    if d.any():
        idx = d.nonzero()
        print('   (wave job, p, lane, r) of the first differing words:', idx[:24].tolist())
        import collections
        print('   by p:', collections.Counter(idx[:, 1].tolist()), ' by r:', collections.Counter(idx[:, 3].tolist()), ' lanes:', sorted(set(idx[:, 2].tolist())))
