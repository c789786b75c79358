import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib
lib = _lib.lib
dev = torch.device('cuda:0')
st = _lib.current_stream(dev)
torch.manual_seed(0)
w = torch.randn((3, 3, 128, 128), device=dev) * 0.05
wp = torch.empty(lib.ic_conv3x3_c128_both_packed_floats(), device=dev)
_lib.check(lib.ic_pack_conv3x3_c128_both_f32(_lib.ptr(w), _lib.ptr(wp), 0, st))
sc, sh = torch.ones(128, device=dev), torch.zeros(128, device=dev)
for (N, H, W) in [(1, 16, 64), (1, 8, 32), (1, 4, 32), (2, 12, 96)]:
    x = torch.randn((N, 128, H, W), device=dev)
    outs = {}
    for name, f in (('wholek', _lib.CONV3_WINO_WHOLEK), ('seg1', _lib.CONV3_WINO_SEG1), ('seg2', _lib.CONV3_WINO_SEG2), ('seg3', _lib.CONV3_WINO_SEG3)):
        y = torch.full((N, 128, H, W), float('nan'), device=dev)
        _lib.check(lib.ic_conv3x3_c128_auto_f32(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(sc), _lib.ptr(sh), None, None, _lib.ptr(y), N, H, W, 0, f, st))
        torch.cuda.synchronize()
        outs[name] = y
    for name in ('seg1', 'seg2', 'seg3'):
        d = (outs[name] - outs['wholek']).abs()
        d = torch.nan_to_num(d, nan=1e9)
        print((N, H, W), name, 'max err', float(d.max()))
        if float(d.max()) > 1e-4:
            bad = d > 1e-4
            # per (n, tile row, segment col): fraction bad ; per channel tile
            bt = bad.view(N, 8, 16, H // 2, 2, W // 32, 32).float().mean(dim=(2, 4, 6))   # n, ct, trow, segcol
            for n in range(N):
                for ct in range(8):
                    print('  n', n, 'ct', ct, [[round(float(v), 2) for v in row] for row in bt[n, ct]])
