#!/usr/bin/env python
"""Per-wave shader-clock stamps of the Winograd kernels (a -DWN_PROF build of the library: IMGCOMP_HIP_LIB=.../lib_prof.so).
Prints, per form: prologue / k-loop / epilogue clocks per wave (median, max), the spread of wave start times, launch wall time
and the span first start -> last end in s_memtime ticks."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib

lib = _lib.lib
dev = torch.device('cuda:0')
st = _lib.current_stream(dev)
N, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (1, 128, 192)))
w = torch.randn((3, 3, 128, 128), device=dev) * 0.05
wp = torch.empty(lib.ic_conv3x3_c128_both_packed_floats(), device=dev)
_lib.check(lib.ic_pack_conv3x3_c128_both_f32(_lib.ptr(w), _lib.ptr(wp), 0, st))
sc, sh = torch.ones(128, device=dev), torch.zeros(128, device=dev)
x = torch.randn((N, 128, H, W), device=dev)
r = torch.randn((N, 128, H, W), device=dev)
y = torch.empty_like(x)
setp = lib.ic_wino3x3_c128_debug_set_prof_buffer
setp.argtypes = [ctypes.c_void_p]
setp.restype = None
ev = [ctypes.c_void_p(), ctypes.c_void_p()]
for e in ev:
    lib.ic_event_create(ctypes.byref(e))
for name, f in (('wholek', _lib.CONV3_WINO_WHOLEK), ('t16', _lib.CONV3_WINO_T16), ('seg1', _lib.CONV3_WINO_SEG1), ('seg2', _lib.CONV3_WINO_SEG2),
                ('seg3', _lib.CONV3_WINO_SEG3), ('seg3_pk', _lib.CONV3_WINO_SEG3 | _lib.CONV3_PACKED_TRANSFORM)):
    nwg = 8192
    buf = torch.zeros(nwg * 4 * 4, dtype=torch.int64, device=dev)
    go = lambda: _lib.check(lib.ic_conv3x3_c128_auto_f32(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(r), None, _lib.ptr(y), N, H, W, 1, f, st))
    setp(None)
    for _ in range(20):
        go()
    torch.cuda.synchronize()
    lib.ic_event_record(ev[0], st)
    for _ in range(20):
        go()
    lib.ic_event_record(ev[1], st)
    ms = ctypes.c_float()
    lib.ic_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
    us = ms.value / 20 * 1e3
    setp(ctypes.c_void_p(buf.data_ptr()))
    go()
    torch.cuda.synchronize()
    setp(None)
    d = buf.view(nwg * 4, 4).cpu()
    d = d[d[:, 3] != 0]
    pro, loop, epi, t0 = d[:, 0].float(), d[:, 1].float(), d[:, 2].float(), d[:, 3]
    tot = pro + loop + epi
    end = (t0 + d[:, 0] + d[:, 1] + d[:, 2])
    span = int(end.max() - t0.min())
    print('{:8s} {:7.2f} us/launch | waves {:5d} | prologue med {:6.0f} max {:6.0f} | loop med {:6.0f} max {:6.0f} | epilogue med {:6.0f} max {:6.0f} | '
          'wave total med {:6.0f} max {:6.0f} | start spread {:6d} | first start -> last end {:6d} ticks'.format(
              name, us, d.shape[0], float(pro.median()), float(pro.max()), float(loop.median()), float(loop.max()), float(epi.median()), float(epi.max()),
              float(tot.median()), float(tot.max()), int(t0.max() - t0.min()), span), flush=True)
