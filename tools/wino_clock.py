#!/usr/bin/env python
"""Shader clock the NB-segment 3x3 kernel actually runs at (a -DWN_PROF -DWN_PROF3 build: IMGCOMP_HIP_LIB=.../lib_prof3.so):
per wave, s_memtime (shader clocks) against s_memrealtime (constant 100 MHz) over the whole kernel and over the k-loop, in
the steady state of a back-to-back launch sequence, for random input, post-ReLU input and all-zero input."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib
lib = _lib.lib
dev = torch.device('cuda:0'); st = _lib.current_stream(dev)
N, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (1, 128, 192)))
w = torch.randn((3, 3, 128, 128), device=dev) * 0.05
wp = torch.empty(lib.ic_conv3x3_c128_both_packed_floats(), device=dev)
_lib.check(lib.ic_pack_conv3x3_c128_both_f32(_lib.ptr(w), _lib.ptr(wp), 0, st))
sc, sh = torch.ones(128, device=dev), torch.zeros(128, device=dev)
setp = lib.ic_wino3x3_c128_debug_set_prof_buffer; setp.argtypes = [ctypes.c_void_p]; setp.restype = None
ev = [ctypes.c_void_p(), ctypes.c_void_p()]
for e in ev: lib.ic_event_create(ctypes.byref(e))
inputs = {'random': torch.randn((N, 128, H, W), device=dev), 'relu(random)': torch.relu(torch.randn((N, 128, H, W), device=dev)),
          'zeros': torch.zeros((N, 128, H, W), device=dev)}
for fname, f in (('seg3', _lib.CONV3_WINO_SEG3), ('seg1', _lib.CONV3_WINO_SEG1)):
    for iname, x in inputs.items():
        y = torch.empty_like(x)
        buf = torch.zeros(8192 * 16, dtype=torch.int64, device=dev)
        setp(ctypes.c_void_p(buf.data_ptr()))
        go = lambda: _lib.check(lib.ic_conv3x3_c128_auto_f32(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(sc), _lib.ptr(sh), None, None, _lib.ptr(y), N, H, W, 1, f, st))
        for _ in range(300): go()
        lib.ic_event_record(ev[0], st)
        for _ in range(300): go()
        lib.ic_event_record(ev[1], st)
        torch.cuda.synchronize(); setp(None)
        ms = ctypes.c_float(); lib.ic_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
        d = buf.view(-1, 4).cpu(); d = d[d[:, 1] != 0].double()
        mhz_all = (d[:, 0].sum() / d[:, 1].sum() * 100).item(); mhz_loop = (d[:, 2].sum() / d[:, 3].sum() * 100).item()
        print('{:5s} {:13s} {:6.2f} us/launch | wave life {:6.0f} clk = {:5.2f} us | k-loop {:6.0f} clk = {:5.2f} us | shader clock {:5.0f} MHz (whole kernel), {:5.0f} MHz (k-loop)'.format(
            fname, iname, ms.value / 300 * 1e3, d[:, 0].median().item(), d[:, 1].median().item() / 100, d[:, 2].median().item(), d[:, 3].median().item() / 100, mhz_all, mhz_loop), flush=True)
