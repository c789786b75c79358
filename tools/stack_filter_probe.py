#!/usr/bin/env python
"""Does the 3x3 stack pay for fetching each layer's filter fragments from HBM?  ic_ae_res_stack_f32 over a Kodak-sized map with
(a) 32 distinct packed filters (2 MB each: the real case) and (b) the same filter for every layer (always L2-resident)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib
lib = _lib.lib
dev = torch.device('cuda:0'); st = _lib.current_stream(dev)
N, H, W, B = 1, 128, 192, 5
nl = 6 * B + 2
pk = lib.ic_conv3x3_c128_both_packed_floats()
ws_ = [torch.empty(pk, device=dev) for _ in range(nl)]
for wp in ws_:
    w = torch.randn((3, 3, 128, 128), device=dev) * 0.03
    _lib.check(lib.ic_pack_conv3x3_c128_both_f32(_lib.ptr(w), _lib.ptr(wp), 0, st))
sc, sh = torch.ones(128, device=dev), torch.zeros(128, device=dev)
x = torch.relu(torch.randn((N, 128, H, W), device=dev)); y = torch.empty_like(x)
need = lib.ic_ae_res_stack_workspace_bytes(N, H, W)
ws = torch.empty(need, dtype=torch.uint8, device=dev)
ev = [ctypes.c_void_p(), ctypes.c_void_p()]
for e in ev: lib.ic_event_create(ctypes.byref(e))
def run(tab, reps=20):
    for _ in range(5): _lib.check(lib.ic_ae_res_stack_f32(_lib.ptr(x), tab, B, _lib.ptr(y), N, H, W, _lib.ptr(ws), need, 0, st))
    lib.ic_event_record(ev[0], st)
    for _ in range(reps): _lib.check(lib.ic_ae_res_stack_f32(_lib.ptr(x), tab, B, _lib.ptr(y), N, H, W, _lib.ptr(ws), need, 0, st))
    lib.ic_event_record(ev[1], st)
    ms = ctypes.c_float(); lib.ic_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
    return ms.value / reps / nl * 1e3
t_dist = []; t_same = []
for i in range(nl): t_dist += [ws_[i], sc, sh]; t_same += [ws_[0], sc, sh]
ta, tb = _lib.ptr_table(t_dist), _lib.ptr_table(t_same)
for rnd in range(3):
    print('distinct filters {:.2f} us per layer | one filter for all layers {:.2f} us per layer'.format(run(ta), run(tb)), flush=True)
