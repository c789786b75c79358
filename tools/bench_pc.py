#!/usr/bin/env python
"""Micro-benchmark: the parallel context-model pass (bitcost) at the Kodak symbol-volume shape, HIP events."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import probclass, config_parser as cp, weights as W, _lib
lib = _lib.lib
dev = torch.device('cuda:0'); st = _lib.current_stream(dev)
ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
wts = W.synthetic_weights(ae_cfg, pc_cfg)
pc = probclass.get_network_cls(pc_cfg)(pc_cfg, num_centers=ae_cfg.num_centers).load_weights(wts, dev)
N, C, h, w = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (1, 32, 64, 96)))
centers = torch.as_tensor(wts['autoencoder/encoder/centers']).to(dev)
sym = torch.randint(0, 6, (N, C, h, w), device=dev)
q = centers[sym].contiguous()
ev = [ctypes.c_void_p(), ctypes.c_void_p()]
for e in ev:
    _lib.check(lib.ic_event_create(ctypes.byref(e)))
f = lambda: pc.bitcost(q, sym, False, pad_value=float(centers[0]))
for _ in range(3): f()
torch.cuda.synchronize()
lib.ic_event_record(ev[0], st)
for _ in range(20): f()
lib.ic_event_record(ev[1], st)
ms = ctypes.c_float(); _lib.check(lib.ic_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms)))
print('bitcost N={} C={} {}x{}: {:.1f} us per pass'.format(N, C, h, w, ms.value / 20 * 1e3))
