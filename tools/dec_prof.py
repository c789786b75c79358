#!/usr/bin/env python
"""Tuning: phase clocks of the persistent decoder (needs a -DPC_DEC_PROF build via IMGCOMP_HIP_LIB)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from imgcomp_cvpr_amd import autoencoder, probclass, config_parser as cp, weights as W, bit_counter, _lib
lib = _lib.lib
dev = torch.device('cuda:0')
ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
wts = W.synthetic_weights(ae_cfg, pc_cfg)
ae = autoencoder.get_network_cls(ae_cfg)(ae_cfg).load_weights(wts, dev)
pc = probclass.get_network_cls(pc_cfg)(pc_cfg, num_centers=ae_cfg.num_centers).load_weights(wts, dev)
x = torch.as_tensor(W.synthetic_image((1, 3, 128, 192), 'natural', seed=0)).float().to(dev)
sym = ae.encode(x, False).symbols[0].cpu().numpy()
pred = probclass.PredictionNetwork(pc, pc_cfg, ae.get_centers_variable())
padded = pred.pad_symbols_volume(sym)
fd, path = tempfile.mkstemp()
nbits, first, _ = bit_counter._encode(fd, padded, sym, pred)
data = open(path, 'rb').read()
C, h, w = sym.shape
d = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
out = torch.empty((C, h, w), dtype=torch.int64, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
need = lib.ic_pc_decode_workspace_bytes(C, h, w, 24)
ws = torch.zeros(need, dtype=torch.uint8, device=dev)
centers = ae.get_centers_variable().contiguous().float()
_lib.check(lib.ic_pc_decode_f32(_lib.ptr(d), len(data), int(first), pc._tab, _lib.ptr(centers), 24, 6, 1e9, _lib.ptr(out), _lib.ptr(status), C, h, w, _lib.ptr(ws), need, _lib.current_stream(dev)))
torch.cuda.synchronize()
assert np.array_equal(out.cpu().numpy(), sym)
# the state block sits after vol, ctx, logits in the workspace
al = lambda b: (b + 255) & ~255
off = al((C + 4) * (h + 8) * (w + 8) * 4) + al(405 * 4) + al(64)
ph = ws[off:off + 48].view(torch.int64).cpu().numpy().astype(np.float64) / (sym.size - 1)
print('clocks per symbol: gather %.0f  layer0 %.0f  conv1 %.0f  conv2 %.0f  final %.0f  coder %.0f   total %.0f' % (*ph, ph.sum()))
