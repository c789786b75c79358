#!/usr/bin/env python
"""Experiment: the inference step replayed from a captured HIP graph vs launched kernel by kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import autoencoder, probclass, bits, config_parser as cp, weights as W
dev = torch.device('cuda:0')
H, Wd = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 768)
ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
wts = W.synthetic_weights(ae_cfg, pc_cfg)
ae = autoencoder.get_network_cls(ae_cfg)(ae_cfg).load_weights(wts, dev)
pc = probclass.get_network_cls(pc_cfg)(pc_cfg, num_centers=ae_cfg.num_centers).load_weights(wts, dev)
x = torch.as_tensor(W.synthetic_image((1, 3, H, Wd), 'natural', seed=0)).float().to(dev)
pad = float(wts['autoencoder/encoder/centers'][0])

def step():
    enc = ae.encode(x, False)
    bc = pc.bitcost(enc.qbar, enc.symbols, False, pad_value=pad)
    bpp = bits.bitcost_to_bpp(bc, x)
    return bpp, ae.decode(enc.qhard, False)

def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

print('eager  %.3f ms' % timeit(step))
s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    out = step()
torch.cuda.synchronize()
print('graph  %.3f ms' % timeit(g.replay))
