#!/usr/bin/env python
"""Tuning: wall time of the inference step with / without the context model and with / without the second stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import autoencoder, probclass, bits, config_parser as cp, weights as W
dev = torch.device('cuda:0')
ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
wts = W.synthetic_weights(ae_cfg, pc_cfg)
ae = autoencoder.get_network_cls(ae_cfg)(ae_cfg).load_weights(wts, dev)
pc = probclass.get_network_cls(pc_cfg)(pc_cfg, num_centers=ae_cfg.num_centers).load_weights(wts, dev)
x = torch.as_tensor(W.synthetic_image((1, 3, 512, 768), 'natural', seed=0)).float().to(dev)
pad = float(wts['autoencoder/encoder/centers'][0])
side = torch.cuda.Stream(device=dev)
hi = torch.cuda.Stream(device=dev, priority=-1)

def run(mode):
    cur = torch.cuda.current_stream(dev)
    enc = ae.encode(x, False)
    if mode == 'serial':
        bc = pc.bitcost(enc.qbar, enc.symbols, False, pad_value=pad); bits.bitcost_to_bpp(bc, x)
        ae.decode(enc.qhard, False)
    elif mode == 'nopc':
        ae.decode(enc.qhard, False)
    elif mode == 'side':
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            bc = pc.bitcost(enc.qbar, enc.symbols, False, pad_value=pad); bits.bitcost_to_bpp(bc, x)
        ae.decode(enc.qhard, False)
        cur.wait_stream(side)
    elif mode == 'prio':             # decoder on a high-priority stream, context model stays on the current one
        hi.wait_stream(cur)
        with torch.cuda.stream(hi):
            ae.decode(enc.qhard, False)
        bc = pc.bitcost(enc.qbar, enc.symbols, False, pad_value=pad); bits.bitcost_to_bpp(bc, x)
        cur.wait_stream(hi)
    elif mode == 'prio2':            # whole autoencoder on the high-priority stream
        pass
    elif mode == 'side_after':       # decoder first, context model enqueued afterwards on the side stream
        ev = torch.cuda.Event(); ev.record(cur)
        ae.decode(enc.qhard, False)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            bc = pc.bitcost(enc.qbar, enc.symbols, False, pad_value=pad); bits.bitcost_to_bpp(bc, x)
        cur.wait_stream(side)

for mode in ('nopc', 'serial', 'side', 'prio', 'nopc', 'serial', 'side', 'prio'):
    for _ in range(3): run(mode)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): run(mode)
    torch.cuda.synchronize(); print(mode, round((time.perf_counter() - t) / 20 * 1e3, 3), 'ms')
