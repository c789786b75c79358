#!/usr/bin/env python
"""Throughput of the inference step with S independent batch-1 pipelines in flight (one HIP stream and one set of
workspaces each): Kodak-shaped images fill only 768 of the 1024 SIMDs per 3x3 launch; kernels of other streams use the rest."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import autoencoder, probclass, bits, config_parser as cp, weights as W


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--streams', default='1,2,3,4,6')
    p.add_argument('--steps', type=int, default=24)
    p.add_argument('--height', type=int, default=512)
    p.add_argument('--width', type=int, default=768)
    a = p.parse_args()
    dev = torch.device('cuda:0')
    ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
    pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
    wts = W.synthetic_weights(ae_cfg, pc_cfg)
    pad = float(wts['autoencoder/encoder/centers'][0])
    x = torch.as_tensor(W.synthetic_image((1, 3, a.height, a.width), 'natural', seed=0)).float().to(dev)
    smax = max(int(s) for s in a.streams.split(','))
    nets = []
    for _ in range(smax):
        ae = autoencoder.get_network_cls(ae_cfg)(ae_cfg).load_weights(wts, dev)
        pc = probclass.get_network_cls(pc_cfg)(pc_cfg, num_centers=ae_cfg.num_centers).load_weights(wts, dev)
        nets.append((ae, pc, torch.cuda.Stream(device=dev)))

    def step(ae, pc):
        enc = ae.encode(x, False)
        bc = pc.bitcost(enc.qbar, enc.symbols, False, pad_value=pad)
        bits.bitcost_to_bpp(bc, x)
        return ae.decode(enc.qhard, False)

    for S in (int(s) for s in a.streams.split(',')):
        for i in range(2 * S):
            ae, pc, st = nets[i % S]
            with torch.cuda.stream(st):
                step(ae, pc)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(a.steps):
            ae, pc, st = nets[i % S]
            with torch.cuda.stream(st):
                step(ae, pc)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print('streams {}: {:.3f} ms per image  {:.1f} Mpix/s'.format(S, dt / a.steps * 1e3, a.height * a.width * a.steps / dt / 1e6))


if __name__ == '__main__':
    main()
