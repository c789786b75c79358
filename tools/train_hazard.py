"""Bisect the training step's hazard (ms_ssim > 1, trajectories that differ run to run).

For each arrangement of the loss (graph + side stream | graph on the main stream | eager) run the same seeded cfg3 trajectory
twice from fresh Trainers and print, per step: the step's own ms_ssim, an eager float64 MS-SSIM of the step's x / x_out, and a
checksum of the variables after the update.  Equal checksums = reproducible; graph value == eager value = the graph is sound.

  python tools/train_hazard.py [--steps 15] [--batch 32] [--modes overlap,graph,eager]
"""
import argparse
import hashlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

import torch

from imgcomp_cvpr_amd import config_parser as cp, weights as W, training, ms_ssim


def digest(tr):
    h = hashlib.sha1()
    for g in ('pc', 'dec', 'enc'):
        h.update(tr.graph.flat_params[g].detach().cpu().numpy().tobytes())
    return h.hexdigest()[:12]


def trajectory(mode, steps, batch, dev, size=128, sync=True):
    ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'med'))
    pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
    wts = W.synthetic_weights(ae_cfg, pc_cfg)
    tr = training.Trainer(ae_cfg, pc_cfg, wts, dev, num_itr_per_epoch=1000)
    tr.graph.HIP_LOSS = mode == 'hip'              # csrc/msssim.hip (the default since round 4)
    tr.graph.GRAPH_LOSS = mode in ('overlap', 'graph')
    tr.graph.OVERLAP_LOSS = mode == 'overlap'
    x = torch.as_tensor(W.synthetic_image((batch, 3, size, size), 'natural', seed=0)).float().to(dev)
    rows = []
    for s in range(steps):
        out = tr.step(x)
        if not sync:                      # what bench.py --mode train does: nothing between the steps
            rows.append((out['ms_ssim'], float('nan'), out['d_loss_scaled'], out['bpp'], digest(tr) if s == steps - 1 else '-'))
            continue
        xo = tr.graph.last['x_out']
        ref = float(ms_ssim.multiscale_ssim(x.double(), xo.double()))
        torch.cuda.synchronize(dev)
        rows.append((out['ms_ssim'], ref, out['d_loss_scaled'], out['bpp'], digest(tr)))
    return rows


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--steps', type=int, default=15)
    p.add_argument('--batch', type=int, default=32)
    p.add_argument('--modes', default='hip,overlap,graph,eager')
    p.add_argument('--nosync', action='store_true', help='no synchronize / read-back between the steps (bench.py --mode train)')
    a = p.parse_args()
    dev = torch.device('cuda', 0)
    for mode in a.modes.split(','):
        runs = [trajectory(mode, a.steps, a.batch, dev, sync=not a.nosync) for _ in range(2)]
        print('== mode', mode)
        for s in range(a.steps):
            r0, r1 = runs[0][s], runs[1][s]
            print('step {:2d}  ms_ssim {:.6f} (eager f64 {:.6f})  d_loss {:10.4f}  bpp {:.5f}  {}  | run2 ms_ssim {:.6f} {} {}'.format(
                s, r0[0], r0[1], r0[2], r0[3], r0[4], r1[0], r1[4], 'same' if r0[4] == r1[4] else 'DIFFERENT'), flush=True)


if __name__ == '__main__':
    main()
