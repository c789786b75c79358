#!/usr/bin/env python
"""Freezes the float64 ORACLE's end-to-end outputs for one seeded case into oracle_e2e.npz.

This is a regression pin of the oracle against itself (it does not involve the reference -- TF cannot run here, see
oracle/oracle.py): the CPU suite checks that today's oracle still produces these numbers, the GPU suite checks the HIP path
against the frozen numbers as well as against the live oracle.

usage: python tests/golden/make_oracle_e2e.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from imgcomp_cvpr_amd import config_parser as cp, weights as W    # noqa: E402
from oracle import oracle as O                                     # noqa: E402

SHAPE, IMG_SEED = (1, 3, 64, 96), 3


def compute():
    ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
    pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
    wts = W.synthetic_weights(ae_cfg, pc_cfg)
    x = W.synthetic_image(SHAPE, 'natural', seed=IMG_SEED)
    enc = O.encode(torch.as_tensor(x).double(), wts, ae_cfg.as_dict())
    centers = wts['autoencoder/encoder/centers']
    q = torch.as_tensor(centers)[enc.symbols].double()
    bc, _ = O.bitcost(q, enc.symbols, wts, float(centers[0]))
    x_out = O.decode(q, wts, ae_cfg.as_dict())
    return {'z': enc.z.numpy(), 'heatmap': enc.heatmap.numpy(), 'symbols': enc.symbols.numpy().astype(np.int8),
            'bitcost': bc.numpy(), 'x_out': x_out.numpy(), 'bpp': np.float64(O.bitcost_to_bpp(bc, torch.as_tensor(x)))}


if __name__ == '__main__':
    out = compute()
    np.savez_compressed(os.path.join(HERE, 'oracle_e2e.npz'), **out)
    print({k: (v.shape, float(np.abs(v).max())) for k, v in out.items()})
