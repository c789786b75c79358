"""Scalar quantiser plugin -- mirror of the reference's code/quantizer.py on the HIP device.

    qsoft, qhard, symbols = quantize(x, centers, sigma)       (reference code/quantizer.py:37-40)

x: (N,C,H,W) float32 device tensor; centers: (L,) float32 device tensor; symbols int64.
The arithmetic runs in ic_quantize_f32 (csrc/quantize.hip).
"""
import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr

_HARD_SIGMA = 1e7      # reference: code/quantizer.py:5 (compiled into the kernel)


def create_centers_variable(config, device='cuda'):
    """(L,) float32 ~ U(centers_initial_range), seed 666 (reference code/quantizer.py:11-31).
    TF's random stream cannot be reproduced; numpy RandomState(666) is used instead."""
    assert config.num_centers is not None
    minval, maxval = map(int, config.centers_initial_range)
    c = np.random.RandomState(666).uniform(minval, maxval, size=(config.num_centers,)).astype(np.float32)
    return torch.from_numpy(c).to(device)


def create_centers_regularization_term(config, centers):
    """reg * l2_loss(centers) = reg * sum(c^2) / 2 (reference code/quantizer.py:18-24)."""
    if config.regularization_factor_centers == 0:
        return torch.zeros((), dtype=torch.float32, device=centers.device)
    return float(config.regularization_factor_centers) * 0.5 * (centers * centers).sum()


def quantize(x, centers, sigma):
    """:return qsoft, qhard, symbols -- each of x's shape; symbols int64."""
    assert x.dtype == torch.float32, 'x should be float32'
    assert centers.dtype == torch.float32, 'centers should be float32'
    assert x.dim() == 4, 'x should be NCHW, got {}'.format(tuple(x.shape))
    assert centers.dim() == 1, 'centers should be (L,), got {}'.format(tuple(centers.shape))
    _lib.require_cuda(x, 'x')
    _lib.require_cuda(centers, 'centers')
    x = x.contiguous()
    qsoft = torch.empty_like(x)
    qhard = torch.empty_like(x)
    symbols = torch.empty(x.shape, dtype=torch.int64, device=x.device)
    check(lib.ic_quantize_f32(ptr(x), ptr(centers.contiguous()), centers.numel(), float(sigma),
                              ptr(qsoft), ptr(qhard), ptr(symbols), x.numel(),
                              _lib.current_stream(x.device)), 'ic_quantize_f32')
    return qsoft, qhard, symbols
