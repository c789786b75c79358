"""bit cost -> bits per pixel; mirror of the reference's code/bits.py."""
import torch

from . import _lib
from ._lib import lib, check, ptr


def num_pixels_in_input_batch(input_batch):
    """N*H*W of an N3HW batch (reference code/bits.py:17-20)."""
    assert input_batch.dim() == 4 and int(input_batch.shape[1]) == 3, 'Expected N3HW, got {}'.format(
        tuple(input_batch.shape))
    return input_batch.numel() // 3


def bitcost_to_bpp(bit_cost, input_batch):
    """sum(bit_cost) / num_pixels -> 0-d float32 device tensor (reference code/bits.py:4-14).
    Sum and division are one deterministic two-stage tree reduction on the device (ic_mean_f32)."""
    assert bit_cost.dim() == input_batch.dim() == 4, 'Expected NChw and N3HW'
    _lib.require_cuda(bit_cost, 'bit_cost')
    bit_cost = bit_cost.contiguous()
    buf = torch.empty(1024 + 1, dtype=torch.float32, device=bit_cost.device)          # [partial sums | result]
    out = buf[1024:]
    check(lib.ic_mean_f32(ptr(bit_cost), bit_cost.numel(), float(num_pixels_in_input_batch(input_batch)), ptr(buf), ptr(out),
                          _lib.current_stream(bit_cost.device)), 'ic_mean_f32')
    return out[0]


def bitcost_to_bpp_per_image(bit_cost, input_batch):
    """bitcost_to_bpp of every image of the batch separately -> (N,) float32 device tensor: the same reduction per image as N calls on
    batches of one (val.py reports per-image measures of a batched step); one call of the library."""
    assert bit_cost.dim() == input_batch.dim() == 4 and bit_cost.shape[0] == input_batch.shape[0], 'Expected NChw and N3HW'
    _lib.require_cuda(bit_cost, 'bit_cost')
    bit_cost = bit_cost.contiguous()
    n = int(bit_cost.shape[0])
    buf = torch.empty(1024 + n, dtype=torch.float32, device=bit_cost.device)          # [partial sums | results]
    out = buf[1024:]
    check(lib.ic_mean_rows_f32(ptr(bit_cost), n, bit_cost.numel() // n, float(num_pixels_in_input_batch(input_batch[:1])), ptr(buf), ptr(out),
                               _lib.current_stream(bit_cost.device)), 'ic_mean_rows_f32')
    return out

