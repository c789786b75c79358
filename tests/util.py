"""Shared helpers for the parity tests."""
import numpy as np
import torch

# Parity bar.  BASELINE.json north_star asks for "within 1e-4 fp32"; the kernels achieve far better, and an assert with
# 10-100x of slack would hide a regression, so the default bound is what is measured with a 2x margin: the HIP result
# must agree with the float64 evaluation of the oracle to 2e-5 relative to the tensor's scale, max(1, max|ref|)
# (round-2 GPU runs: single ops <= 3e-6, whole-network z / x_out / bit cost <= 1e-5).  Every comparison is recorded --
# absolute and relative maximum error -- and printed in the terminal summary (tests/conftest.py), so the log of a run shows
# the achieved accuracy, not just "passed".
RTOL = 2e-5                 # a single op
NET_RTOL = 5e-5             # a whole network (35 layers deep): the errors of the layers add up (measured <= 2.7e-5)
HEATMAP_RTOL = 1e-4         # heatmap = clip(sigmoid(z0) * C - c, 0, 1): amplifies the error of z0 by up to C / 4 (8 .. 16 x)
NORTH_STAR_RTOL = 1e-4
REPORT = []          # (label, max abs error, relative error, bound)
FLIPS = []           # (label, flipped symbols, total symbols)


def rel_err(got, ref64):
    got = got.detach().double().cpu() if torch.is_tensor(got) else torch.as_tensor(got).double()
    ref64 = ref64.detach().double().cpu() if torch.is_tensor(ref64) else torch.as_tensor(ref64).double()
    assert got.shape == ref64.shape, (got.shape, ref64.shape)
    scale = max(1.0, float(ref64.abs().max()))
    return float((got - ref64).abs().max()) / scale


def abs_err(got, ref64):
    got = got.detach().double().cpu() if torch.is_tensor(got) else torch.as_tensor(got).double()
    ref64 = ref64.detach().double().cpu() if torch.is_tensor(ref64) else torch.as_tensor(ref64).double()
    return float((got - ref64).abs().max())


def assert_close(got, ref64, what, rtol=RTOL):
    e = rel_err(got, ref64)
    REPORT.append((what, abs_err(got, ref64), e, rtol))
    assert e <= rtol, '{}: max error {:.3e} (relative to tensor scale; absolute {:.3e}) exceeds {:.1e}'.format(
        what, e, REPORT[-1][1], rtol)
    return e


def record_flips(what, flips):
    """symbol flips between the fp32 device path and the float64 oracle (they are only legitimate inside the fp32 error band
    of a quantiser decision midpoint); the rate goes into the terminal summary."""
    flips = np.asarray(flips)
    FLIPS.append((what, int(flips.sum()), int(flips.size)))
    return float(flips.mean())


def dev(a, device, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).to(device)


MSSSIM_CASES = {'a': (176, 208, 0), 'b': (256, 256, 1), 'c': (192, 320, 2)}


def msssim_case(name):
    """seeded NHWC uint8 image pair shared by tests/golden/make_golden.py and the metric tests."""
    h, w, k = MSSSIM_CASES[name]
    rs = np.random.RandomState(100 + k)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 60 * np.sin(xx / (7.0 + c) + yy / 11.0) + 30 * np.cos(yy / (5.0 + c))
                     for c in range(3)], -1)
    img1 = np.clip(base + rs.normal(0, 6, base.shape), 0, 255).astype(np.uint8)[None]
    img2 = np.clip(img1.astype(np.float64) + rs.normal(0, 3 + 4 * k, img1.shape), 0, 255).astype(np.uint8)
    return img1, img2
