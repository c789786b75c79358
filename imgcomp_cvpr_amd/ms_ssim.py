"""Differentiable MS-SSIM used as the TRAINING loss -- host-side PyTorch (north_star: "Host code stays Python on
PyTorch-ROCm for tensor plumbing and the MS-SSIM loss"), restating the reference's TF graph code/ms_ssim.py:

  * 5 scales, weights (0.0448, 0.2856, 0.3001, 0.2363, 0.1333), result = prod_{l<4} cs_l^w_l * ssim_4^w_4 (:176-186)
  * per scale: separable 1-D Gaussian (sigma = size * 1.5 / 11, size = min(11, H, W), kernel LENGTH 2*(size//2)+1,
    :3-11, :81-90), 'VALID', the image REFLECT-padded first when it is smaller than the kernel, with the reference's
    asymmetric pads `total_pad + 1 // 2` (= total_pad) before and `total_pad // 2` after (:19-22)
  * between scales: 2-tap box [0.5, 0.5] along both axes after a REFLECT pad of (0, 1), then every second sample
    (:46-51, :169-175)
  * ssim / cs are means over batch, positions and channels (:110-113)
Works on any device; the validation METRIC is the float64 numpy version in metrics.py (code/ms_ssim_np.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _kernel1d(sigma, size):
    n = size // 2
    x = np.arange(-n, n + 1, 1.0)
    g = np.exp(-x * x / (2 * sigma * sigma))
    return g / np.sum(np.abs(g))


_BANDS = {}


def _band(n, k, dtype, device):
    """(n, n - len(k) + 1) matrix B with B[i + j, i] = k[j]: right-multiplying by it is a 'VALID' correlation."""
    key = (n, tuple(float(v) for v in k), dtype, str(device))
    if key not in _BANDS:
        _BANDS[key] = _make_band(n, k, dtype, device)
    return _BANDS[key]


def _make_band(n, k, dtype, device):
    m = n - len(k) + 1
    B = torch.zeros((n, m), dtype=dtype, device=device)
    idx = torch.arange(m, device=device)
    for j, kv in enumerate(k):
        B[idx + j, idx] = float(kv)
    return B


def _separable_valid(img, k):
    """depthwise 'VALID' correlation with the 1-D kernel k along W, then along H (img: NCHW), written as two dense
    matmuls against banded matrices (rocBLAS) -- depthwise F.conv2d falls onto MIOpen's naive kernels on ROCm."""
    k = np.asarray(k, dtype=np.float64)
    Bw = _band(img.shape[3], k, img.dtype, img.device)
    Bh = _band(img.shape[2], k, img.dtype, img.device)
    return torch.matmul(Bh.t(), torch.matmul(img, Bw))


def _blur(img, sigma, size):
    k = _kernel1d(sigma, size)
    total_pad = max(len(k) - img.shape[3], 0)       # the reference reads shape[2] of its NHWC tensor (ms_ssim.py:19, :160-162): the WIDTH
    before, after = total_pad + 1 // 2, total_pad // 2
    if before or after:
        img = F.pad(img, (before, after, before, after), mode='reflect')
    return _separable_valid(img, k)


def _ssim_and_cs(a, b, max_val, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03):
    size = min(filter_size, a.shape[2], a.shape[3])
    sigma = size * filter_sigma / filter_size
    mu_a, mu_b = _blur(a, sigma, size), _blur(b, sigma, size)
    var_a = _blur(a * a, sigma, size) - mu_a * mu_a
    var_b = _blur(b * b, sigma, size) - mu_b * mu_b
    cov = _blur(a * b, sigma, size) - mu_a * mu_b
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    v1, v2 = 2.0 * cov + c2, var_a + var_b + c2
    ssim = (((2.0 * mu_a * mu_b + c1) * v1) / ((mu_a * mu_a + mu_b * mu_b + c1) * v2)).mean()
    return ssim, (v1 / v2).mean()


def _halve(img):
    img = F.pad(img, (0, 1, 0, 1), mode='reflect')
    return _separable_valid(img, [0.5, 0.5])[:, :, ::2, ::2]


def multiscale_ssim(img1, img2, max_val=255.0):
    """img1, img2: (N,C,H,W) float tensors in [0, max_val] -> 0-d tensor (differentiable)."""
    if img1.shape != img2.shape:
        raise RuntimeError('Input images must have the same shape ({} vs. {}).'.format(img1.shape, img2.shape))
    if img1.dim() != 4:
        raise RuntimeError('Input images must have four dimensions, not {}'.format(img1.dim()))
    a, b = img1, img2
    ssims, css = [], []
    for level in range(len(WEIGHTS)):
        s, c = _ssim_and_cs(a, b, max_val)
        ssims.append(s)
        css.append(c)
        if level + 1 < len(WEIGHTS):
            a, b = _halve(a), _halve(b)
    out = ssims[-1] ** WEIGHTS[-1]
    for l in range(len(WEIGHTS) - 1):
        out = out * css[l] ** WEIGHTS[l]
    return out
