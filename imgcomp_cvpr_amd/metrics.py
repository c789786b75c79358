"""Host-side validation metrics, restated from the definitions the reference uses in val.py:

  * MS-SSIM in float64 (reference code/ms_ssim_np.py:51-200, called through tf.py_func at val.py:93):
    5 scales, weights (0.0448, 0.2856, 0.3001, 0.2363, 0.1333), 11-tap Gaussian (sigma 1.5) 'valid' blur
    shrunk for small images, 2x2 box down-sampling with 'reflect' at the far edge.
  * PSNR on uint8 images, 10 log10(255^2 / MSE) (reference code/val.py:227-232).

The numpy versions run on the host exactly as in the reference and are pinned against outputs of the reference's
own implementation (tests/golden/msssim.npz).  val.py spends 0.5 s per Kodak image in them against 3.6 ms on the GPU
for everything else, so the same float64 computation is also available on the device (`*_device`, torch float64, the
same operations in the same order; agreement with the numpy path to ~1e-15 is tested).
"""
import numpy as np

_MSSSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _gauss1d(size, sigma):
    """normalised 1-D Gaussian whose outer product is the fspecial('gaussian') window."""
    r = size // 2
    if size % 2 == 0:
        x = np.arange(-r, r, dtype=np.float64) + 0.5
    else:
        x = np.arange(-r, r + 1, dtype=np.float64)
    g = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return g / g.sum()


def _blur_valid(img, g):
    """separable 'valid' correlation of an (N,H,W,C) float64 array with the symmetric window g x g."""
    k = g.size
    H, W = img.shape[1], img.shape[2]
    tmp = np.zeros((img.shape[0], H - k + 1, W, img.shape[3]), np.float64)
    for i in range(k):
        tmp += g[i] * img[:, i:i + H - k + 1, :, :]
    out = np.zeros((img.shape[0], H - k + 1, W - k + 1, img.shape[3]), np.float64)
    for i in range(k):
        out += g[i] * tmp[:, :, i:i + W - k + 1, :]
    return out


def _ssim_and_cs(img1, img2, max_val=255.0, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03):
    img1 = img1.astype(np.float64)
    img2 = img2.astype(np.float64)
    _, h, w, _ = img1.shape
    size = min(filter_size, h, w)
    sigma = size * filter_sigma / filter_size
    g = _gauss1d(size, sigma)
    mu1, mu2 = _blur_valid(img1, g), _blur_valid(img2, g)
    s11 = _blur_valid(img1 * img1, g) - mu1 * mu1
    s22 = _blur_valid(img2 * img2, g) - mu2 * mu2
    s12 = _blur_valid(img1 * img2, g) - mu1 * mu2
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    v1 = 2.0 * s12 + c2
    v2 = s11 + s22 + c2
    ssim = np.mean(((2.0 * mu1 * mu2 + c1) * v1) / ((mu1 * mu1 + mu2 * mu2 + c1) * v2))
    cs = np.mean(v1 / v2)
    return ssim, cs


def _downsample2(img):
    """average of (i, i+1) x (j, j+1) with the far edge reflected, then every second sample."""
    p = np.pad(img, ((0, 0), (0, 1), (0, 1), (0, 0)), mode='symmetric')
    box = 0.25 * (p[:, :-1, :-1] + p[:, 1:, :-1] + p[:, :-1, 1:] + p[:, 1:, 1:])
    return box[:, ::2, ::2, :]


def multiscale_ssim(img1, img2, max_val=255.0):
    """img1, img2: (N,H,W,C) arrays (uint8 or float).  -> float64 MS-SSIM."""
    if img1.shape != img2.shape:
        raise RuntimeError('Input images must have the same shape ({} vs. {}).'.format(img1.shape, img2.shape))
    if img1.ndim != 4:
        raise RuntimeError('Input images must have four dimensions, not {}'.format(img1.ndim))
    w = np.array(_MSSSIM_WEIGHTS)
    im1, im2 = img1.astype(np.float64), img2.astype(np.float64)
    mssim, mcs = [], []
    for _ in range(w.size):
        s, c = _ssim_and_cs(im1, im2, max_val=max_val)
        mssim.append(s)
        mcs.append(c)
        im1, im2 = _downsample2(im1), _downsample2(im2)
    mcs, mssim = np.array(mcs), np.array(mssim)
    return float(np.prod(mcs[:-1] ** w[:-1]) * (mssim[-1] ** w[-1]))


def msssim_nchw_uint8(x, y):
    """val.py's metric: uint8 NCHW batches -> float32 MS-SSIM (tf_msssim_np, data_format='NCHW')."""
    x = np.transpose(np.asarray(x), (0, 2, 3, 1))
    y = np.transpose(np.asarray(y), (0, 2, 3, 1))
    assert x.dtype == np.uint8 and y.dtype == np.uint8, 'Expected uint8 input'
    return np.float32(multiscale_ssim(x, y, max_val=255.0))


def psnr_uint8(a, b):
    """10 log10(255^2 / mean((a-b)^2)) on uint8 images (skimage compare_psnr with data_range 255)."""
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.dtype == np.uint8 and b.dtype == np.uint8, 'Expected uint8 input'
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    if mse == 0:
        return np.float32(np.inf)
    return np.float32(10.0 * np.log10(255.0 ** 2 / mse))


# ---- the same metrics in float64 on the device --------------------------------------------------------------------------

def _blur_valid_t(img, g):
    """torch twin of _blur_valid: (N,H,W,C) float64 tensor, taps accumulated in the same order."""
    k = len(g)
    H, W = img.shape[1], img.shape[2]
    tmp = g[0] * img[:, 0:H - k + 1, :, :]
    for i in range(1, k):
        tmp = tmp + g[i] * img[:, i:i + H - k + 1, :, :]
    out = g[0] * tmp[:, :, 0:W - k + 1, :]
    for i in range(1, k):
        out = out + g[i] * tmp[:, :, i:i + W - k + 1, :]
    return out


def _ssim_and_cs_t(img1, img2, max_val=255.0, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03):
    _, h, w, _ = img1.shape
    size = min(filter_size, h, w)
    g = [float(v) for v in _gauss1d(size, size * filter_sigma / filter_size)]
    mu1, mu2 = _blur_valid_t(img1, g), _blur_valid_t(img2, g)
    s11 = _blur_valid_t(img1 * img1, g) - mu1 * mu1
    s22 = _blur_valid_t(img2 * img2, g) - mu2 * mu2
    s12 = _blur_valid_t(img1 * img2, g) - mu1 * mu2
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    v1 = 2.0 * s12 + c2
    v2 = s11 + s22 + c2
    ssim = (((2.0 * mu1 * mu2 + c1) * v1) / ((mu1 * mu1 + mu2 * mu2 + c1) * v2)).mean()
    return ssim, (v1 / v2).mean()


def _downsample2_t(img):
    import torch
    p = torch.cat([img, img[:, -1:, :, :]], dim=1)                     # 'symmetric' pad by one = repeat the edge sample
    p = torch.cat([p, p[:, :, -1:, :]], dim=2)
    box = 0.25 * (p[:, :-1, :-1] + p[:, 1:, :-1] + p[:, :-1, 1:] + p[:, 1:, 1:])
    return box[:, ::2, ::2, :]


def msssim_nchw_uint8_device(x, y):
    """x, y: uint8 NCHW torch tensors on the device -> python float (float32-rounded like msssim_nchw_uint8)."""
    return msssim_from_scale_values(msssim_scale_values_device(x, y).tolist())       # one device -> host transfer


class ValMetricsWorkspace(object):
    """scratch of val_metrics_device, owned by the caller that issues the calls of ONE stream (val.Fetcher): grown on demand, reused"""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            import torch
            self.buf = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
        return self.buf


def val_metrics_device(x, y, workspace=None):
    """both metrics of val.py for uint8 NCHW tensors ON A HIP DEVICE in one call of the library (csrc/val_metrics.hip: float64,
    the numpy code's operations in the numpy code's order, 16 launches): -> (the 5 scale values msssim_from_scale_values takes,
    the mean squared error) as views of ONE float64 device tensor, nothing waited for.  workspace: a ValMetricsWorkspace of the
    calling stream (None: a fresh allocation for this call)."""
    import torch
    from . import _lib
    assert x.dtype == torch.uint8 and y.dtype == torch.uint8, 'Expected uint8 input'
    if x.shape != y.shape:
        raise RuntimeError('Input images must have the same shape ({} vs. {}).'.format(tuple(x.shape), tuple(y.shape)))
    _lib.require_cuda(x, 'x')
    x, y = x.contiguous(), y.contiguous()
    N, C, H, W = x.shape
    need = _lib.lib.ic_val_metrics_workspace_bytes(N, C, H, W)
    ws = (workspace or ValMetricsWorkspace()).get(need, x.device)
    out = torch.empty(6, dtype=torch.float64, device=x.device)
    _lib.check(_lib.lib.ic_val_metrics_u8_f64(_lib.ptr(x), _lib.ptr(y), N, C, H, W, _lib.ptr(out), _lib.ptr(ws), need,
                                              _lib.current_stream(x.device)), 'ic_val_metrics_u8_f64')
    return out[:5], out[5]


def val_metrics_device_per_image(x, y, workspace=None):
    """val_metrics_device of every image of a uint8 NCHW batch separately -> (N, 6) float64 device tensor, row n = [the 5 scale values,
    the mean squared error] of image n: the same kernels on the same operands as N single-image calls (bit-identical), one call of the
    library; nothing waited for."""
    import torch
    from . import _lib
    assert x.dtype == torch.uint8 and y.dtype == torch.uint8, 'Expected uint8 input'
    if x.shape != y.shape:
        raise RuntimeError('Input images must have the same shape ({} vs. {}).'.format(tuple(x.shape), tuple(y.shape)))
    _lib.require_cuda(x, 'x')
    x, y = x.contiguous(), y.contiguous()
    N, C, H, W = x.shape
    need = _lib.lib.ic_val_metrics_workspace_bytes(1, C, H, W)
    ws = (workspace or ValMetricsWorkspace()).get(need, x.device)
    out = torch.empty((N, 6), dtype=torch.float64, device=x.device)
    _lib.check(_lib.lib.ic_val_metrics_per_image_u8_f64(_lib.ptr(x), _lib.ptr(y), N, C, H, W, _lib.ptr(out), _lib.ptr(ws), need,
                                                        _lib.current_stream(x.device)), 'ic_val_metrics_per_image_u8_f64')
    return out


def msssim_scale_values_device(x, y):
    """the device half of msssim_nchw_uint8_device: the per-scale contrast terms and the last scale's SSIM as ONE float64 device
    tensor, nothing waited for (val.py keeps several images in flight and reads the values later)"""
    import torch
    assert x.dtype == torch.uint8 and y.dtype == torch.uint8, 'Expected uint8 input'
    if x.shape != y.shape:
        raise RuntimeError('Input images must have the same shape ({} vs. {}).'.format(tuple(x.shape), tuple(y.shape)))
    if x.is_cuda:
        return val_metrics_device(x, y)[0]
    # (CPU tensors: the torch twin of the numpy code, tap by tap)
    im1 = x.permute(0, 2, 3, 1).to(torch.float64)
    im2 = y.permute(0, 2, 3, 1).to(torch.float64)
    mssim, mcs = [], []
    for _ in range(len(_MSSSIM_WEIGHTS)):
        s, c = _ssim_and_cs_t(im1, im2)
        mssim.append(s)
        mcs.append(c)
        im1, im2 = _downsample2_t(im1), _downsample2_t(im2)
    return torch.stack(mcs[:-1] + [mssim[-1]])


def msssim_from_scale_values(vals):
    """the host half: prod(mcs ** w) * mssim_last ** w_last, float32-rounded like msssim_nchw_uint8"""
    w = _MSSSIM_WEIGHTS
    out = 1.0
    for v, wt in zip(vals[:-1], w[:-1]):
        out *= v ** wt
    return float(np.float32(out * vals[-1] ** w[-1]))


def mse_uint8_device(a, b):
    import torch
    assert a.dtype == torch.uint8 and b.dtype == torch.uint8, 'Expected uint8 input'
    if a.is_cuda and a.dim() == 4:
        return val_metrics_device(a, b)[1]
    return ((a.to(torch.float64) - b.to(torch.float64)) ** 2).mean()


def psnr_from_mse(mse):
    if mse == 0:
        return float('inf')
    return float(np.float32(10.0 * np.log10(255.0 ** 2 / mse)))


def psnr_uint8_device(a, b):
    return psnr_from_mse(float(mse_uint8_device(a, b)))
