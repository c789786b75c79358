"""CPU oracle of ONE TRAINING STEP's loss and gradients (TEST INFRASTRUCTURE ONLY; see oracle/oracle.py).

Restates, with torch-CPU autograd supplying the derivatives, the training graph of the reference:
  code/train.py:101-106   enc = ae.encode(x, True); x_out = ae.decode(enc.qbar, True);
                          bc = pc.bitcost(stop_gradient(enc.qbar), enc.symbols, True, pad=centers[0])
  code/train.py:303-336   get_loss: H_real, H_mask, H_soft, pc_loss = beta * max(H_soft - H_target, 0), L2 terms
  code/train.py:352-394   Distortions: d_loss_scaled = K_ms_ssim * (1 - MS-SSIM)
  code/ms_ssim.py:3-186   the TF (fp32, separable, REFLECT-padded) MS-SSIM used as the training loss
  code/autoencoder.py:106-125  BatchNorm with is_training=True: batch statistics (biased variance), eps 1e-5
Parity with TF-1.4 is unpinned here exactly as in oracle.py; what this file pins is the build's HIP backward
kernels against an independent autograd evaluation of the same forward expression.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as O

MSSSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


# ---- MS-SSIM training loss (code/ms_ssim.py) ----------------------------------------------------------------

def _gauss_kernel(sigma, size):
    n = size // 2
    x = np.arange(-n, n + 1, 1.0)
    g = np.exp(-x * x / (2 * sigma * sigma))
    return g / np.sum(np.abs(g))


def _sep_valid(img, k1d):
    """depthwise separable VALID correlation of an NCHW tensor with k1d along W then H."""
    C = img.shape[1]
    k = torch.as_tensor(k1d, dtype=img.dtype)
    kw = k.view(1, 1, 1, -1).repeat(C, 1, 1, 1)
    kh = k.view(1, 1, -1, 1).repeat(C, 1, 1, 1)
    return F.conv2d(F.conv2d(img, kw, groups=C), kh, groups=C)


def _gaussian_blur(img, sigma, size):
    k = _gauss_kernel(sigma, size)
    total_pad = max(k.shape[0] - img.shape[3], 0)          # ms_ssim.py:19 reads shape[2] of the NHWC tensor of :160-162 = the WIDTH (NCHW here: [3])
    p1, p2 = total_pad + 1 // 2, total_pad // 2          # sic: `total_pad + 1 // 2` (ms_ssim.py:20)
    if p1 or p2:
        img = F.pad(img, (p1, p2, p1, p2), mode='reflect')
    return _sep_valid(img, k)


def _ssim_cs(a, b, max_val=255.0, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03):
    h, w = a.shape[2], a.shape[3]
    size = min(filter_size, h, w)
    sigma = size * filter_sigma / filter_size
    mu1, mu2 = _gaussian_blur(a, sigma, size), _gaussian_blur(b, sigma, size)
    s11 = _gaussian_blur(a * a, sigma, size) - mu1 * mu1
    s22 = _gaussian_blur(b * b, sigma, size) - mu2 * mu2
    s12 = _gaussian_blur(a * b, sigma, size) - mu1 * mu2
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    v1 = 2.0 * s12 + c2
    v2 = s11 + s22 + c2
    ssim = (((2.0 * mu1 * mu2 + c1) * v1) / ((mu1 * mu1 + mu2 * mu2 + c1) * v2)).mean()
    return ssim, (v1 / v2).mean()


def ms_ssim(a, b):
    """a, b: NCHW in 0..255.  -> scalar MS-SSIM (ms_ssim.py:115-186)."""
    w = MSSSIM_WEIGHTS
    mssim, mcs = [], []
    for _ in range(len(w)):
        s, c = _ssim_cs(a, b)
        mssim.append(s)
        mcs.append(c)
        # kernel_blur(ones(2)/2, pad=True): REFLECT pad (0, 1), VALID, then every second sample
        a, b = [_sep_valid(F.pad(t, (0, 1, 0, 1), mode='reflect'), np.ones(2) / 2.0)[:, :, ::2, ::2] for t in (a, b)]
    out = mssim[-1] ** w[-1]
    for l in range(len(w) - 1):
        out = out * mcs[l] ** w[l]
    return out


# ---- training-mode forward --------------------------------------------------------------------------------------

def _bn_train(x, p, scope):
    return F.batch_norm(x, None, None, p[scope + '/BatchNorm/gamma'], p[scope + '/BatchNorm/beta'],
                        training=True, momentum=0.1, eps=O.BN_EPS)


def _conv(x, w_tf, stride):
    kh, kw = w_tf.shape[0], w_tf.shape[1]
    pt, pb = O.same_pads(x.shape[2], kh, stride)
    pl, pr = O.same_pads(x.shape[3], kw, stride)
    return F.conv2d(F.pad(x, (pl, pr, pt, pb)), w_tf.permute(3, 2, 0, 1), stride=stride)


def _deconv(x, w_tf, stride=2):
    kh, kw = w_tf.shape[0], w_tf.shape[1]
    full = F.conv_transpose2d(x, w_tf.permute(3, 2, 0, 1), stride=stride)
    oh, ow = x.shape[2] * stride, x.shape[3] * stride
    pt, _ = O.same_pads(oh, kh, stride)
    pl, _ = O.same_pads(ow, kw, stride)
    nh, nw = pt + oh - full.shape[2], pl + ow - full.shape[3]
    if nh > 0 or nw > 0:
        full = F.pad(full, (0, max(nw, 0), 0, max(nh, 0)))
    return full[:, :, pt:pt + oh, pl:pl + ow]


def _cba(x, p, scope, stride, relu, transpose=False):
    y = _deconv(x, p[scope + '/weights']) if transpose else _conv(x, p[scope + '/weights'], stride)
    y = _bn_train(y, p, scope)
    return F.relu(y) if relu else y


def _res_block(x, p, scope, relu_first=True):
    y = _cba(x, p, scope + '/conv1', 1, relu_first)
    y = _cba(y, p, scope + '/conv2', 1, False)
    return y + x


def _res_stack(net, p, root, kind, B):
    final = 'res_block_enc_final' if kind == 'enc' else 'dec_after_res'
    res0 = net
    for b in range(B):
        res_b = net
        for i in (1, 2, 3):
            net = _res_block(net, p, '{}/res_block_{}_{}/{}_{}_{}'.format(root, kind, b, kind, b, i))
        net = net + res_b
    net = _res_block(net, p, '{}/{}'.format(root, final), relu_first=False)
    return net + res0


def _pc_bitcost(q, symbols, p, pad_value):
    first, other = O.pc_masks(3)
    dt = q.dtype
    mf = torch.as_tensor(first, dtype=dt)[..., None, None]
    mo = torch.as_tensor(other, dtype=dt)[..., None, None]

    def c3(x, scope, mask, relu):
        w = (p[scope + '/weights'] * mask).permute(4, 3, 0, 1, 2)
        y = F.conv3d(x, w, p[scope + '/biases'])
        return F.relu(y) if relu else y
    x = O.pad_for_probclass3d(q, 9, pad_value).unsqueeze(1)
    net = c3(x, O.PC + '/conv3d_conv0_mask', mf, True)
    r = net
    net = c3(net, O.PC + '/res1/conv3d_conv1_mask', mo, True)
    net = c3(net, O.PC + '/res1/conv3d_conv2_mask', mo, False) + r[:, :, 2:, 2:-2, 2:-2]
    logits = c3(net, O.PC + '/conv3d_conv2_mask', mo, True).permute(0, 2, 3, 4, 1)
    nll = -torch.gather(torch.log_softmax(logits, -1), -1, symbols.unsqueeze(-1)).squeeze(-1)
    return nll * float(np.log2(np.e))


def train_loss(x_uint8, weights, ae_cfg, pc_cfg, dtype=torch.float64):
    """-> (total_loss tensor, dict of components, dict name -> parameter tensor with .grad after backward()).
    ae_cfg / pc_cfg: dicts of the config values."""
    p = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=('moving_' not in k)) for k, v in weights.items()}
    x = torch.as_tensor(np.asarray(x_uint8)).to(dtype)
    B = ae_cfg['arch_param_B']
    C = ae_cfg['num_chan_bn']
    # encoder (autoencoder.py:218-244, is_training=True)
    net = O.normalize(x, ae_cfg.get('normalization', 'FIXED'))
    net = _cba(net, p, O.ENC + '/h1', 2, True)
    net = _cba(net, p, O.ENC + '/h2', 2, True)
    net = _res_stack(net, p, O.ENC, 'enc', B)
    net = _cba(net, p, O.ENC + '/to_bn', 2, False)
    hm = torch.clamp(torch.clamp(torch.sigmoid(net[:, 0:1]) * C - torch.arange(C, dtype=dtype).view(1, C, 1, 1), max=1.0), min=0.0)
    z = hm * net[:, 1:]
    c = p[O.ENC + '/centers']
    dist = (z.unsqueeze(-1) - c) ** 2
    qsoft = (torch.softmax(-dist, -1) * c).sum(-1)
    symbols = torch.argmin(dist.detach(), -1)
    qhard = c.detach()[symbols]
    qbar = qsoft + (qhard - qsoft).detach()
    # decoder on qbar (train.py:102)
    net = _cba(qbar, p, O.DEC + '/from_bn', 2, True, transpose=True)
    net = _res_stack(net, p, O.DEC, 'dec', B)
    net = _cba(net, p, O.DEC + '/h12', 2, True, transpose=True)
    net = _cba(net, p, O.DEC + '/h13', 2, False, transpose=True)
    x_out = torch.clamp(O.denormalize(net, ae_cfg.get('normalization', 'FIXED')), 0, 255)
    # context model on stop_gradient(qbar) (train.py:104-105)
    bc = _pc_bitcost(qbar.detach(), symbols, p, float(c.detach()[0]))
    # loss (train.py:303-336)
    # Distortions._get_distortion_to_minimize (train.py:379-390); in training the minimised metric stays float
    kind = ae_cfg.get('distortion_to_minimize', 'ms_ssim')
    mse_per_img = ((x_out - x) ** 2).mean(dim=(1, 2, 3))
    if kind == 'ms_ssim':
        d_loss = ae_cfg['K_ms_ssim'] * (1.0 - ms_ssim(x, x_out))
    elif kind == 'mse':
        d_loss = mse_per_img.mean()
    elif kind == 'psnr':
        d_loss = ae_cfg['K_psnr'] - (10.0 * torch.log10(255.0 * 255.0 / mse_per_img)).mean()
    else:
        raise ValueError('Invalid: {}'.format(kind))
    H_real = bc.mean()
    H_mask = (bc * hm).mean()
    H_soft = 0.5 * (H_mask + H_real)
    pc_loss = ae_cfg['beta'] * torch.clamp(H_soft - ae_cfg['H_target'], min=0.0)
    f = ae_cfg['regularization_factor']
    reg = sum(f * 0.5 * (t * t).sum() for k, t in p.items() if k.startswith('autoencoder/') and k.endswith('/weights'))
    reg = reg + ae_cfg['regularization_factor_centers'] * 0.5 * (c * c).sum()
    if pc_cfg.get('regularization_factor') is not None:
        reg = reg + sum(pc_cfg['regularization_factor'] * 0.5 * (t * t).sum()
                        for k, t in p.items() if k.startswith('probclass3d/') and k.endswith('/weights'))
    total = d_loss + pc_loss + reg
    comps = {'d_loss_scaled': d_loss, 'pc_loss': pc_loss, 'reg': reg, 'H_real': H_real, 'H_mask': H_mask,
             'x_out': x_out, 'symbols': symbols, 'bc': bc, 'heatmap': hm, 'z': z, 'qbar': qbar}
    return total, comps, p


def train_steps(x_uint8, weights, ae_cfg, pc_cfg, steps, dtype=torch.float64):
    """`steps` iterations of the reference's training loop on one fixed batch, float64 throughout:
      code/train.py:339-349      get_train_op: two tf.train.AdamOptimizer (AE variables with the AE config's lr, context model
                                 with its own), one train_op
      code/training_helpers.py:38-48   Adam with TF's defaults (beta1 0.9, beta2 0.999, epsilon 1e-8)
    tf.train.AdamOptimizer: lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t); m, v moving averages; var -= lr_t * m / (sqrt(v) + eps).
    The learning rates are the configs' lr_initial (the staircase decay does not move inside a handful of steps).
    -> (list of per-step dicts of floats, dict name -> final float64 numpy array).  The moving averages of BatchNorm play no
    part in a training-mode forward (batch statistics), so they are carried through unchanged."""
    b1, b2, eps = 0.9, 0.999, 1e-8
    cur = {k: np.asarray(v, dtype=np.float64).copy() for k, v in weights.items()}
    m = {k: np.zeros_like(v) for k, v in cur.items()}
    v_ = {k: np.zeros_like(v) for k, v in cur.items()}
    hist = []
    for t in range(1, steps + 1):
        total, comps, p = train_loss(x_uint8, cur, ae_cfg, pc_cfg, dtype)
        total.backward()
        K = float(ae_cfg['K_ms_ssim'])
        comps = {k_: (v.detach() if torch.is_tensor(v) else v) for k_, v in comps.items()}
        hist.append({'d_loss_scaled': float(comps['d_loss_scaled']), 'ms_ssim': 1.0 - float(comps['d_loss_scaled']) / K,
                     'pc_loss': float(comps['pc_loss']), 'H_real': float(comps['H_real']), 'H_mask': float(comps['H_mask']),
                     'bpp': float(comps['bc'].sum()) / (comps['x_out'].shape[0] * comps['x_out'].shape[2] * comps['x_out'].shape[3])})
        for k, t_ in p.items():
            if t_.grad is None:
                continue
            is_pc = k.startswith('probclass3d/')
            if (is_pc and not ae_cfg.get('train_probclass', True)) or (not is_pc and not ae_cfg.get('train_autoencoder', True)):
                continue
            lr = float(pc_cfg['lr_initial'] if is_pc else ae_cfg['lr_initial'])
            g = t_.grad.detach().numpy().astype(np.float64)
            m[k] = b1 * m[k] + (1 - b1) * g
            v_[k] = b2 * v_[k] + (1 - b2) * g * g
            lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
            cur[k] = cur[k] - lr_t * m[k] / (np.sqrt(v_[k]) + eps)
    return hist, cur
