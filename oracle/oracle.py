"""CPU oracle: restatement of the imgcomp-cvpr hot path (TEST INFRASTRUCTURE ONLY).

This module is the *checker*.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product path
(``imgcomp_cvpr_amd``) never imports anything from ``oracle/`` and fails loudly when
its HIP library is missing.

PARITY PINNING STATUS
---------------------
The reference's arithmetic for this path lives in tensorflow-gpu==1.4.1 /
tf.contrib.slim and fjcommon==0.1.69 (reference ``requirements.txt:9,12``), none of
which is installed or installable here, and the reference holds no tests / golden
tensors for the path.  So **parity with TF-1.4 is unpinned** for the conv / BN /
conv3d arithmetic: this file restates the documented TF/slim semantics that the
reference's call sites rely on (each function cites its call site).  What *is*
pinned against the reference run in this container (see ``tests/golden/make_golden.py``):
the context-model masks, the pad geometry, the block iterator, MS-SSIM (numpy) and
the arithmetic coder.

Conventions: tensors are NCHW torch CPU tensors; ``dtype`` is torch.float32 (the
reference's precision) or torch.float64 (shadow used to bound fp32 error).
Weights are a ``dict name -> numpy array`` in the reference's TF variable layouts
and names.
"""
from collections import namedtuple

import numpy as np
import torch
import torch.nn.functional as F

# reference: code/autoencoder.py:15,18
EncoderOutput = namedtuple('EncoderOutput', ['qbar', 'qhard', 'symbols', 'z', 'heatmap'])

HARD_SIGMA = 1e7          # reference: code/quantizer.py:5
BN_EPS = 1e-5             # reference: code/autoencoder.py:118
ARCH_PARAM_N = 128        # reference: code/autoencoder.py:211

# reference: code/autoencoder.py:162-163
_MEAN = np.array([121.85369873, 113.58860779, 100.63715363], dtype=np.float32)
_VAR = np.array([4746.37695312, 4454.13964844, 4812.234375], dtype=np.float32)


def _t(a, dtype):
    return torch.as_tensor(np.asarray(a)).to(dtype)


# ----------------------------------------------------------------------------------
# normalisation  (reference: code/autoencoder.py:136-169)
# ----------------------------------------------------------------------------------

def norm_consts(dtype):
    """mean and sqrt(var + 1e-10) as the reference computes them: the float32 numpy
    arrays are combined with the python float 1e-10 *in float32* (numpy keeps the
    array dtype), then np.sqrt in float32 (autoencoder.py:143)."""
    std = np.sqrt(_VAR + np.float32(1e-10)).astype(np.float32)
    return _t(_MEAN, dtype).view(1, 3, 1, 1), _t(std, dtype).view(1, 3, 1, 1)


def normalize(x, style='FIXED'):
    if style == 'OFF':
        return x
    mean, std = norm_consts(x.dtype)
    return (x - mean) / std


def denormalize(x, style='FIXED'):
    if style == 'OFF':
        return x
    mean, std = norm_consts(x.dtype)
    return x * std + mean


# ----------------------------------------------------------------------------------
# conv primitives with TF 'SAME' semantics
# ----------------------------------------------------------------------------------

def same_pads(in_size, k, stride):
    """TF SAME padding: out = ceil(in/stride); total = max((out-1)*stride + k - in, 0);
    before = total // 2, after = total - before."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    return total // 2, total - total // 2


def conv2d_same(x, w_tf, stride):
    """slim.conv2d(..., padding='SAME') without bias (a normalizer_fn is set, so slim
    creates no bias).  w_tf is [kh, kw, cin, cout].  reference call sites:
    code/autoencoder.py:222,223,237,285."""
    kh, kw, cin, cout = w_tf.shape
    w = _t(w_tf, x.dtype).permute(3, 2, 0, 1).contiguous()
    pt, pb = same_pads(x.shape[2], kh, stride)
    pl, pr = same_pads(x.shape[3], kw, stride)
    x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, w, stride=stride)


def conv2d_transpose_same(x, w_tf, stride):
    """slim.conv2d_transpose(..., padding='SAME'): output size = stride * input size; it
    is the adjoint (gradient wrt input) of the SAME forward conv mapping
    (stride*in) -> in.  w_tf is [kh, kw, cout, cin] (TF conv2d_transpose filter layout:
    height, width, output_channels, in_channels).  reference call sites:
    code/autoencoder.py:251,264,265."""
    kh, kw, cout, cin = w_tf.shape
    assert cin == x.shape[1], (w_tf.shape, x.shape)
    w = _t(w_tf, x.dtype).permute(3, 2, 0, 1).contiguous()   # torch: (in, out, kh, kw)
    full = F.conv_transpose2d(x, w, stride=stride)             # size (in-1)*s + k
    oh, ow = x.shape[2] * stride, x.shape[3] * stride
    pt, _ = same_pads(oh, kh, stride)
    pl, _ = same_pads(ow, kw, stride)
    # the full transposed conv may be shorter than pt+oh at the far end (k=3: full =
    # 2*in+1, slice [0:2in]); pad with zeros if ever needed.
    need_h, need_w = pt + oh - full.shape[2], pl + ow - full.shape[3]
    if need_h > 0 or need_w > 0:
        full = F.pad(full, (0, max(need_w, 0), 0, max(need_h, 0)))
    return full[:, :, pt:pt + oh, pl:pl + ow]


def bn_scale_shift(weights, scope, dtype=None):
    """Inference BatchNorm folded to y = x * scale + shift (per channel):
    scale = gamma / sqrt(moving_variance + eps), shift = beta - moving_mean * scale.
    reference: code/autoencoder.py:106-125 (decay 0.9, epsilon 1e-5, scale=True)."""
    g = weights[scope + '/BatchNorm/gamma'].astype(np.float64)
    b = weights[scope + '/BatchNorm/beta'].astype(np.float64)
    m = weights[scope + '/BatchNorm/moving_mean'].astype(np.float64)
    v = weights[scope + '/BatchNorm/moving_variance'].astype(np.float64)
    scale = g / np.sqrt(v + BN_EPS)
    shift = b - m * scale
    return scale, shift


def batch_norm_infer(x, weights, scope):
    """slim.batch_norm(is_training=False): gamma * (x - mean) / sqrt(var + eps) + beta."""
    dt = x.dtype
    g = _t(weights[scope + '/BatchNorm/gamma'], dt).view(1, -1, 1, 1)
    b = _t(weights[scope + '/BatchNorm/beta'], dt).view(1, -1, 1, 1)
    m = _t(weights[scope + '/BatchNorm/moving_mean'], dt).view(1, -1, 1, 1)
    v = _t(weights[scope + '/BatchNorm/moving_variance'], dt).view(1, -1, 1, 1)
    return (x - m) * (g * torch.rsqrt(v + BN_EPS)) + b


def conv_bn_act(x, weights, scope, stride, relu, transpose=False):
    """conv -> BN -> activation, the order slim uses with normalizer_fn
    (SURVEY.md Appendix A item 3)."""
    w = weights[scope + '/weights']
    y = conv2d_transpose_same(x, w, stride) if transpose else conv2d_same(x, w, stride)
    y = batch_norm_infer(y, weights, scope)
    return F.relu(y) if relu else y


def residual_block(x, weights, scope, relu_first=True):
    """reference: code/autoencoder.py:274-287.  conv1 has ReLU unless the caller passed
    activation_fn=None (the *_final blocks: both convs linear); conv2 never has one."""
    y = conv_bn_act(x, weights, scope + '/conv1', 1, relu_first)
    y = conv_bn_act(y, weights, scope + '/conv2', 1, False)
    return y + x


# ----------------------------------------------------------------------------------
# heatmap + quantiser  (reference: code/autoencoder.py:171-200, code/quantizer.py:43-100)
# ----------------------------------------------------------------------------------

def heatmap3d(bottleneck):
    C = bottleneck.shape[1] - 1
    h2d = torch.sigmoid(bottleneck[:, 0:1]) * C                       # N1HW
    c = torch.arange(C, dtype=bottleneck.dtype).view(1, C, 1, 1)
    return torch.clamp(torch.clamp(h2d - c, max=1.0), min=0.0)       # max(min(h-c,1),0)


def quantize(z, centers, sigma=1.0):
    """-> (qsoft, qhard, symbols int64).  dist = square(abs(z - c)); phi_soft =
    softmax(-sigma dist); symbols = argmax softmax(-1e7 dist) (first index on ties);
    qhard = centers[symbols] (one_hot . centers)."""
    c = _t(centers, z.dtype)
    dist = torch.square(torch.abs(z.unsqueeze(-1) - c))               # (..., L)
    phi_soft = torch.softmax(-sigma * dist, dim=-1)
    phi_hard = torch.softmax(-HARD_SIGMA * dist, dim=-1)
    symbols = torch.argmax(phi_hard, dim=-1)
    qsoft = (phi_soft * c).sum(-1)
    onehot = F.one_hot(symbols, c.numel()).to(z.dtype)
    qhard = (onehot * c).sum(-1)
    return qsoft, qhard, symbols


# ----------------------------------------------------------------------------------
# autoencoder  (reference: code/autoencoder.py:218-268)
# ----------------------------------------------------------------------------------

ENC = 'autoencoder/encoder'
DEC = 'autoencoder/decoder'


def _res_stack(net, weights, root, kind, B):
    """kind = 'enc' | 'dec'.  reference: code/autoencoder.py:224-234 / :252-262."""
    final = 'res_block_enc_final' if kind == 'enc' else 'dec_after_res'
    res0 = net
    for b in range(B):
        res_b = net
        for i in (1, 2, 3):
            net = residual_block(
                net, weights, '{}/res_block_{}_{}/{}_{}_{}'.format(root, kind, b, kind, b, i))
        net = net + res_b
    net = residual_block(net, weights, '{}/{}'.format(root, final), relu_first=False)
    return net + res0


def encode(x, weights, cfg):
    """x: (N,3,H,W) float 0..255.  -> EncoderOutput.  cfg needs arch_param_B, heatmap,
    normalization.  reference: code/autoencoder.py:218-244."""
    net = normalize(x, cfg.get('normalization', 'FIXED'))
    net = conv_bn_act(net, weights, ENC + '/h1', 2, True)
    net = conv_bn_act(net, weights, ENC + '/h2', 2, True)
    net = _res_stack(net, weights, ENC, 'enc', cfg['arch_param_B'])
    net = conv_bn_act(net, weights, ENC + '/to_bn', 2, False)
    if cfg.get('heatmap', True):
        hm = heatmap3d(net)
        z = hm * net[:, 1:]
    else:
        hm, z = None, net
    qsoft, qhard, symbols = quantize(z, weights[ENC + '/centers'], sigma=1.0)
    qbar = qsoft + (qhard - qsoft)          # forward value of the STE expression (:133)
    return EncoderOutput(qbar, qhard, symbols, z, hm)


def decode(q, weights, cfg):
    """reference: code/autoencoder.py:246-268."""
    net = conv_bn_act(q, weights, DEC + '/from_bn', 2, True, transpose=True)
    net = _res_stack(net, weights, DEC, 'dec', cfg['arch_param_B'])
    net = conv_bn_act(net, weights, DEC + '/h12', 2, True, transpose=True)
    net = conv_bn_act(net, weights, DEC + '/h13', 2, False, transpose=True)
    net = denormalize(net, cfg.get('normalization', 'FIXED'))
    return torch.clamp(net, 0, 255)


# ----------------------------------------------------------------------------------
# context model  (reference: code/probclass.py:63-106,145-261,268-292)
# ----------------------------------------------------------------------------------

PC = 'probclass3d/logits'


def pc_masks(K=3):
    """first mask: D-slice K//2 zero at the centre row from the centre rightwards and all
    rows below; other mask keeps the centre.  reference: code/probclass.py:150-176.
    returns two (K//2+1, K, K) float32 arrays."""
    shape = (K // 2 + 1, K, K)
    first = np.ones(shape, np.float32)
    first[-1, K // 2, K // 2:] = 0
    first[-1, K // 2 + 1:, :] = 0
    other = np.ones(shape, np.float32)
    other[-1, K // 2, K // 2 + 1:] = 0
    other[-1, K // 2 + 1:, :] = 0
    return first, other


def pad_for_probclass3d(q, context_size, pad_value):
    """(N,C,H,W) -> (N,C+pad,H+2pad,W+2pad), constant pad, depth in FRONT only.
    reference: code/probclass.py:268-292."""
    pad = context_size // 2
    N, C, H, W = q.shape
    out = torch.full((N, C + pad, H + 2 * pad, W + 2 * pad), float(pad_value), dtype=q.dtype)
    out[:, pad:, pad:pad + H, pad:pad + W] = q
    return out


def _conv3d(x, weights, scope, mask, relu):
    """x: (N, Cin, D, H, W).  tf.nn.conv3d VALID stride 1 with weights*mask, bias,
    activation.  TF filter layout [d, h, w, in, out].  reference: code/probclass.py:227-261."""
    w = weights[scope + '/weights'] * mask[..., None, None]
    b = weights[scope + '/biases']
    wt = _t(w, x.dtype).permute(4, 3, 0, 1, 2).contiguous()
    y = F.conv3d(x, wt, _t(b, x.dtype))
    return F.relu(y) if relu else y


def pc_logits(q_pad, weights, K=3):
    """q_pad: (N, D, H, W) padded volume.  -> logits (N, D-4, H-8, W-8, L) for the
    res_shallow net (4 layers).  Final layer keeps conv3d's default ReLU
    (reference: code/probclass.py:214-221,233)."""
    first, other = pc_masks(K)
    x = q_pad.unsqueeze(1)                                               # N,1,D,H,W
    net = _conv3d(x, weights, PC + '/conv3d_conv0_mask', first, True)
    res_in = net
    net = _conv3d(net, weights, PC + '/res1/conv3d_conv1_mask', other, True)
    net = _conv3d(net, weights, PC + '/res1/conv3d_conv2_mask', other, False)
    net = net + res_in[:, :, 2:, 2:-2, 2:-2]                             # probclass.py:196
    net = _conv3d(net, weights, PC + '/conv3d_conv2_mask', other, True)
    return net.permute(0, 2, 3, 4, 1).contiguous()                       # N,C,h,w,L


def bitcost(q, symbols, weights, pad_value, K=3, num_layers=4):
    """-> (bits (N,C,h,w), logits (N,C,h,w,L)).  bits = softmax-cross-entropy * log2(e).
    reference: code/probclass.py:63-106."""
    ctx = num_layers * (K - 1) + 1                                       # probclass.py:47-52
    q_pad = pad_for_probclass3d(q, ctx, pad_value)
    logits = pc_logits(q_pad, weights, K)
    assert logits.shape[:4] == symbols.shape, (logits.shape, symbols.shape)
    logp = torch.log_softmax(logits, dim=-1)
    nll = -torch.gather(logp, -1, symbols.unsqueeze(-1)).squeeze(-1)
    return nll * float(np.log2(np.e)), logits


def bitcost_to_bpp(bits, x):
    """sum(bits) / (N*H*W).  reference: code/bits.py:4-20."""
    return float(bits.sum()) / (x.shape[0] * x.shape[2] * x.shape[3])


# ----------------------------------------------------------------------------------
# val.py wiring (reference: code/val.py:81-94)
# ----------------------------------------------------------------------------------

def validate_forward(x_uint8, weights, cfg, dtype=torch.float32):
    """x_uint8: (N,3,H,W) uint8 numpy.  Reproduces the val graph: encode, decode(qhard),
    bitcost(qbar, symbols, pad=centers[0]), bpp, uint8 truncation of the output."""
    x = _t(x_uint8, dtype)
    enc = encode(x, weights, cfg)
    x_out = decode(enc.qhard, weights, cfg)
    centers = weights[ENC + '/centers']
    bits, logits = bitcost(enc.qbar, enc.symbols, weights, pad_value=float(centers[0]))
    return {
        'enc': enc, 'x_out': x_out, 'bits': bits, 'logits': logits,
        'bpp': bitcost_to_bpp(bits, x),
        'x_out_uint8': x_out.to(torch.uint8),          # tf.cast truncates (val.py:91)
    }


def psnr_uint8(a, b):
    """PSNR on uint8 images, 10 log10(255^2 / MSE).  reference: code/val.py:227-232."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    mse = np.mean((a - b) ** 2)
    return float(10 * np.log10(255.0 ** 2 / mse))
