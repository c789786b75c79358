"""Variable inventory + seeded synthetic initialisation for the CVPR autoencoder and the
res_shallow context model.

Names and shapes follow the reference's TF variable scopes (SURVEY.md Appendix B;
reference code/autoencoder.py:21-23,222-265,280-282, code/quantizer.py:13-14,
code/probclass.py:28,189,249-257), arrays are kept in the reference's TF layouts:

  conv2d            weights [kh, kw, cin, cout]
  conv2d_transpose  weights [kh, kw, cout, cin]
  conv3d            weights [kd, kh, kw, cin, cout] (stored UNMASKED), biases [cout]
  BatchNorm         gamma, beta, moving_mean, moving_variance  [cout]
  centers           [L]

There is no network access for the published checkpoints, so benchmarks and tests run on
weights drawn as SURVEY.md section 8(d) prescribes (RandomState(1234): Xavier-uniform by
TF's fan rule, non-trivial BN statistics, small pc biases, centres linspace(-2, 2, L)).
"""
from collections import OrderedDict

import numpy as np

ENC = 'autoencoder/encoder'
DEC = 'autoencoder/decoder'
PC = 'probclass3d/logits'
N_FILTERS = 128           # reference: code/autoencoder.py:211 (arch_param_n)


def ae_conv_specs(num_chan_bn, B=5, heatmap=True):
    """-> ordered list of (scope, kind, weight_shape); kind in {'conv', 'deconv'}."""
    n = N_FILTERS
    C = num_chan_bn + (1 if heatmap else 0)
    specs = [(ENC + '/h1', 'conv', (5, 5, 3, n // 2)),
             (ENC + '/h2', 'conv', (5, 5, n // 2, n))]
    for b in range(B):
        for i in (1, 2, 3):
            for c in (1, 2):
                specs.append(('{}/res_block_enc_{}/enc_{}_{}/conv{}'.format(ENC, b, b, i, c),
                              'conv', (3, 3, n, n)))
    for c in (1, 2):
        specs.append(('{}/res_block_enc_final/conv{}'.format(ENC, c), 'conv', (3, 3, n, n)))
    specs.append((ENC + '/to_bn', 'conv', (5, 5, n, C)))
    specs.append((DEC + '/from_bn', 'deconv', (3, 3, n, num_chan_bn)))
    for b in range(B):
        for i in (1, 2, 3):
            for c in (1, 2):
                specs.append(('{}/res_block_dec_{}/dec_{}_{}/conv{}'.format(DEC, b, b, i, c),
                              'conv', (3, 3, n, n)))
    for c in (1, 2):
        specs.append(('{}/dec_after_res/conv{}'.format(DEC, c), 'conv', (3, 3, n, n)))
    specs.append((DEC + '/h12', 'deconv', (5, 5, n // 2, n)))
    specs.append((DEC + '/h13', 'deconv', (5, 5, 3, n // 2)))
    return specs


def pc_conv_specs(L, k=24, K=3):
    fs = (K // 2 + 1, K, K)
    return [(PC + '/conv3d_conv0_mask', fs + (1, k)),
            (PC + '/res1/conv3d_conv1_mask', fs + (k, k)),
            (PC + '/res1/conv3d_conv2_mask', fs + (k, k)),
            (PC + '/conv3d_conv2_mask', fs + (k, L))]


def _xavier_uniform(rs, shape, fan_in, fan_out):
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rs.uniform(-limit, limit, size=shape).astype(np.float32)


def synthetic_weights(ae_config, pc_config, seed=1234, centers='linspace', gain=0.4, heatmap_bias=2.5):
    """Seeded random-init weights for (ae_config, pc_config) -> OrderedDict name -> ndarray.

    `centers`: 'linspace' -> linspace(-2, 2, L); 'uniform' -> sorted U(-2, 2) drawn from
    RandomState(666) (the seed the reference's initializer uses, quantizer.py:31).
    `gain` scales the Xavier conv weights (0.4 keeps the 64-conv residual stacks at O(1)
    activations so z spreads over all centres instead of saturating); `heatmap_bias`
    is written to to_bn's BN beta of the heatmap channel so that about half of the
    bottleneck channels are unmasked.  Both only shape the *synthetic* statistics.
    """
    rs = np.random.RandomState(seed)
    w = OrderedDict()
    L = int(ae_config.num_centers)
    lo, hi = map(float, ae_config.centers_initial_range)
    if centers == 'linspace':
        w[ENC + '/centers'] = np.linspace(lo, hi, L).astype(np.float32)
    else:
        w[ENC + '/centers'] = np.sort(
            np.random.RandomState(666).uniform(lo, hi, L)).astype(np.float32)
    for scope, kind, shape in ae_conv_specs(int(ae_config.num_chan_bn), int(ae_config.arch_param_B),
                                            bool(ae_config.heatmap)):
        kh, kw, a, b = shape
        cin, cout = (a, b) if kind == 'conv' else (b, a)
        w[scope + '/weights'] = _xavier_uniform(rs, shape, kh * kw * cin, kh * kw * cout) * np.float32(gain)
        w[scope + '/BatchNorm/gamma'] = rs.uniform(0.5, 1.5, cout).astype(np.float32)
        w[scope + '/BatchNorm/beta'] = rs.normal(0, 0.1, cout).astype(np.float32)
        w[scope + '/BatchNorm/moving_mean'] = rs.normal(0, 0.1, cout).astype(np.float32)
        w[scope + '/BatchNorm/moving_variance'] = rs.uniform(0.5, 1.5, cout).astype(np.float32)
    if bool(ae_config.heatmap) and heatmap_bias is not None:
        w[ENC + '/to_bn/BatchNorm/beta'][0] = np.float32(heatmap_bias)
    k = int(pc_config.arch_param__k)
    K = int(pc_config.kernel_size)
    for scope, shape in pc_conv_specs(L, k, K):
        kd, kh, kw, cin, cout = shape
        w[scope + '/weights'] = _xavier_uniform(rs, shape, kd * kh * kw * cin, kd * kh * kw * cout)
        w[scope + '/biases'] = rs.normal(0, 0.01, cout).astype(np.float32)
    return w


def num_parameters(weights, trainable_only=True):
    n = 0
    for name, a in weights.items():
        if trainable_only and ('moving_mean' in name or 'moving_variance' in name):
            continue
        n += a.size
    return n


def synthetic_image(shape, kind='natural', seed=0):
    """Seeded uint8 test images (N,3,H,W).  'noise': uniform uint8; 'natural': a sum of 8
    random low-frequency 2-D cosines per channel plus N(0, 8) noise."""
    N, C, H, W = shape
    rs = np.random.RandomState(seed)
    if kind == 'noise':
        return rs.randint(0, 256, size=shape).astype(np.uint8)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing='ij')
    img = np.zeros(shape, np.float64)
    for n in range(N):
        for c in range(C):
            acc = np.full((H, W), 128.0)
            for _ in range(8):
                fy, fx = rs.uniform(0, 6.0, 2) * 2 * np.pi / max(H, W) * 4
                ph = rs.uniform(0, 2 * np.pi)
                amp = rs.uniform(8, 40)
                acc += amp * np.cos(fy * yy + fx * xx + ph)
            img[n, c] = acc + rs.normal(0, 8.0, (H, W))
    return np.clip(np.round(img), 0, 255).astype(np.uint8)
