"""Convolutional autoencoder plugin -- MI355X mirror of the reference's code/autoencoder.py.

Same plugin surface (reference code/autoencoder.py:26-29, :33-86, :214-216):

    cls = get_network_cls(ae_config)            # keyed by config.arch ('CVPR')
    ae = cls(ae_config)
    enc = ae.encode(x, is_training=False)       # EncoderOutput(qbar, qhard, symbols, z, heatmap)
    x_out = ae.decode(enc.qhard, is_training=False)
    ae.get_centers_variable(); ae.get_subsampling_factor(); ae.encoder_variables(); ...

but on torch tensors that live on a HIP device, with every op executed by libimgcomp_hip.so
(hand-written gfx950 kernels behind the C ABI of include/imgcomp_hip.h).  TF's implicit variable
store is replaced by an explicit ``load_weights(dict)`` (names/layouts = the reference's checkpoint
variables).  There is no CPU path: tensors on the CPU raise.
"""
from collections import namedtuple, OrderedDict

import numpy as np
import torch

from . import _lib
from . import quantizer
from . import weights as _weights
from ._lib import lib, check, ptr

# reference: code/autoencoder.py:15,18
EncoderOutput = namedtuple('EncoderOutput', ['qbar', 'qhard', 'symbols', 'z', 'heatmap'])
_QuantizerOutput = namedtuple('_QuantizerOutput', ['qbar', 'qsoft', 'qhard', 'symbols'])

SCOPE_AE = 'autoencoder'
SCOPE_AE_ENC = SCOPE_AE + '/encoder'
SCOPE_AE_DEC = SCOPE_AE + '/decoder'

BN_EPSILON = 1e-5      # reference: code/autoencoder.py:118
BN_DECAY = 0.9         # reference: code/autoencoder.py:117
arch_param_n = 128     # reference: code/autoencoder.py:211


def get_network_cls(config):
    """reference: code/autoencoder.py:26-29."""
    return {
        'CVPR': _CVPR,
    }[config.arch]


def fold_batch_norm(gamma, beta, moving_mean, moving_variance, eps=BN_EPSILON):
    """Inference BatchNorm as y = x * scale + shift (float64 math, rounded once to fp32)."""
    g, b, m, v = (np.asarray(a, np.float64) for a in (gamma, beta, moving_mean, moving_variance))
    scale = g / np.sqrt(v + eps)
    shift = b - m * scale
    return scale.astype(np.float32), shift.astype(np.float32)


class _Network(object):
    def __init__(self, config, quantize=True):
        self.config = config
        self.quantize = quantize
        self.num_chan_bn_including_heatmap = config.num_chan_bn + 1
        self._train_graph = None   # training.TrainGraph.bind(): is_training=True calls run that graph's training-mode forward
        self._graph_version = -1
        self.plan_flags = 0        # IC_CONV3_* bits OR-ed into every call of THIS object (tests force a kernel form with it)
        self._centers = None       # set by load_weights(); access with get_centers_variable()
        self._params = None        # OrderedDict name -> device tensor (reference layouts)
        self._device = None

    # -- plugin surface -----------------------------------------------------------------------

    @staticmethod
    def get_subsampling_factor():
        raise NotImplementedError()

    def encode(self, x, is_training, plan_flags=0):
        """x: (N,3,H,W) float32 in 0..255 on the HIP device.  -> EncoderOutput.
        plan_flags (not in the reference): per-call IC_CONV3_* launch-plan bits for the 3x3 layers, see _lib.CONV3_*."""
        assert x.dtype == torch.float32, 'Expected float32 for x, got {}'.format(x.dtype)
        if is_training:
            return self._training_graph().plugin_encode(x)
        self._sync_with_training_graph()
        self._require_weights()
        return self._encode(x, is_training, int(plan_flags) | int(self.plan_flags))

    def decode(self, q, is_training, plan_flags=0):
        if is_training:
            return self._training_graph().plugin_decode(q)
        self._sync_with_training_graph()
        self._require_weights()
        return self._decode(q, is_training, int(plan_flags) | int(self.plan_flags))

    def _training_graph(self):
        if self._train_graph is None:
            raise ValueError('is_training=True needs the training graph that owns the variables and the backward: '
                             'training.TrainGraph(ae_config, pc_config, weights).bind(ae, pc) (see train.py)')
        return self._train_graph

    def _sync_with_training_graph(self):
        """test-in-train (train.py:115-127): is_training=False on a bound object evaluates the CURRENT training variables --
        moving averages folded into the convs, filters re-packed -- whenever an optimiser step has changed them"""
        g = self._train_graph
        if g is not None and self._graph_version != g.version:
            self.load_weights(OrderedDict((n, t.detach().cpu().numpy()) for n, t in g.params.items()), g.dev)
            self._graph_version = g.version

    def get_centers_variable(self):
        if self._train_graph is not None:
            return self._train_graph.params[SCOPE_AE_ENC + '/centers']          # the live training variable
        if self._centers is None:
            raise ValueError('Call load_weights(...) before trying to access centers')
        return self._centers

    def encoder_variables(self):
        """Trainable encoder variables, centres included (reference :70-73)."""
        return self._trainable(SCOPE_AE_ENC)

    def decoder_variables(self):
        return self._trainable(SCOPE_AE_DEC)

    def encoder_regularization_loss(self):
        """factor * l2_loss(conv weights) + centres term (reference :79-82, quantizer.py:18-24)."""
        return self._reg_loss(SCOPE_AE_ENC)

    def decoder_regularization_loss(self):
        return self._reg_loss(SCOPE_AE_DEC)

    # -- weights ------------------------------------------------------------------------------

    def load_weights(self, weights, device='cuda'):
        """weights: dict name -> array in the reference's TF layouts (checkpoint variable names)."""
        device = torch.device(device)
        if device.type != 'cuda':
            raise _lib.HipLibraryError('the autoencoder runs only on a HIP device, got {}'.format(device))
        self._device = device
        self._params = OrderedDict()
        for name, arr in weights.items():
            if name.startswith(SCOPE_AE + '/'):
                self._params[name] = torch.as_tensor(np.ascontiguousarray(arr), dtype=torch.float32).to(device)
        self._centers = self._params[SCOPE_AE_ENC + '/centers']
        self._prepare(weights)
        return self

    def init_weights(self, pc_config=None, seed=1234, device='cuda'):
        raise NotImplementedError('use weights.synthetic_weights(ae_config, pc_config) + load_weights')

    def _require_weights(self):
        if self._params is None:
            raise ValueError('no weights: call load_weights(dict) first (replaces TF variable init / Saver.restore)')

    def _trainable(self, scope):
        self._require_weights()
        return [t for n, t in self._params.items()
                if n.startswith(scope + '/') and 'moving_' not in n]

    def _reg_loss(self, scope):
        """value of the L2 terms of a scope.  On an object bound to a training graph the live variables are read; their
        gradients do not come from this value (they are folded into the filter-gradient kernels, training.py)."""
        if self._train_graph is None:
            self._require_weights()
        params = self._train_graph.params if self._train_graph is not None else self._params
        device = self._train_graph.dev if self._train_graph is not None else self._device
        centers = params[SCOPE_AE_ENC + '/centers']
        total = torch.zeros((), dtype=torch.float32, device=device)
        f = float(self.config.regularization_factor)
        for n, t in params.items():
            if n.startswith(scope + '/') and n.endswith('/weights'):
                total = total + f * 0.5 * (t * t).sum()
        if scope == SCOPE_AE_ENC and self.config.regularization_factor_centers != 0:
            total = total + quantizer.create_centers_regularization_term(self.config, centers)
        return total.detach()

    def _prepare(self, weights):
        raise NotImplementedError()

    def _encode(self, x, is_training, plan_flags=0):
        raise NotImplementedError()

    def _decode(self, q, is_training, plan_flags=0):
        raise NotImplementedError()


class _CVPR(_Network):
    """reference: code/autoencoder.py:214-268."""

    @staticmethod
    def get_subsampling_factor():
        return 8

    # -- device-side plan: packed filters, folded BN, pointer tables -------------------------------

    def _prepare(self, weights):
        cfg = self.config
        self._B = int(cfg.arch_param_B)
        self._C = int(cfg.num_chan_bn)
        self._L = int(cfg.num_centers)
        dev = self._device
        st = _lib.current_stream(dev)
        specs = _weights.ae_conv_specs(self._C, self._B, bool(cfg.heatmap))
        self._plan = {}           # scope -> (w_dev, scale_dev, shift_dev); keeps tensors alive
        self._edge_both = False
        enc_tab, dec_tab = [], []
        packed_n = lib.ic_conv3x3_c128_both_packed_floats()
        for scope, kind, shape in specs:
            w = self._params[scope + '/weights']
            scale, shift = fold_batch_norm(*(weights[scope + '/BatchNorm/' + k] for k in
                                             ('gamma', 'beta', 'moving_mean', 'moving_variance')))
            scale_d = torch.from_numpy(scale).to(dev)
            shift_d = torch.from_numpy(shift).to(dev)
            kh, kw, a_, b_ = shape
            cin, cout = (a_, b_) if kind == 'conv' else (b_, a_)
            stride = 1 if (kh, kw) == (3, 3) and kind == 'conv' else 2
            n_mfma = lib.ic_conv2d_mfma_packed_floats(kh, kw, cin, cout, stride, int(kind == 'deconv'))
            if kind == 'conv' and tuple(shape) == (3, 3, arch_param_n, arch_param_n):
                wp = torch.empty(packed_n, dtype=torch.float32, device=dev)
                # direct-form and Winograd fragments side by side; the library picks the form per launch
                check(lib.ic_pack_conv3x3_c128_both_f32(ptr(w), ptr(wp), 0, st), 'ic_pack_conv3x3_c128_both_f32')
                w_use = wp
            elif scope.endswith(('/h2', '/h12')) and (kh, kw, stride) == (5, 5, 2) and {cin, cout} == {64, 128}:
                # h2 / h12: MFMA fragments and the F(4x4)-over-phases fragments in one blob; the library picks per call
                # (IC_CONV5_BOTH_PACKED tells it the blob has both, see _encode / _decode)
                tr = int(kind == 'deconv')
                wp = torch.empty(lib.ic_conv5s2_both_packed_floats(tr), dtype=torch.float32, device=dev)
                check(lib.ic_pack_conv5s2_both_f32(ptr(w), ptr(wp), tr, st), 'ic_pack_conv5s2_both_f32')
                w_use = wp
                self._edge_both = True
            elif n_mfma and not scope.endswith('/h1'):
                # h2, to_bn, h12: matrix-core path, filter in MFMA fragment order
                wp = torch.empty(n_mfma, dtype=torch.float32, device=dev)
                check(lib.ic_pack_conv2d_mfma_f32(ptr(w), ptr(wp), kh, kw, cin, cout, stride, int(kind == 'deconv'), st),
                      'ic_pack_conv2d_mfma_f32')
                w_use = wp
            else:
                w_use = w
            self._plan[scope] = (w_use, scale_d, shift_d)
            (enc_tab if scope.startswith(SCOPE_AE_ENC) else dec_tab).extend([w_use, scale_d, shift_d])
        enc_tab.append(self._centers)
        torch.cuda.synchronize(dev)
        self._enc_tab = _lib.ptr_table(enc_tab)
        self._dec_tab = _lib.ptr_table(dec_tab)
        self._ws = None

    def _abi_flags(self, plan_flags):
        """the caller's per-call plan bits + what this object's filter blobs hold"""
        return int(plan_flags) | (_lib.CONV5_BOTH_PACKED if self._edge_both else 0)

    def sharing_weights(self):
        """a second object of this network over the SAME device weights, packed filters and pointer tables, with its own workspace
        and per-call state -- for a caller that keeps several images in flight, one object per stream (val.py), without uploading
        and packing the weights once per stream."""
        import copy
        other = copy.copy(self)
        other._ws = None
        other._last_qsoft = None
        return other

    def _workspace(self, N, H, W):
        need = lib.ic_ae_workspace_bytes(N, H, W, self._C)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self._device)
        return self._ws, need

    # -- forward -----------------------------------------------------------------------------------

    def _encode(self, x, is_training, plan_flags=0):
        assert not is_training          # (training mode is dispatched to the bound TrainGraph in encode())
        _lib.require_cuda(x, 'x')
        x = x.contiguous()
        N, three, H, W = x.shape
        assert three == 3, 'Expected N3HW, got {}'.format(tuple(x.shape))
        f = self.get_subsampling_factor()
        if H % f or W % f:
            raise ValueError('H and W must be multiples of {} (val.py pads images first), got {}x{}'.format(f, H, W))
        C, hh, ww = self._C, H // f, W // f
        mk = lambda: torch.empty((N, C, hh, ww), dtype=torch.float32, device=x.device)
        heat_on = bool(self.config.heatmap)
        if not self.quantize:
            # autoencoder.py:127-129: no quantiser -- the encoder hands on the (masked) bottleneck itself, _QuantizerOutput(z, None, None, None)
            z = mk()
            heatmap = mk() if heat_on else None
            ws, need = self._workspace(N, H, W)
            check(lib.ic_ae_encode_f32(ptr(x), self._enc_tab, self._B, C, self._L, int(heat_on),
                                       int(self.config.normalization == 'FIXED'), ptr(heatmap), ptr(z), None, None, None, None,
                                       N, H, W, ptr(ws), need, self._abi_flags(plan_flags), _lib.current_stream(x.device)), 'ic_ae_encode_f32')
            return EncoderOutput(z, None, None, z, heatmap)
        z, qsoft, qhard = mk(), mk(), mk()
        qbar = mk() if heat_on else None
        heatmap = mk() if heat_on else None
        symbols = torch.empty((N, C, hh, ww), dtype=torch.int64, device=x.device)
        ws, need = self._workspace(N, H, W)
        check(lib.ic_ae_encode_f32(ptr(x), self._enc_tab, self._B, C, self._L, int(heat_on),
                                   int(self.config.normalization == 'FIXED'),
                                   ptr(heatmap), ptr(z), ptr(qsoft), ptr(qhard), ptr(qbar), ptr(symbols),
                                   N, H, W, ptr(ws), need, self._abi_flags(plan_flags), _lib.current_stream(x.device)), 'ic_ae_encode_f32')
        if qbar is None:
            qbar = qhard      # forward value of qsoft + stop_gradient(qhard - qsoft)
        self._last_qsoft = qsoft
        return EncoderOutput(qbar, qhard, symbols, z, heatmap)

    def _decode(self, q, is_training, plan_flags=0):
        assert not is_training
        _lib.require_cuda(q, 'q')
        q = q.contiguous()
        N, C, hh, ww = q.shape
        assert C == self._C, 'Expected {} bottleneck channels, got {}'.format(self._C, C)
        f = self.get_subsampling_factor()
        H, W = hh * f, ww * f
        x_out = torch.empty((N, 3, H, W), dtype=torch.float32, device=q.device)
        ws, need = self._workspace(N, H, W)
        check(lib.ic_ae_decode_f32(ptr(q), self._dec_tab, self._B, C, int(self.config.normalization == 'FIXED'),
                                   ptr(x_out), N, H, W, ptr(ws), need, self._abi_flags(plan_flags), _lib.current_stream(q.device)),
              'ic_ae_decode_f32')
        return x_out
