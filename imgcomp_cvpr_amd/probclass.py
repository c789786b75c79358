"""Context-model ("probability classifier") plugin -- mirror of the reference's code/probclass.py.

    cls = get_network_cls(pc_config)                 # keyed by pc_config.arch ('res_shallow')
    pc = cls(pc_config, num_centers=L)
    bits = pc.bitcost(q, target_symbols, is_training=False, pad_value=pc.auto_pad_value(ae))   # NCHW
    logits = pc.logits(q_padded, is_training=False)                                            # N,C,h,w,L

All positions are evaluated in parallel by libimgcomp_hip.so (csrc/probclass.hip); the volume is
padded on load, never materialised.  Weights come from ``load_weights(dict)`` in the reference's
variable names/layouts (unmasked conv3d filters [2,3,3,cin,cout] + biases).
"""
import itertools
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from . import weights as _weights
from ._lib import lib, check, ptr


def get_network_cls(pc_config):
    """reference: code/probclass.py:11-15."""
    return {
        'res_shallow': _ResShallow,
    }[pc_config.arch]


def context_shape_from_context_size(context_size):
    """:return context shape as DHW (reference code/probclass.py:18-20)."""
    return context_size // 2 + 1, context_size, context_size


def context_size_from_context_shape(context_shape):
    return context_shape[-1]


class _Network3D(object):
    _PROBCLASS_SCOPE = 'probclass3d'

    def __init__(self, pc_config, num_centers):
        self.config = pc_config
        self.L = int(num_centers)
        self._params = None
        self._device = None
        self._ws = None
        self._last_logits = None
        self._train_graph = None      # training.TrainGraph.bind()
        self._graph_version = -1

    # -- static geometry (reference :40-57) --

    @classmethod
    def get_num_layers(cls):
        raise NotImplementedError()

    @classmethod
    def get_context_size(cls, config):
        """width / height of the receptive field."""
        return cls.get_num_layers() * (config.kernel_size - 1) + 1

    @classmethod
    def get_context_shape(cls, config):
        """Shape as DHW."""
        return context_shape_from_context_size(cls.get_context_size(config))

    @property
    def filter_shape(self):
        K = self.config.kernel_size
        return K // 2 + 1, K, K

    def auto_pad_value(self, ae):
        """0, or centers[0] when use_centers_for_padding (reference :59-61)."""
        return 0 if not self.config.use_centers_for_padding else ae.get_centers_variable()[0]

    # -- masks, as numpy (reference :150-176); the kernels skip the zeroed taps --

    def create_first_mask(self):
        K = self.config.kernel_size
        mask = np.ones(self.filter_shape, dtype=np.float32)
        mask[-1, K // 2, K // 2:] = 0
        mask[-1, K // 2 + 1:, :] = 0
        return mask[..., None, None]

    def create_other_mask(self):
        K = self.config.kernel_size
        mask = np.ones(self.filter_shape, dtype=np.float32)
        mask[-1, K // 2, K // 2 + 1:] = 0
        mask[-1, K // 2 + 1:, :] = 0
        return mask[..., None, None]

    # -- weights --

    def load_weights(self, weights, device='cuda'):
        device = torch.device(device)
        if device.type != 'cuda':
            raise _lib.HipLibraryError('the context model runs only on a HIP device, got {}'.format(device))
        self._device = device
        self._params = OrderedDict()
        for name, arr in weights.items():
            if name.startswith(self._PROBCLASS_SCOPE + '/'):
                self._params[name] = torch.as_tensor(np.ascontiguousarray(arr), dtype=torch.float32).to(device)
        self._prepare()
        return self

    def variables(self):
        self._require_weights()
        return list(self._params.values())

    def get_network_variables(self):
        return self.variables()

    def regularization_loss(self):
        """None unless config.regularization_factor is set (reference :115-119)."""
        if self.config.regularization_factor is None:
            return None
        if self._train_graph is not None:
            params = {n: t for n, t in self._train_graph.params.items() if n.startswith(self._PROBCLASS_SCOPE + '/')}
        else:
            self._require_weights()
            params = self._params
        f = float(self.config.regularization_factor)
        return (f * sum(0.5 * (t * t).sum() for n, t in params.items() if n.endswith('/weights'))).detach()

    def _require_weights(self):
        if self._params is None:
            raise ValueError('no weights: call load_weights(dict) first')

    def _prepare(self):
        raise NotImplementedError()

    # -- forward --

    def _pad_value_as_float(self, pad_value):
        """The C ABI takes the pad value by value.  A tensor (centers[0], reference probclass.py:59-61) is fetched from the
        device once per distinct VALUE SOURCE: a .item() per image would stall the host until everything queued on the
        stream -- the whole encoder -- has finished, before the decoder could be enqueued.
          * bound to a training graph (test-in-train, train.py:122-126): the optimiser updates the centres through a raw
            pointer, which neither moves the storage nor bumps torch's version counter, so the cache key is the graph's
            own `version` (bumped by every apply_gradients) and the value is the graph's pinned copy of centres[0];
          * unbound: keyed on the identity of the tensor that OWNS the storage (a weak reference: a fresh centres tensor at
            a recycled address is a different owner) plus its version counter."""
        if not torch.is_tensor(pad_value):
            return float(pad_value)
        g = self._train_graph
        if g is not None:
            key = ('graph', id(g), g.version)
            if getattr(self, '_pad_cache', (None, None))[0] != key:
                self._pad_cache = (key, float(g._pad_value()))
            return self._pad_cache[1]
        owner = pad_value._base if pad_value._base is not None else pad_value
        cached = getattr(self, '_pad_cache', None)
        if (cached is not None and cached[0][0] == 'tensor' and cached[0][1]() is owner
                and cached[0][2:] == (pad_value.data_ptr(), owner._version)):
            return cached[1]
        import weakref
        self._pad_cache = (('tensor', weakref.ref(owner), pad_value.data_ptr(), owner._version), float(pad_value.item()))
        return self._pad_cache[1]

    def bitcost(self, q, target_symbols, is_training, pad_value=0):
        """q: NCHW float32, target_symbols: NCHW int64 -> bit cost per symbol, NCHW
        (reference code/probclass.py:63-106)."""
        raise NotImplementedError()

    def logits(self, q, is_training):
        """q: ALREADY padded volume (N,D,H,W) [the reference passes N,D,H,W,1] -> (N,D-4,H-8,W-8,L)
        (reference code/probclass.py:130-135)."""
        raise NotImplementedError()


class _ResShallow(_Network3D):
    """conv0 -> residual(conv1, conv2) -> conv2(final), reference code/probclass.py:199-221."""
    _NUM_RESIDUAL = 1

    @classmethod
    def get_num_layers(cls):
        return 2 + _ResShallow._NUM_RESIDUAL * 2

    def _prepare(self):
        if self.config.kernel_size != 3:
            raise NotImplementedError('the HIP context model implements kernel_size = 3 (res_shallow configs)')
        if self.config.learn_pad_var:
            raise NotImplementedError('learn_pad_var=True is dead code in the reference configs')
        self._k = int(self.config.arch_param__k)
        tabs = []
        for scope, shape in _weights.pc_conv_specs(self.L, self._k, int(self.config.kernel_size)):
            w, b = self._params[scope + '/weights'], self._params[scope + '/biases']
            assert tuple(w.shape) == tuple(shape), (scope, tuple(w.shape), shape)
            tabs += [w, b]
        # the matrix-core layers' filter fragments, packed once: inference runs without a per-call packing launch
        n = lib.ic_pc_packed_floats(self._k, self.L)
        packed = None
        if n:
            packed = torch.empty(n, dtype=torch.float32, device=self._device)
            check(lib.ic_pc_pack_filters_f32(_lib.ptr_table(tabs + [None]), self._k, self.L, ptr(packed), _lib.current_stream(self._device)),
                  'ic_pc_pack_filters_f32')
            torch.cuda.synchronize(self._device)
        self._tab_tensors = tabs + [packed]
        self._tab = _lib.ptr_table(self._tab_tensors)

    def sharing_weights(self):
        """a second object of this network over the SAME device weights and packed filters with its own workspace and caches: one
        per stream for a caller that keeps several images in flight (val.py)."""
        import copy
        other = copy.copy(self)
        other._ws = None
        other._last_logits = None
        if hasattr(other, '_pad_cache'):
            del other._pad_cache
        return other

    def _workspace(self, N, C, h, w):
        need = lib.ic_pc_workspace_bytes(N, C, h, w, self._k)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self._device)
        return self._ws, need

    def _sync_with_training_graph(self):
        g = self._train_graph
        if g is not None and self._graph_version != g.version:
            self.load_weights(OrderedDict((n, t.detach().cpu().numpy()) for n, t in g.params.items()
                                          if n.startswith(self._PROBCLASS_SCOPE + '/')), g.dev)
            self._graph_version = g.version

    def bitcost(self, q, target_symbols, is_training, pad_value=0, return_logits=False):
        if is_training:
            if self._train_graph is None:
                raise ValueError('is_training=True needs the training graph that owns the variables and the backward: '
                                 'training.TrainGraph(ae_config, pc_config, weights).bind(ae, pc) (see train.py)')
            assert not return_logits
            return self._train_graph.plugin_bitcost(q, target_symbols, pad_value)
        self._sync_with_training_graph()
        self._require_weights()
        assert q.dim() == 4, 'Expected NCHW'
        _lib.require_cuda(q, 'q')
        assert target_symbols.dtype == torch.int64 and target_symbols.shape == q.shape
        q = q.contiguous()
        target_symbols = target_symbols.contiguous()
        N, C, h, w = q.shape
        bits = torch.empty_like(q)
        logits = torch.empty((N, C, h, w, self.L), dtype=torch.float32, device=q.device) if return_logits else None
        ws, need = self._workspace(N, C, h, w)
        check(lib.ic_pc_bitcost_f32(ptr(q), ptr(target_symbols), self._tab, self._k, self.L,
                                    self._pad_value_as_float(pad_value), ptr(logits), ptr(bits),
                                    N, C, h, w, ptr(ws), need, _lib.current_stream(q.device)), 'ic_pc_bitcost_f32')
        if return_logits:
            return bits, logits
        return bits

    def logits_unpadded(self, q, pad_value):
        """logits (N,C,h,w,L) for an un-padded q, padding on load with pad_value."""
        self._require_weights()
        _lib.require_cuda(q, 'q')
        q = q.contiguous()
        N, C, h, w = q.shape
        out = torch.empty((N, C, h, w, self.L), dtype=torch.float32, device=q.device)
        ws, need = self._workspace(N, C, h, w)
        check(lib.ic_pc_logits_f32(ptr(q), self._tab, self._k, self.L, self._pad_value_as_float(pad_value),
                                   ptr(out), N, C, h, w, ptr(ws), need, _lib.current_stream(q.device)),
              'ic_pc_logits_f32')
        return out

    def logits(self, q, is_training):
        if is_training:
            raise NotImplementedError('is_training=True: the context model trains inside imgcomp_cvpr_amd.training.TrainGraph '
                                      '(forward with tape + hand-written backward); this plugin method is inference only')
        self._require_weights()
        _lib.require_cuda(q, 'q')
        if q.dim() == 5:
            assert q.shape[-1] == 1, 'Expected NDHW1'
            q = q[..., 0]
        q = q.contiguous()
        N, D, H, W = q.shape
        out = torch.empty((N, D - 4, H - 8, W - 8, self.L), dtype=torch.float32, device=q.device)
        ws, need = self._workspace(N, D - 4, H - 8, W - 8)
        check(lib.ic_pc_logits_padded_f32(ptr(q), self._tab, self._k, self.L, ptr(out), N, D, H, W,
                                          ptr(ws), need, _lib.current_stream(q.device)), 'ic_pc_logits_padded_f32')
        return out


################################################################################
# Helpers for arithmetic coding (val.py --real_bpp), reference code/probclass.py:393-482
################################################################################


class ProbclassNetworkTesting(object):
    """bit cost of a whole symbol volume, fully convolutionally (reference code/probclass.py:393-421)."""

    def __init__(self, pc, ae, sess=None):
        self.pc = pc
        self.centers = ae.get_centers_variable()
        self.pad_value = pc.auto_pad_value(ae)

    def get_total_bit_cost(self, symbols):
        """:param symbols: CHW or NCHW numpy / tensor of all symbols of an image -> total bits (float)."""
        sym = symbols if torch.is_tensor(symbols) else torch.as_tensor(np.asarray(symbols))
        if sym.dim() == 3:
            sym = sym[None]
        assert sym.dim() == 4
        sym = sym.to(self.centers.device).long().contiguous()
        q = self.centers[sym]                                    # tf.gather(centers, symbols)
        bc = self.pc.bitcost(q, sym, is_training=False, pad_value=self.pad_value)
        return float(bc.double().sum())


class PredictionNetwork(object):
    """Frequency tables for the arithmetic coder (reference code/probclass.py:425-482).

    get_pr / get_freqs keep the reference's one-context-at-a-time interface (the decoder needs it: a symbol's
    table depends on the symbols decoded before it).  get_all(...) is the parallel path the README asks for
    (README.md:71): ONE pass of the context model yields the tables of every position, and because the kernels
    evaluate a fixed, position-independent fp32 expression per logit, both paths give bit-identical tables."""

    def __init__(self, pc, config, centers, sess=None, freqs_resolution=1e9):
        self.pc = pc
        self.pc_class = pc.__class__
        self.config = config
        self.centers = centers
        self.input_ctx_shape = self.pc_class.get_context_shape(config)
        self.freqs_resolution = float(freqs_resolution)

    def pad_symbols_volume(self, symbols):
        assert symbols.ndim == 3
        return pad_for_probclass3d(symbols, self.pc_class.get_context_size(self.config))

    def undo_pad_symbols_volume(self, symbols):
        assert symbols.ndim == 3
        return undo_pad_for_probclass3d(symbols, self.pc_class.get_context_size(self.config))

    def _tables(self, vol_padded):
        """vol_padded: (D,H,W) int symbols (already padded) -> (pr, freqs) for every context, raster C,H,W."""
        dev = self.centers.device
        sym = torch.as_tensor(np.ascontiguousarray(vol_padded)).to(dev).long()
        q = self.centers[sym][None].contiguous()                       # gather -> (1,D,H,W) float
        logits = self.pc.logits(q, is_training=False)                  # (1,D-4,H-8,W-8,L)
        n = logits.numel() // self.pc.L
        freqs = torch.empty((n, self.pc.L), dtype=torch.int64, device=dev)
        pr = torch.empty((n, self.pc.L), dtype=torch.float32, device=dev)
        check(lib.ic_pc_logits_to_freqs_f32(ptr(logits), n, self.pc.L, self.freqs_resolution, ptr(freqs), ptr(pr),
                                            _lib.current_stream(dev)), 'ic_pc_logits_to_freqs_f32')
        return pr.cpu().numpy(), freqs.cpu().numpy()

    def get_all(self, symbols_padded):
        """(pr, freqs), each (num_contexts, L), contexts in the order of iter_over_blocks."""
        return self._tables(symbols_padded)

    def decode_stream(self, stream_bytes, symbols_shape, first_sym, flags=0):
        """Row N3: the whole sequential decode on the device (ic_pc_decode_f32) -- per symbol the same context-model
        kernels as get_freqs, the table, the arithmetic-decoder step and the gather of the next context are enqueued
        back to back without a host round trip.  stream_bytes: what the encoder wrote; symbols_shape: un-padded
        (C,h,w); first_sym: the uncoded first symbol.  -> (C,h,w) int64 numpy."""
        C, h, w = (int(v) for v in symbols_shape)
        dev = self.centers.device
        data = torch.frombuffer(bytearray(stream_bytes) or bytearray(1), dtype=torch.uint8).to(dev)
        out = torch.empty((C, h, w), dtype=torch.int64, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        need = lib.ic_pc_decode_workspace_bytes(C, h, w, self.pc._k)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        centers = self.centers.contiguous().float()
        check(lib.ic_pc_decode_f32(ptr(data), len(stream_bytes), int(first_sym), self.pc._tab, ptr(centers), self.pc._k,
                                   self.pc.L, self.freqs_resolution, ptr(out), ptr(status), C, h, w, ptr(ws), need,
                                   int(flags), _lib.current_stream(dev)), 'ic_pc_decode_f32')
        if int(status.item()) != 0:
            raise ValueError('Cannot decode symbol because total is too large')
        return out.cpu().numpy()

    def get_pr(self, input_ctx):
        """:param input_ctx: symbols of ONE context, CHW = input_ctx_shape -> (L,) float32."""
        assert tuple(input_ctx.shape) == tuple(self.input_ctx_shape), '{} != {}'.format(
            input_ctx.shape, self.input_ctx_shape)
        return self._tables(input_ctx)[0][0]

    def get_freqs(self, input_ctx):
        """:param input_ctx: symbols of ONE context, CHW -> (L,) int64, all > 0."""
        assert tuple(input_ctx.shape) == tuple(self.input_ctx_shape), '{} != {}'.format(
            input_ctx.shape, self.input_ctx_shape)
        f = self._tables(input_ctx)[1][0]
        assert np.all(f > 0), 'We do not want zero frequencies!: {}'.format(f)
        return f


# -- host-side helpers of the reference's NumPy branch (reference code/probclass.py:268-292,341-351,367-387) --

def pad_for_probclass3d(x, context_size, pad_value=0, learn_pad_var=False):
    """numpy CHW / NCHW or torch NCHW: constant-pad depth (front only), H and W by context_size // 2."""
    assert not learn_pad_var, 'learn_pad_var is not supported'
    pad = context_size // 2
    assert pad >= 1
    if isinstance(x, np.ndarray):
        if x.ndim == 3:
            return pad_for_probclass3d(x[None], context_size, pad_value)[0]
        pads = [[0, 0], [pad, 0], [pad, pad], [pad, pad]]
        return np.pad(x, pads, mode='constant', constant_values=pad_value)
    N, C, H, W = x.shape
    out = x.new_full((N, C + pad, H + 2 * pad, W + 2 * pad), float(pad_value))
    out[:, pad:, pad:pad + H, pad:pad + W] = x
    return out


def undo_pad_for_probclass3d(x, context_size):
    if isinstance(x, np.ndarray) and x.ndim == 3:
        return undo_pad_for_probclass3d(x[None], context_size)[0]
    pad = context_size // 2
    assert pad >= 1
    return x[:, pad:, pad:-pad, pad:-pad]


def _iter_block_idices(syms_shape, block_sizes):
    C, H, W = syms_shape
    bC, bH, bW = block_sizes
    for c, h, w in itertools.product(range(C - bC + 1), range(H - bH + 1), range(W - bW + 1)):
        yield slice(c, c + bC), slice(h, h + bH), slice(w, w + bW)


def iter_over_blocks(syms, block_sizes):
    """blocks of a CHW volume in raster order C, then H, then W fastest."""
    for cs, hs, ws in _iter_block_idices(syms.shape, block_sizes):
        yield syms[cs, hs, ws]


def num_blocks(syms_shape, block_sizes):
    C, H, W = syms_shape
    bC, bH, bW = block_sizes
    return max(C - bC + 1, 0) * max(H - bH + 1, 0) * max(W - bW + 1, 0)
