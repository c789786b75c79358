"""One training step of the reference's graph on the MI355X kernels (forward in training mode + backward).

Mirrors code/train.py:101-106 (wiring), :303-336 (get_loss), :339-349 (two Adam optimisers), :352-394
(Distortions) and the is_training=True branches of code/autoencoder.py / code/probclass.py:

    enc   = ae.encode(x, is_training=True)                 # BatchNorm on batch statistics
    x_out = ae.decode(enc.qbar, is_training=True)          # decoder sees qbar (straight-through estimator)
    bc    = pc.bitcost(stop_gradient(enc.qbar), enc.symbols, is_training=True, pad_value=centers[0])
    loss  = K_ms_ssim * (1 - MS-SSIM(x, x_out)) + beta * max(0.5 * (mean(bc * heatmap) + mean(bc)) - H_target, 0)
            + L2(encoder, decoder[, pc]) + 0.1 * l2_loss(centers)

TensorFlow derives the backward graph automatically; here it is written out by hand on top of the HIP kernels
(conv data gradients reuse the forward conv kernels, filter gradients use ic_conv2d_wgrad_f32, BatchNorm /
quantiser / context-model gradients have their own kernels).  PyTorch supplies tensors, the MS-SSIM loss
(ms_ssim.py, autograd) and elementwise glue.  Gradients are written straight into flat per-group buffers
(context model | decoder | encoder) so that data-parallel training all-reduces three contiguous buckets over
RCCL, each launched as soon as its group's backward is complete (pc first, then decoder, then encoder).

The step can be driven two ways, one code path underneath:
  * the reference's own call sites (train.py:101-127) through the plugin objects -- TrainGraph.bind(ae, pc), then
    ae.encode(x, True), ae.decode(enc.qbar, True), pc.bitcost(stop_gradient(qbar), symbols, True, pad_value), Distortions,
    get_loss, total_loss.backward(): the three calls are torch.autograd Functions whose backward is the hand-written HIP
    backward of that section (parameter gradients land in the flat buckets; the L2 terms' gradients are folded into the
    filter-gradient kernels, their values are returned by the *_regularization_loss() accessors);
  * TrainGraph.forward_backward(x) / Trainer.step(x), which is exactly that sequence.

BatchNorm under data parallelism: cross-replica statistics by default (sync_bn=None / True) -- the reference normalises over
its whole batch on one device (autoencoder.py:115-125), so 8 ranks x 4 crops must see the statistics of all 32.  Per layer and
direction the ranks sum 2 C float64 values (ic_bn_moments_f32 / ic_bn_backward_reduce_f32): 140 exchanges per step, each on the
critical path.  sync_bn=True / None: an RCCL all-reduce each; sync_bn='p2p': one small launch each over peer-mapped memory
(peer.py, xGMI stores + flags, summed in rank order); sync_bn=False keeps the statistics local to a rank (a per-tower
replication of the TF graph: no exchange at all, but 8 x 4 crops are then not the reference's 32).
"""
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib, ms_ssim
from . import weights as _weights
from ._lib import lib, check, ptr

BN_EPS = 1e-5
# checkpoint names of the two Adam optimisers' beta-power accumulators (TF-1.x `_create_non_slot_variable`: named
# `beta1_power` / `beta2_power`, no optimiser prefix, uniquified `_1` for the second optimiser of the graph)
BETA_POWER_NAMES = {'Adam_AE': ('beta1_power', 'beta2_power'), 'Adam_PC': ('beta1_power_1', 'beta2_power_1')}
BN_DECAY = 0.9
_IMG_MEAN = (121.85369873, 113.58860779, 100.63715363)
_IMG_VAR = (4746.37695312, 4454.13964844, 4812.234375)


def _staged_all_reduce(t, pg):
    """all-reduce(sum) of a device tensor over a backend without device support (gloo: the world-size-2 tests that share
    one GPU between two processes): through a host copy, synchronously.  RCCL reduces device tensors in place."""
    import torch.distributed as dist
    h = t.detach().cpu()
    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=pg)
    t.copy_(h.to(t.device))


def _is_gloo(pg):
    import torch.distributed as dist
    return dist.get_backend(pg) == 'gloo'


class GradBuckets(object):
    """Data-parallel gradient exchange over flat buckets: `ready(name)` starts an asynchronous all-reduce of that
    bucket as soon as the caller has finished writing it, `wait()` blocks on all of them and turns the sums into
    means.  Works on any process group (RCCL on the GPUs; gloo in the CPU tests).  World size 1: no-ops, unless
    `always_reduce` (tests: exercises the RCCL path and its stream ordering on a single GPU)."""

    def __init__(self, flat_tensors, process_group=None, always_reduce=False):
        self.flat = flat_tensors            # dict name -> 1-D tensor
        self.pg = process_group
        self.always_reduce = always_reduce
        self._pending = []

    def _world(self):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.pg)

    def ready(self, name):
        import torch.distributed as dist
        world = self._world()
        if world == 1 and not self.always_reduce:
            return
        if self.flat[name].is_cuda and _is_gloo(self.pg):
            _staged_all_reduce(self.flat[name], self.pg)
            self.flat[name].div_(world)
            return
        # ProcessGroupNCCL orders the collective after the work already queued on the current stream
        work = dist.all_reduce(self.flat[name], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self._pending.append((work, self.flat[name], world))

    def wait(self):
        for work, flat, world in self._pending:
            work.wait()
            flat.div_(world)
        self._pending = []


# Which 3x3 convolutions of a training step run in Winograd F(4x4,3x3) form (2 x 8-tile segments on the crops' small maps; filter
# gradients stay in the F(2x2) domain, conv3x3_wgrad_wino.hip): False / 'fwd' / 'bwd' / True (both).  cfg3 step on the MI355X
# (tools/train_f4_report.py: forward_backward ms; gradient error against float64 autograd, relative to the tensor scale), round 5:
#     False 16.17 ms   'bwd' 15.77 ms   'fwd' 15.90 ms   True 15.36 ms
#   'bwd' leaves every one of the 219 gradients where F(2x2) has it (worst 1.405e-2 either way).
#   'fwd' / True move z from 7.1e-6 to 3.6e-5 of its scale -- inside the 1e-4 the specification sets for OUTPUTS, no symbol flip --
#   and multiply the errors of the gradients the cfg3 test checks by 1.1 - 4.5 (enc_2_2/conv1/weights 9.7e-4 -> 4.3e-3, h12/weights
#   9.6e-4 -> 2.8e-3, to_bn/weights 4.5e-4 -> 1.5e-3): the forward error is what every later layer's gradient is evaluated at.
# Round 6: True is the default -- the specification bounds outputs, which hold; the gradient errors of both settings are in the parity
# report (tests/test_gpu_configs.py::test_cfg3_training_step_full_size[shipped] and [f2x2_forward], the latter with the tight bounds).
# Per TrainGraph (argument `wino4`, else the environment variable IMGCOMP_TRAIN_WINO4 read when the graph is built): no process-wide
# state on this side of the ABI either.
_WINO4_MODES = {'0': False, '': False, 'false': False, 'off': False, 'no': False, '1': True, '2': True, 'true': True, 'both': True,
                'on': True, 'yes': True, 'fwd': 'fwd', 'forward': 'fwd', 'bwd': 'bwd', 'backward': 'bwd'}


def wino4_mode(value=None):
    """normalised F(4x4) mode of a training graph: False, 'fwd', 'bwd' or True.  value None: IMGCOMP_TRAIN_WINO4, default both."""
    if value is None:
        value = os.environ.get('IMGCOMP_TRAIN_WINO4', 'both')
    if isinstance(value, bool):
        return value
    key = str(value).strip().lower()
    if key not in _WINO4_MODES:
        raise ValueError("IMGCOMP_TRAIN_WINO4 / TrainGraph(wino4=...) must be one of 0, 1 (= both), fwd, bwd; got {!r}".format(value))
    return _WINO4_MODES[key]


class _Layer(object):
    __slots__ = ('scope', 'kind', 'kh', 'kw', 'cin', 'cout', 'stride', 'group')


class TrainGraph(object):
    """Parameters, gradients and the hand-written forward/backward of the CVPR autoencoder + res_shallow context model."""

    def __init__(self, ae_config, pc_config, weights, device='cuda', process_group=None, sync_bn=None, wino4=None):
        self.ae_config, self.pc_config = ae_config, pc_config
        self.WINO4 = wino4_mode(wino4)  # which 3x3 convolutions run in F(4x4) form (wino4_mode above)
        self.sync_bn = sync_bn          # None: on whenever the process group has more than one rank
        self.version = 0                # bumped whenever the variables change (bound plugin objects re-fold their inference plan)
        self._hook = None
        self.dev = torch.device(device)
        if self.dev.type != 'cuda':
            raise _lib.HipLibraryError('training runs only on a HIP device, got {}'.format(self.dev))
        self.pg = process_group
        self.B = int(ae_config.arch_param_B)
        self.C = int(ae_config.num_chan_bn)
        self.L = int(ae_config.num_centers)
        self.k = int(pc_config.arch_param__k)
        self.heatmap = bool(ae_config.heatmap)
        if pc_config.kernel_size != 3:
            raise NotImplementedError('context model: kernel_size 3 only')
        # ---- layer table ----
        self.layers = OrderedDict()
        for scope, kind, shape in _weights.ae_conv_specs(self.C, self.B, self.heatmap):
            l = _Layer()
            l.scope, l.kind = scope, kind
            l.kh, l.kw = shape[0], shape[1]
            l.cin, l.cout = (shape[2], shape[3]) if kind == 'conv' else (shape[3], shape[2])
            l.stride = 1 if (kind == 'conv' and l.kh == 3) else 2
            l.group = 'enc' if scope.startswith(_weights.ENC) else 'dec'
            self.layers[scope] = l
        self.pc_scopes = [s for s, _ in _weights.pc_conv_specs(self.L, self.k, 3)]
        # ---- parameters (checkpoint names and TF layouts) and flat gradient buckets ----
        self.params = OrderedDict()
        for name, arr in weights.items():
            self.params[name] = torch.as_tensor(np.ascontiguousarray(arr), dtype=torch.float32).to(self.dev)
        self.trainable = OrderedDict((n, t) for n, t in self.params.items() if 'moving_' not in n)
        groups = {'pc': [], 'dec': [], 'enc': []}
        for n in self.trainable:
            groups['pc' if n.startswith('probclass3d/') else ('dec' if n.startswith(_weights.DEC) else 'enc')].append(n)
        self.group_names = groups
        # Variables and gradients of a group live in two flat buffers with ONE layout (every tensor starts on a 256-byte
        # boundary): a bucket is all-reduced as one message and updated by one fused Adam launch (ic_adam_tf_f32).
        self.flat_grads, self.flat_params, self.grads = {}, {}, OrderedDict()
        for g, names in groups.items():
            offs, total = [], 0
            for n in names:
                offs.append(total)
                total += (self.trainable[n].numel() + 63) // 64 * 64
            flat = torch.zeros(total, dtype=torch.float32, device=self.dev)
            flat_p = torch.zeros(total, dtype=torch.float32, device=self.dev)
            self.flat_grads[g], self.flat_params[g] = flat, flat_p
            for n, off in zip(names, offs):
                t = self.trainable[n]
                k = t.numel()
                view = flat_p[off:off + k].view(t.shape)
                view.copy_(t)
                self.params[n] = self.trainable[n] = view
                self.grads[n] = flat[off:off + k].view(t.shape)
        # ---- constants and scratch ----
        self.ones = torch.ones(256, device=self.dev)
        self.zeros = torch.zeros(256, device=self.dev)
        self.img_mean = torch.tensor(_IMG_MEAN, device=self.dev)
        self.img_std = torch.sqrt(torch.tensor(_IMG_VAR, device=self.dev) + 1e-10)
        self.bn_ws = torch.empty(lib.ic_bn_workspace_bytes(256), dtype=torch.uint8, device=self.dev)
        self.packed3 = lib.ic_conv3x3_c128_both_packed_floats()
        self._ws = {}
        self.buckets = GradBuckets(self.flat_grads, process_group)

    # ------------------------------------------------------------------------------------------------
    # small helpers
    # ------------------------------------------------------------------------------------------------
    def _st(self):
        return _lib.current_stream(self.dev)

    def _scratch(self, key, nbytes):
        t = self._ws.get(key)
        if t is None or t.numel() < nbytes:
            t = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=self.dev)
            self._ws[key] = t
        return t

    def _new(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.dev)

    # ---- raw convolution (no BN, no activation): forward and data-gradient use ----
    def _pack_all_3x3(self, N, H, W):
        """Winograd fragments of every 3x3 filter, forward and adjoint, in two launches (the filters change every step): F(4x4)
        fragments where that kernel takes the shape (TrainGraph.WINO4), else F(2x2).  No-op when the shape runs the direct form."""
        self._wino_pk = None
        if lib.ic_conv3x3_c128_pick_algo(N, H, W, 0) != 1:
            return
        if not hasattr(self, '_w3_names'):
            self._w3_names = [l.scope + '/weights' for l in self.layers.values()
                              if l.kind == 'conv' and l.kh == 3 and l.cin == 128 and l.cout == 128]
            self._w3_index = {n: i for i, n in enumerate(self._w3_names)}
            self._w3_table = torch.tensor([self.params[n].data_ptr() for n in self._w3_names], dtype=torch.int64,
                                          device=self.dev)
            self._w3_bufs = {}
        # F(4x4) from ~160 work-groups on (a work-group's serial chain is ~25 us however small the launch; tools/w4sweep.py: 32 maps of
        # 32 x 32 31.8 against 40.3 us, 30 of 40 x 40 68 against 95) and where at least half of the segments' tiles exist
        wgs = int(lib.ic_wino4_3x3_c128_workgroups(N, H, W))
        tiles = N * (-(-H // 4)) * (-(-W // 4))
        fits = wgs >= 160 and 2 * tiles >= 8 * wgs
        mode = self.WINO4
        # per direction: (forward, data gradient) -- WINO4 True: both, 'fwd' / 'bwd': one of them (A/B runs and the parity report)
        self._w3_f4 = (fits and mode in (True, 'fwd'), fits and mode in (True, 'bwd'))
        bufs = []
        for b in (0, 1):
            f4 = self._w3_f4[b]
            if (b, f4) not in self._w3_bufs:
                n_pk = lib.ic_wino4_3x3_c128_packed_floats() if f4 else lib.ic_wino3x3_c128_packed_floats()
                self._w3_bufs[(b, f4)] = self._new(len(self._w3_names), n_pk)
            buf = self._w3_bufs[(b, f4)]
            pack = lib.ic_pack_wino4_3x3_c128_batch_f32 if f4 else lib.ic_pack_wino3x3_c128_batch_f32
            check(pack(ptr(self._w3_table), ptr(buf), len(self._w3_names), b, self._st()), 'batched winograd pack')
            bufs.append(buf)
        self._wino_pk = bufs

    def _conv3x3(self, x, name, backward=False, res1=None, res2=None):
        """raw 3x3 128->128 conv (backward: its adjoint = the data gradient), + res1 + res2 in the kernel's epilogue (the skip
        gradients of the residual stack: (conv + res1) + res2, the order the separate adds had).  name: a parameter name -- its
        fragments come from this step's batched packing when there is one -- or a filter tensor in the TF layout."""
        N, _, H, W = x.shape
        y = self._new(N, 128, H, W)
        st = self._st()
        w_tf = self.params[name] if isinstance(name, str) else name
        if lib.ic_conv3x3_c128_pick_algo(N, H, W, 0) == 1:
            if isinstance(name, str) and getattr(self, '_wino_pk', None) is not None:
                wp = self._wino_pk[int(backward)][self._w3_index[name]]
                if self._w3_f4[int(backward)]:
                    # Winograd F(4x4,3x3) (csrc/conv3x3_wino4.hip; 2 x 8-tile segments on the crops' small maps): same contract
                    check(lib.ic_wino4_3x3_c128_bn_act_f32(ptr(x), ptr(wp), ptr(self.ones), ptr(self.zeros), ptr(res1), ptr(res2), ptr(y),
                                                           N, H, W, 0, 0, st), 'conv3x3 (winograd F(4x4))')
                    return y
            else:
                wp = self._new(lib.ic_wino3x3_c128_packed_floats())
                check(lib.ic_pack_wino3x3_c128_f32(ptr(w_tf), ptr(wp), int(backward), st))
            check(lib.ic_wino3x3_c128_bn_act_f32(ptr(x), ptr(wp), ptr(self.ones), ptr(self.zeros), ptr(res1), ptr(res2), ptr(y),
                                                 N, H, W, 0, 0, st), 'conv3x3 (winograd)')
        else:
            wp = self._new(lib.ic_conv3x3_c128_packed_floats())
            f = lib.ic_pack_conv3x3_c128_bwd_f32 if backward else lib.ic_pack_conv3x3_c128_f32
            check(f(ptr(w_tf), ptr(wp), st))
            check(lib.ic_conv3x3_c128_bn_act_f32(ptr(x), ptr(wp), ptr(self.ones), ptr(self.zeros), ptr(res1), ptr(res2), ptr(y),
                                                 N, H, W, 0, 0, st), 'conv3x3')
        return y

    def _conv_s(self, x, w_tf, kh, kw, cin, cout, stride):
        """TF-SAME conv with filter [kh,kw,cin,cout]; matrix-core path when the shape has one."""
        N, _, H, W = x.shape
        y = self._new(N, cout, -(-H // stride), -(-W // stride))
        n = lib.ic_conv2d_mfma_packed_floats(kh, kw, cin, cout, stride, 0)
        if n:
            wp = self._new(n)
            check(lib.ic_pack_conv2d_mfma_f32(ptr(w_tf), ptr(wp), kh, kw, cin, cout, stride, 0, self._st()))
            check(lib.ic_conv2d_mfma_bn_act_f32(ptr(x), ptr(wp), ptr(self.ones), ptr(self.zeros), ptr(y), N, cin, H, W,
                                                cout, kh, kw, stride, 0, 0, self._st()), 'conv mfma')
        else:
            check(lib.ic_conv2d_bn_act_f32(ptr(x), ptr(w_tf), ptr(self.ones), ptr(self.zeros), None, None, ptr(y),
                                           N, cin, H, W, cout, kh, kw, stride, 0, None, None, self._st()), 'conv direct')
        return y

    def _deconv_s(self, x, w_tf, kh, kw, cin, cout):
        """stride-2 TF-SAME transposed conv with filter [kh,kw,cout,cin]."""
        N, _, H, W = x.shape
        y = self._new(N, cout, 2 * H, 2 * W)
        n = lib.ic_conv2d_mfma_packed_floats(kh, kw, cin, cout, 2, 1)
        if n:
            wp = self._new(n)
            check(lib.ic_pack_conv2d_mfma_f32(ptr(w_tf), ptr(wp), kh, kw, cin, cout, 2, 1, self._st()))
            check(lib.ic_conv2d_mfma_bn_act_f32(ptr(x), ptr(wp), ptr(self.ones), ptr(self.zeros), ptr(y), N, cin, H, W,
                                                cout, kh, kw, 2, 1, 0, self._st()), 'deconv mfma')
        else:
            check(lib.ic_deconv2d_bn_act_f32(ptr(x), ptr(w_tf), ptr(self.ones), ptr(self.zeros), ptr(y), N, cin, H, W,
                                             cout, kh, kw, 0, None, None, 0, self._st()), 'deconv direct')
        return y

    def _raw_forward(self, l, x):
        w = self.params[l.scope + '/weights']
        if l.kind == 'conv' and l.kh == 3 and l.cin == 128 and l.cout == 128:
            return self._conv3x3(x, l.scope + '/weights')
        if l.kind == 'conv':
            return self._conv_s(x, w, l.kh, l.kw, l.cin, l.cout, l.stride)
        return self._deconv_s(x, w, l.kh, l.kw, l.cin, l.cout)

    def _raw_backward_data(self, l, g, add1=None, add2=None):
        """gradient wrt the layer input of the raw conv (g = gradient wrt its output), + add1 + add2."""
        w = self.params[l.scope + '/weights']
        if l.kind == 'conv' and l.kh == 3 and l.cin == 128 and l.cout == 128:
            return self._conv3x3(g, l.scope + '/weights', backward=True, res1=add1, res2=add2)
        if l.kind == 'conv':      # adjoint of a strided conv = transposed conv with the SAME array read as [kh,kw,out=cin,in=cout]
            dx = self._deconv_s(g, w, l.kh, l.kw, l.cout, l.cin)
        else:                     # adjoint of a transposed conv = strided conv with the same array read as [kh,kw,cin=cout_l,cout=cin_l]
            dx = self._conv_s(g, w, l.kh, l.kw, l.cout, l.cin, 2)
        for a in (add1, add2):
            if a is not None:
                dx = dx + a
        return dx

    WINOGRAD_WGRAD = True     # 3x3 128 -> 128 filter gradients in the Winograd domain (ic_conv3x3_c128_wgrad_f32)
    WGRAD_SIDE_STREAM = False  # filter gradients on a second stream beside the data-gradient chain: measured SLOWER (see _wgrad_beside)

    def _wgrad_beside(self, l, x_in, g):
        """The filter gradient of a layer depends on the layer's input (tape) and on the gradient of its raw output, and nothing
        in the backward chain depends on IT: it runs on a side stream while the main stream goes on with the data gradient and
        the next layer's BatchNorm backward.  Both are full-chip launches, so what the overlap buys is the bubbles -- the kernel
        boundaries of the chain and the BatchNorm reductions' 128 work-groups (one per channel: half the CUs idle).  Same
        kernels, same operands, same order per gradient buffer: the gradients are bit-identical to the serial order (tests pass
        with it on).  MEASURED on the MI355X (cfg3 step): 19.7 ms against 16.9 ms serial -- both kernels want every CU whole (512
        registers per SIMD lane each), so their work-groups interleave on the CUs and the data-gradient chain, which is the
        critical path, waits behind filter-gradient work-groups.  Off by default."""
        if not self.WGRAD_SIDE_STREAM:
            return self._wgrad(l, x_in, g)
        main = torch.cuda.current_stream(self.dev)
        if getattr(self, '_wgrad_stream', None) is None:
            self._wgrad_stream = torch.cuda.Stream(device=self.dev)
        side = self._wgrad_stream
        side.wait_stream(main)                       # g is complete
        with torch.cuda.stream(side):
            self._wgrad(l, x_in, g)
        g.record_stream(side)                        # the caching allocator must not hand these blocks out again before
        x_in.record_stream(side)                     # the side stream is done with them
        self._wgrad_pending = True

    def _join_wgrad(self):
        if getattr(self, '_wgrad_pending', False):
            torch.cuda.current_stream(self.dev).wait_stream(self._wgrad_stream)
            self._wgrad_pending = False

    def _wgrad(self, l, x_in, g):
        N = x_in.shape[0]
        dw = self.grads[l.scope + '/weights']
        wd = float(self.ae_config.regularization_factor)
        w = self.params[l.scope + '/weights']
        if l.kind == 'conv' and l.kh == 3 and l.cin == 128 and l.cout == 128 and self.WINOGRAD_WGRAD:
            H, W = x_in.shape[2], x_in.shape[3]
            need = lib.ic_conv3x3_c128_wgrad_workspace_bytes(N, H, W)
            if need:                                   # 0: odd width / beyond 31-bit offsets -> the direct form below
                ws = self._scratch('wgrad', need)
                check(lib.ic_conv3x3_c128_wgrad_f32(ptr(x_in), ptr(g), ptr(dw), N, H, W, ptr(w), wd, ptr(ws), need, self._st()),
                      'winograd wgrad ' + l.scope)
                return
        if l.kind == 'conv':
            U, V, A, Bc, UH, UW = x_in, g, l.cin, l.cout, x_in.shape[2], x_in.shape[3]
        else:
            U, V, A, Bc, UH, UW = g, x_in, l.cout, l.cin, g.shape[2], g.shape[3]
        VH, VW = -(-UH // l.stride), -(-UW // l.stride)
        need = lib.ic_conv2d_wgrad_workspace_bytes(N, A, Bc, VH, VW, l.kh, l.kw)
        ws = self._scratch('wgrad', need)
        check(lib.ic_conv2d_wgrad_f32(ptr(U), ptr(V), ptr(dw), N, A, UH, UW, Bc, l.kh, l.kw, l.stride, ptr(w), wd,
                                      ptr(ws), need, self._st()), 'wgrad ' + l.scope)

    # ---- conv + BatchNorm(train) + activation (+ residual adds) ----
    def _bn_world(self):
        """ranks whose statistics a BatchNorm layer sums over: 1 unless sync BatchNorm is on and there is more than one rank"""
        import torch.distributed as dist
        if self.sync_bn is False or not (dist.is_available() and dist.is_initialized()):
            return 1
        w = dist.get_world_size(self.pg)
        return w if (self.sync_bn or self.sync_bn is None) else 1

    def _allreduce_sums(self, sums):
        import torch.distributed as dist
        if self.sync_bn == 'p2p':
            # one small launch over peer-mapped memory instead of a collective-library call (peer.py / csrc/peer_exchange.hip);
            # the exchange is created on first use -- its set-up is a collective: every rank gets here in the same layer
            if getattr(self, '_peer', None) is None:
                from . import peer
                self._peer = peer.PeerExchange(self.dev, self.pg)
            self._peer.allreduce_f64(sums)
            self._peer_calls = getattr(self, '_peer_calls', 0) + 1
            if self._peer_calls % 1024 == 1:
                self._peer.check_status()                  # (synchronises: once per ~7 steps) a stalled peer raises instead of training on
            return
        if _is_gloo(self.pg):
            _staged_all_reduce(sums, self.pg)
        else:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.pg)

    FUSE_BN_STATS = True      # 3x3 layers in F(4x4) form: the convolution's epilogue leaves the per-segment channel sums, BatchNorm is ONE
                              # launch (ic_wino4_3x3_c128_raw_stats_f32 + ic_bn_train_forward_cstats_f32) instead of statistics pass + apply

    def _conv3x3_with_stats(self, l, x):
        """the raw 3x3 conv of layer l on the F(4x4) kernel with its BatchNorm sums, or None when this layer / shape / mode does not
        take that path (then: _raw_forward + the two-pass BatchNorm)"""
        if not (self.FUSE_BN_STATS and l.kind == 'conv' and l.kh == 3 and l.cin == 128 and l.cout == 128):
            return None
        if getattr(self, '_wino_pk', None) is None or not self._w3_f4[0] or self._bn_world() != 1:
            return None
        N, _, H, W = x.shape
        parts = int(lib.ic_wino4_3x3_c128_stats_parts(N, H, W))
        if parts <= 0:
            return None
        raw, cst = self._new(N, 128, H, W), self._new(128, parts, 2)
        wp = self._wino_pk[0][self._w3_index[l.scope + '/weights']]
        check(lib.ic_wino4_3x3_c128_raw_stats_f32(ptr(x), ptr(wp), ptr(raw), ptr(cst), N, H, W, 0, self._st()), 'conv3x3 + BatchNorm sums')
        return raw, cst, parts

    def _cba_fwd(self, scope, x, relu, res1=None, res2=None, tape=None):
        l = self.layers[scope]
        fused = self._conv3x3_with_stats(l, x)
        raw = fused[0] if fused else self._raw_forward(l, x)
        N, Cc, H, W = raw.shape
        stats = self._new(4, Cc)
        mean, invstd, scale, shift = stats[0], stats[1], stats[2], stats[3]
        P = self.params
        world = self._bn_world()
        y = self._new(N, Cc, H, W)
        if fused:
            check(lib.ic_bn_train_forward_cstats_f32(ptr(raw), ptr(fused[1]), fused[2], ptr(P[scope + '/BatchNorm/gamma']),
                                                     ptr(P[scope + '/BatchNorm/beta']), ptr(P[scope + '/BatchNorm/moving_mean']),
                                                     ptr(P[scope + '/BatchNorm/moving_variance']), BN_DECAY, BN_EPS, ptr(mean), ptr(invstd),
                                                     ptr(scale), ptr(shift), ptr(res1), ptr(res2), ptr(y), N, Cc, H * W, int(relu),
                                                     self._st()), 'bn forward (conv sums)')
        elif world == 1:
            # batch statistics, folded scale / shift, the moving-average update (decay 0.9) and the normalised, activated output
            # (+ the residual adds) in one call of two launches
            check(lib.ic_bn_train_forward_f32(ptr(raw), ptr(P[scope + '/BatchNorm/gamma']), ptr(P[scope + '/BatchNorm/beta']),
                                              ptr(P[scope + '/BatchNorm/moving_mean']), ptr(P[scope + '/BatchNorm/moving_variance']),
                                              BN_DECAY, BN_EPS, ptr(mean), ptr(invstd), ptr(scale), ptr(shift), ptr(res1), ptr(res2),
                                              ptr(y), N, Cc, H * W, int(relu), ptr(self.bn_ws), self._st()), 'bn forward')
        else:
            # cross-replica statistics: this rank's float64 moments, summed over the ranks, folded over the global count
            sums = torch.empty(2 * Cc, dtype=torch.float64, device=self.dev)
            check(lib.ic_bn_moments_f32(ptr(raw), ptr(sums), N, Cc, H * W, ptr(self.bn_ws), self._st()), 'bn moments')
            self._allreduce_sums(sums)
            check(lib.ic_bn_train_fold_moments_f32(ptr(sums), N * H * W * world, ptr(P[scope + '/BatchNorm/gamma']),
                                                   ptr(P[scope + '/BatchNorm/beta']), ptr(P[scope + '/BatchNorm/moving_mean']),
                                                   ptr(P[scope + '/BatchNorm/moving_variance']), BN_DECAY, BN_EPS, ptr(mean),
                                                   ptr(invstd), ptr(scale), ptr(shift), Cc, self._st()), 'bn fold')
            check(lib.ic_bn_apply_f32(ptr(raw), ptr(scale), ptr(shift), ptr(res1), ptr(res2), ptr(y), N, Cc, H * W,
                                      int(relu), self._st()))
        if tape is not None:
            tape.append((l, x, raw, mean, invstd, scale, shift, relu))
        return y

    def _cba_bwd(self, rec, dy, need_dx=True, add1=None, add2=None):
        l, x, raw, mean, invstd, scale, shift, relu = rec
        N, Cc, H, W = raw.shape
        draw = self._new(N, Cc, H, W)
        gamma = self.params[l.scope + '/BatchNorm/gamma']
        dgamma, dbeta = self.grads[l.scope + '/BatchNorm/gamma'], self.grads[l.scope + '/BatchNorm/beta']
        world = self._bn_world()
        if world == 1:
            check(lib.ic_bn_backward_f32(ptr(dy), ptr(raw), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(gamma), ptr(draw),
                                         ptr(dgamma), ptr(dbeta), N, Cc, H * W, int(relu), ptr(self.bn_ws), self._st()), 'bn backward')
        else:
            # dgamma / dbeta stay this rank's sums (the bucket all-reduce averages them like every parameter gradient);
            # the data gradient needs the sums over ALL ranks' elements
            sums = torch.empty(2 * Cc, dtype=torch.float64, device=self.dev)
            check(lib.ic_bn_backward_reduce_f32(ptr(dy), ptr(raw), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(sums),
                                                ptr(dgamma), ptr(dbeta), N, Cc, H * W, int(relu), ptr(self.bn_ws), self._st()),
                  'bn backward reduce')
            self._allreduce_sums(sums)
            check(lib.ic_bn_backward_apply_f32(ptr(dy), ptr(raw), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(gamma),
                                               ptr(sums), N * H * W * world, ptr(draw), N, Cc, H * W, int(relu), self._st()),
                  'bn backward apply')
        self._wgrad_beside(l, x, draw)
        return self._raw_backward_data(l, draw, add1, add2) if need_dx else None

    # ---- residual stack (autoencoder.py:224-234 / :252-262) ----
    def _stack_names(self, kind):
        root = _weights.ENC if kind == 'enc' else _weights.DEC
        final = 'res_block_enc_final' if kind == 'enc' else 'dec_after_res'
        blocks = [['{}/res_block_{}_{}/{}_{}_{}'.format(root, kind, b, kind, b, i) for i in (1, 2, 3)]
                  for b in range(self.B)]
        return blocks, '{}/{}'.format(root, final)

    def _stack_fwd(self, kind, x, tape):
        blocks, final = self._stack_names(kind)
        res0 = net = x
        for grp in blocks:
            res_b = net
            for i, blk in enumerate(grp):
                t = self._cba_fwd(blk + '/conv1', net, True, tape=tape)
                net = self._cba_fwd(blk + '/conv2', t, False, res1=net, res2=res_b if i == 2 else None, tape=tape)
        t = self._cba_fwd(final + '/conv1', net, False, tape=tape)
        return self._cba_fwd(final + '/conv2', t, False, res1=net, res2=res0, tape=tape)

    def _stack_bwd(self, tape, g):
        """tape holds the 6B+2 records of the stack in forward order; returns the gradient wrt the stack input."""
        recs = list(tape)
        # final block: out = conv2(conv1(net)) + net + res0
        g_res0 = g
        # the skip gradients ride in the epilogue of the data-gradient conv: (conv path + skip) + group skip
        gt = self._cba_bwd(recs.pop(), g)
        g_net = self._cba_bwd(recs.pop(), gt, add1=g)
        for b in reversed(range(self.B)):
            g_resb = None
            for i in (2, 1, 0):
                gout = g_net                                   # gradient wrt this block's output
                if i == 2:
                    g_resb = gout                              # group skip taps the last block's output sum
                gt = self._cba_bwd(recs.pop(), gout)
                g_net = self._cba_bwd(recs.pop(), gt, add1=gout, add2=g_resb if i == 0 else None)   # block input: skip + conv path
        assert not recs
        return g_net + g_res0

    # ------------------------------------------------------------------------------------------------
    # the step, in sections: each forward leaves its tape in self._tape, each backward consumes it
    # ------------------------------------------------------------------------------------------------
    def encode_train(self, x):
        """autoencoder.py:218-244 with is_training=True.  -> dict(qbar, qhard, symbols, z, heatmap, qsoft)"""
        cfg = self.ae_config
        st = self._st()
        N, _, H, W = x.shape
        assert H % 8 == 0 and W % 8 == 0
        C, L = self.C, self.L
        h, w = H // 8, W // 8
        self._pack_all_3x3(N, H // 4, W // 4)
        ta, ts, tb = [], [], []
        # _normalize (autoencoder.py:136-144): kept as its own tensor, h1's filter gradient needs the normalised input
        xn = (x - self.img_mean.view(1, 3, 1, 1)) / self.img_std.view(1, 3, 1, 1) if cfg.normalization == 'FIXED' else x
        net = self._cba_fwd(_weights.ENC + '/h1', xn.contiguous(), True, tape=ta)
        net = self._cba_fwd(_weights.ENC + '/h2', net, True, tape=ta)
        net = self._stack_fwd('enc', net, ts)
        bott = self._cba_fwd(_weights.ENC + '/to_bn', net, False, tape=tb)
        centers = self.params[_weights.ENC + '/centers']
        mk = lambda: self._new(N, C, h, w)
        hm, z, qsoft, qhard, qbar = (mk() if self.heatmap else None), mk(), mk(), mk(), mk()
        symbols = torch.empty((N, C, h, w), dtype=torch.int64, device=self.dev)
        if self.heatmap:
            check(lib.ic_heatmap_quantize_f32(ptr(bott), ptr(centers), L, 1.0, ptr(hm), ptr(z), ptr(qsoft), ptr(qhard),
                                              ptr(qbar), ptr(symbols), N, C, h, w, st))
        else:
            check(lib.ic_quantize_f32(ptr(bott), ptr(centers), L, 1.0, ptr(qsoft), ptr(qhard), ptr(symbols),
                                      bott.numel(), st))
            z = bott
            qbar = qsoft + (qhard - qsoft)
        self._tape_enc = (ta, ts, tb, bott, (N, C, h, w))
        return {'qbar': qbar, 'qhard': qhard, 'symbols': symbols, 'z': z, 'heatmap': hm, 'qsoft': qsoft}

    def decode_train(self, qbar):
        """autoencoder.py:246-268 with is_training=True, up to (not including) _denormalize / clip -> the normalised image"""
        ta, ts, tb = [], [], []
        net = self._cba_fwd(_weights.DEC + '/from_bn', qbar.contiguous(), True, tape=ta)
        net = self._stack_fwd('dec', net, ts)
        net = self._cba_fwd(_weights.DEC + '/h12', net, True, tape=tb)
        pre = self._cba_fwd(_weights.DEC + '/h13', net, False, tape=tb)
        self._tape_dec = (ta, ts, tb)
        return pre

    def bitcost_train(self, q, symbols, pad_value):
        """probclass.py:63-106 with is_training=True -> bit cost per symbol (N,C,h,w)"""
        N, C, h, w = q.shape
        wtab_t = []
        for sc in self.pc_scopes:
            wtab_t += [self.params[sc + '/weights'], self.params[sc + '/biases']]
        pc_need = lib.ic_pc_workspace_bytes(N, C, h, w, self.k)
        pc_ws = torch.empty(pc_need, dtype=torch.uint8, device=self.dev)
        logits = self._new(N, C, h, w, self.L)
        bc = self._new(N, C, h, w)
        q = q.contiguous()
        check(lib.ic_pc_bitcost_f32(ptr(q), ptr(symbols), _lib.ptr_table(wtab_t + [None]), self.k, self.L, float(pad_value), ptr(logits),
                                    ptr(bc), N, C, h, w, ptr(pc_ws), pc_need, self._st()), 'pc forward')
        self._tape_pc = (q, symbols, logits, pc_ws, float(pad_value), (N, C, h, w))
        return bc

    def backward_pc(self, d_bc):
        q, symbols, logits, pc_ws, pad_value, (N, C, h, w) = self._tape_pc
        self._tape_pc = None
        self._pc_backward(q, symbols, logits, d_bc, pc_ws, pad_value, N, C, h, w)
        self._bucket_ready('pc')

    def backward_decoder(self, g_pre):
        ta, ts, tb = self._tape_dec
        self._tape_dec = None
        g = self._cba_bwd(tb.pop(), g_pre.contiguous())
        g = self._cba_bwd(tb.pop(), g)
        g = self._stack_bwd(ts, g)
        g_qbar = self._cba_bwd(ta.pop(), g)
        self._bucket_ready('dec')
        return g_qbar

    def backward_encoder(self, g_qbar, d_hm):
        cfg = self.ae_config
        ta, ts, tb, bott, (N, C, h, w) = self._tape_enc
        self._tape_enc = None
        centers = self.params[_weights.ENC + '/centers']
        d_bott = torch.empty_like(bott)
        d_centers = self.grads[_weights.ENC + '/centers']
        qws = self._scratch('qbwd', lib.ic_heatmap_quantize_bwd_workspace_bytes(self.L))
        g_qbar = (g_qbar if g_qbar is not None else torch.zeros((N, C, h, w), device=self.dev)).contiguous()
        d_hm = d_hm.contiguous() if (d_hm is not None and self.heatmap) else None
        check(lib.ic_heatmap_quantize_bwd_f32(ptr(bott), ptr(centers), self.L, 1.0, ptr(g_qbar), ptr(d_hm), ptr(d_bott),
                                              ptr(d_centers), N, C, h, w, int(self.heatmap), ptr(qws), self._st()),
              'quantiser backward')
        if cfg.regularization_factor_centers != 0:
            d_centers.add_(centers, alpha=float(cfg.regularization_factor_centers))
        g = self._cba_bwd(tb.pop(), d_bott)
        g = self._stack_bwd(ts, g)
        g = self._cba_bwd(ta.pop(), g)
        self._cba_bwd(ta.pop(), g, need_dx=False)
        self._bucket_ready('enc')

    # ---- the reference's call sites (train.py:101-105) as autograd Functions over the sections above ----
    def _grad_hook(self):
        # a leaf that requires grad, so that autograd records the Functions although none of their tensor inputs does
        if self._hook is None:
            self._hook = torch.zeros((), device=self.dev, requires_grad=True)
        return self._hook

    def plugin_encode(self, x):
        from . import autoencoder
        _lib.require_cuda(x, 'x')
        outs = _EncodeTrainFn.apply(self, x.contiguous(), self._grad_hook())
        qbar, hm, qhard, symbols, z, qsoft = outs
        self._last_qsoft = qsoft
        return autoencoder.EncoderOutput(qbar, qhard, symbols, z, hm if self.heatmap else None)

    def plugin_decode(self, q):
        pre = _DecodeTrainFn.apply(self, q, self._grad_hook())
        # _denormalize + _clip_to_image_range (autoencoder.py:146-158,267) in torch: autograd carries the loss gradient to `pre`
        if self.ae_config.normalization == 'FIXED':
            return torch.clamp(pre * self.img_std.view(1, 3, 1, 1) + self.img_mean.view(1, 3, 1, 1), 0, 255)
        return torch.clamp(pre, 0, 255)

    def plugin_bitcost(self, q, target_symbols, pad_value):
        if q.requires_grad:
            raise NotImplementedError('the context model is trained on stop_gradient(qbar) (train.py:103-104): pass q.detach()')
        # a tensor pad value is centers[0] (pc.auto_pad_value(ae), probclass.py:59-61): the C ABI takes it by value, the
        # graph keeps an asynchronously refreshed pinned copy (no host stall mid-step)
        pv = self._pad_value() if torch.is_tensor(pad_value) else float(pad_value)
        return _BitcostTrainFn.apply(self, q, target_symbols, pv, self._grad_hook())

    def finish_backward(self):
        """after total_loss.backward(): wait for the gradient all-reduces (what get_train_op's train_op depends on)"""
        self._wait_buckets()

    def bind(self, ae, pc):
        """make ae.encode / ae.decode / pc.bitcost with is_training=True run this graph's training-mode forward, and their
        is_training=False calls evaluate the CURRENT training variables (test-in-train, train.py:115-127)"""
        ae._train_graph = self
        pc._train_graph = self
        ae._graph_version = pc._graph_version = -1
        return ae, pc

    def forward_backward(self, x):
        """x: (N,3,H,W) float32 0..255 on the device.  Fills self.grads; returns a dict of python floats.
        train.py:101-106 + get_loss + the backward of total_loss, through the same Functions the plugin call sites use."""
        cfg = self.ae_config
        N, _, H, W = x.shape
        enc = self.plugin_encode(x)
        x_out = self.plugin_decode(enc.qbar)
        pad_value = self._pad_value() if self.pc_config.use_centers_for_padding else 0.0
        hd = self._hip_distortion(x) if (self.HIP_LOSS and cfg.distortion_to_minimize == 'ms_ssim') else None
        if hd is not None:
            # The distortion and its gradient with respect to x_out come from ONE call -- csrc/msssim.hip (16 launches, ~0.1 ms) --
            # so the backward is staged by hand: the rate branch first (its bucket's all-reduce goes out first), then clip /
            # de-normalise + decoder + encoder with the distortion's gradient fed in.  Same Functions, same kernels, same sums
            # as the single-backward form below.
            d = hd.launch(x, x_out.detach())
            bc = self.plugin_bitcost(enc.qbar.detach(), enc.symbols, pad_value)
            zero = torch.zeros((), device=self.dev)
            _, H_real, pc_comps, _ = get_loss(cfg, None, None, zero, bc, enc.heatmap)
            pc_loss = dict(pc_comps)['pc_loss']
            roots = [bc] + ([enc.heatmap] if enc.heatmap is not None else [])
            grads = torch.autograd.grad(pc_loss, roots, allow_unused=True)
            d_bc = grads[0] if grads[0] is not None else torch.zeros_like(bc)
            d_hm = grads[1] if len(grads) > 1 else None
            torch.autograd.backward([bc], [d_bc])                       # context-model backward (its bucket goes out first)
            g_qbar, = torch.autograd.grad(x_out, [enc.qbar], d.grad)    # clip / de-normalise + decoder backward
            if d_hm is not None:
                torch.autograd.backward([enc.qbar, enc.heatmap], [g_qbar, d_hm])
            else:
                torch.autograd.backward([enc.qbar], [g_qbar])
        else:
            bc = self.plugin_bitcost(enc.qbar.detach(), enc.symbols, pad_value)
            d = self._distortions(x, x_out)
            total, H_real, pc_comps, ae_comps = get_loss(cfg, None, None, d.d_loss_scaled, bc, enc.heatmap)
            total.backward()
        self.finish_backward()
        # one device -> host transfer for all the scalars of the step
        pcd = dict(pc_comps)
        stats = [d.d_loss_scaled.detach(), pcd['pc_loss'].detach(), H_real.detach(), pcd['H_mask'].detach(), bc.detach().sum() / (N * H * W)] + \
                ([d.ms_ssim.detach()] if d.ms_ssim is not None else [])
        vals = torch.stack([t.to(torch.float32).reshape(()) for t in stats]).tolist()
        out = dict(zip(['d_loss_scaled', 'pc_loss', 'H_real', 'H_mask', 'bpp', 'ms_ssim'], vals))
        self.last = {'x_out': x_out.detach(), 'symbols': enc.symbols, 'bc': bc.detach(), 'heatmap': enc.heatmap.detach() if enc.heatmap is not None else None,
                     'z': enc.z, 'qbar': enc.qbar.detach()}
        return out

    # (Rounds 2-3 replayed the torch distortion from a captured HIP graph; inside the training loop the replay read stale
    # intermediates after 5-10 steps -- a fault of graph replay in this ROCm build, docs/history/DESIGN_rounds_1_4.md -- and
    # csrc/msssim.hip made it unnecessary.  Removed in round 5.)
    # HIP_LOSS: the MS-SSIM distortion and its gradient from csrc/msssim.hip (one launch per scale and direction) instead of the
    # ~300 torch kernels of ms_ssim.py -- no graph, no static buffers, any shape with five scales.  False: torch, eagerly.
    HIP_LOSS = True
    def _hip_distortion(self, x):
        """the HIP distortion of this input shape, or None when MS-SSIM is undefined for it (fewer than five scales)"""
        key = tuple(x.shape)
        cache = self.__dict__.setdefault('_hip_distortions', {})
        if key not in cache:
            try:
                cache[key] = _HipMsSsimDistortion(self.ae_config, x.shape, self.dev)
            except NotImplementedError:
                cache[key] = None
        return cache[key]

    def _distortions(self, x, x_out):
        """Distortions(cfg, x, x_out, is_training=True): MS-SSIM and its gradient from csrc/msssim.hip where the shape has five scales,
        else the torch restatement (ms_ssim.py), eagerly."""
        if self.HIP_LOSS and self.ae_config.distortion_to_minimize == 'ms_ssim':
            hd = self._hip_distortion(x)
            if hd is not None:
                return hd(x, x_out)
        return Distortions(self.ae_config, x, x_out, is_training=True)

    # ---- centres[0] on the host (the context model's pad value is a by-value argument of the C ABI) ----
    def refresh_pad_value(self):
        """asynchronous copy of centres[0] into pinned memory; call after the variables changed (Trainer.step does)."""
        if not hasattr(self, '_c0_host'):
            self._c0_host = torch.empty(1, dtype=torch.float32).pin_memory()
            self._c0_event = torch.cuda.Event()
        self._c0_host.copy_(self.params[_weights.ENC + '/centers'][:1], non_blocking=True)
        self._c0_event.record(torch.cuda.current_stream(self.dev))
        self._c0_fresh = True

    def _pad_value(self):
        if getattr(self, '_c0_fresh', False):
            self._c0_event.synchronize()               # recorded a whole step ago: does not stall
            return float(self._c0_host[0])
        return float(self.params[_weights.ENC + '/centers'][0])

    def regularization_loss(self):
        """value of the L2 terms (their gradients are already folded into the filter gradients)."""
        f = float(self.ae_config.regularization_factor)
        tot = sum(float((t * t).sum()) for n, t in self.trainable.items()
                  if n.startswith('autoencoder/') and n.endswith('/weights')) * 0.5 * f
        c = self.params[_weights.ENC + '/centers']
        tot += float(self.ae_config.regularization_factor_centers) * 0.5 * float((c * c).sum())
        if self.pc_config.regularization_factor is not None:
            tot += sum(float((t * t).sum()) for n, t in self.trainable.items()
                       if n.startswith('probclass3d/') and n.endswith('/weights')) * 0.5 * float(self.pc_config.regularization_factor)
        return tot

    # ---- context-model backward (probclass.py:185-261 in reverse) ----
    def _pc_backward(self, q, symbols, logits, d_bc, pc_ws, pad_value, N, C, h, w):
        st, k, L = self._st(), self.k, self.L
        f32 = pc_ws.view(torch.float32)
        n0 = N * k * (C + 3) * (h + 6) * (w + 6)
        n1 = N * k * (C + 2) * (h + 4) * (w + 4)
        n2 = N * k * (C + 1) * (h + 2) * (w + 2)
        b0, b1, b2 = f32[:n0], f32[n0:n0 + n1], f32[n0 + n1:n0 + n1 + n2]
        s0, s1, s2, s3 = self.pc_scopes
        P, G = self.params, self.grads
        cs_ws = self._scratch('chsum', lib.ic_channel_sum_workspace_bytes(max(k, L)))

        def wgrad(U, qv, V, scope, A, Bc, VD, VH, VW, first):
            need = lib.ic_pc_wgrad_workspace_bytes(N, A, Bc, VD, VH, VW)
            ws = self._scratch('pcwgrad', need)
            check(lib.ic_pc_wgrad_f32(ptr(U), ptr(qv), pad_value, ptr(V), ptr(G[scope + '/weights']), N, A, Bc, VD, VH, VW,
                                      int(first), ptr(ws), need, st), 'pc wgrad ' + scope)
            check(lib.ic_channel_sum_f32(ptr(V), ptr(G[scope + '/biases']), N, Bc, VD * VH * VW, ptr(cs_ws), st))
            rf = self.pc_config.regularization_factor
            if rf is not None:
                G[scope + '/weights'].add_(P[scope + '/weights'], alpha=float(rf))

        def bwd_data(g, scope, res, act, Cin, Cout, OD, OH, OW, relu_mask):
            dx = self._new(N, Cin, OD + 1, OH + 2, OW + 2)
            need = lib.ic_pc_bwd_data_workspace_bytes(N, Cin, Cout, OD, OH, OW)      # 0: the VALU kernel serves the shape
            ws = self._scratch('pcbwd', need) if need else None
            check(lib.ic_pc_bwd_data_f32(ptr(g), ptr(P[scope + '/weights']), ptr(res), ptr(act), ptr(dx), N, Cin, Cout,
                                         OD, OH, OW, 0, int(relu_mask), ptr(ws), need, st), 'pc bwd data ' + scope)
            return dx
        # logits (post-ReLU) -> g3 planar (N, L, C*h*w)
        g3 = self._new(N, L, C, h, w)
        d_bc = d_bc.contiguous()
        check(lib.ic_pc_dlogits_f32(ptr(logits), ptr(symbols), ptr(d_bc), ptr(g3), N, C * h * w, L, st))
        wgrad(b2, None, g3, s3, k, L, C, h, w, False)
        gb2 = bwd_data(g3, s3, None, None, k, L, C, h, w, False)                               # (N,k,C+1,h+2,w+2)
        wgrad(b1, None, gb2, s2, k, k, C + 1, h + 2, w + 2, False)
        gb1 = bwd_data(gb2, s2, None, b1, k, k, C + 1, h + 2, w + 2, True)                      # (N,k,C+2,h+4,w+4)
        wgrad(b0, None, gb1, s1, k, k, C + 2, h + 4, w + 4, False)
        gb0 = bwd_data(gb1, s1, gb2, b0, k, k, C + 2, h + 4, w + 4, True)                       # (N,k,C+3,h+6,w+6)
        wgrad(None, q, gb0, s0, 1, k, C + 3, h + 6, w + 6, True)

    # ---- data-parallel gradient exchange: three flat buckets, each reduced as soon as it is complete ----
    def _bucket_ready(self, group):
        self._join_wgrad()                           # the group's filter gradients are complete before anybody reads the bucket
        self.buckets.ready(group)

    def _wait_buckets(self):
        self.buckets.wait()


class _EncodeTrainFn(torch.autograd.Function):
    """ae.encode(x, is_training=True): outputs (qbar, heatmap, qhard, symbols, z, qsoft); qbar and heatmap carry gradients
    (train.py:101-106: the decoder reads qbar, the rate loss reads the heatmap), backward = quantiser + importance map +
    encoder backward, parameter gradients into the graph's encoder bucket."""

    @staticmethod
    def forward(ctx, graph, x, hook):
        o = graph.encode_train(x)
        ctx.graph = graph
        hm = o['heatmap'] if o['heatmap'] is not None else torch.zeros((), device=x.device)
        ctx.mark_non_differentiable(o['qhard'], o['symbols'], o['z'], o['qsoft'])
        return o['qbar'], hm, o['qhard'], o['symbols'], o['z'], o['qsoft']

    @staticmethod
    def backward(ctx, g_qbar, g_hm, *unused):
        ctx.graph.backward_encoder(g_qbar, g_hm if ctx.graph.heatmap else None)
        return None, None, None


class _DecodeTrainFn(torch.autograd.Function):
    """ae.decode(q, is_training=True) up to the normalised image; backward = decoder backward -> gradient wrt q."""

    @staticmethod
    def forward(ctx, graph, q, hook):
        ctx.graph = graph
        return graph.decode_train(q)

    @staticmethod
    def backward(ctx, g_pre):
        return None, ctx.graph.backward_decoder(g_pre), None


class _BitcostTrainFn(torch.autograd.Function):
    """pc.bitcost(stop_gradient(q), symbols, is_training=True): backward = context-model backward (its bucket first)."""

    @staticmethod
    def forward(ctx, graph, q, symbols, pad_value, hook):
        ctx.graph = graph
        return graph.bitcost_train(q, symbols, pad_value)

    @staticmethod
    def backward(ctx, d_bc):
        ctx.graph.backward_pc(d_bc.contiguous())
        return None, None, None, None, None


class Distortions(object):
    """train.py:352-431.  x, x_out: (N,3,H,W) float 0..255.  mse / psnr on integer-cast values unless they are the quantity
    being minimised in training (:361-366); ms_ssim only when it is minimised (:359)."""

    def __init__(self, config, x, x_out, is_training):
        self.config = config
        minimize_for = config.distortion_to_minimize
        assert minimize_for in ('mse', 'psnr', 'ms_ssim')
        cast_psnr = (not is_training) or minimize_for != 'psnr'
        cast_mse = (not is_training) or minimize_for != 'mse'
        self.mse = self.get_mse_per_img(x, x_out, cast_mse).mean()
        self.psnr = self.get_psnr_per_image(x, x_out, cast_psnr).mean()
        self.ms_ssim = ms_ssim.multiscale_ssim(x, x_out) if minimize_for == 'ms_ssim' else None
        self.d_loss_scaled = self._get_distortion_to_minimize(minimize_for)

    def _get_distortion_to_minimize(self, minimize_for):
        if minimize_for == 'mse':
            return self.mse
        if minimize_for == 'psnr':
            return float(self.config.K_psnr) - self.psnr
        return float(self.config.K_ms_ssim) * (1.0 - self.ms_ssim)

    @staticmethod
    def get_mse_per_img(inp, otp, cast_to_int):
        if cast_to_int:
            inp, otp = inp.detach().to(torch.int32), otp.detach().to(torch.int32)      # tf.cast truncates
        return ((otp - inp) ** 2).to(torch.float32).mean(dim=(1, 2, 3))

    @staticmethod
    def get_psnr_per_image(inp, otp, cast_to_int):
        return 10.0 * torch.log10(255.0 * 255.0 / Distortions.get_mse_per_img(inp, otp, cast_to_int))


class _HipMsSsimDistortion(object):
    """Distortions(config, x, x_out, is_training=True) for distortion_to_minimize = ms_ssim (train.py:352-394) and
    d(d_loss_scaled) / d(x_out), from csrc/msssim.hip: the blur matrices of the shape are made once on the host
    (ic_msssim_plan_fill) and uploaded; a call is ~16 launches on the current stream, no host synchronisation."""

    def __init__(self, config, shape, device):
        N, C, H, W = (int(v) for v in shape)
        nb = int(lib.ic_msssim_plan_bytes(H, W))
        if nb == 0:
            raise NotImplementedError('MS-SSIM needs five scales: {} x {} is too small'.format(H, W))
        host = torch.empty(nb, dtype=torch.uint8)
        check(lib.ic_msssim_plan_fill(H, W, host.data_ptr(), nb), 'ic_msssim_plan_fill')
        self.plan = host.to(device)
        self.ws_bytes = int(lib.ic_msssim_workspace_bytes(N, C, H, W))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)
        self.shape, self.K, self.config, self.dev = (N, C, H, W), float(config.K_ms_ssim), config, torch.device(device)

    def launch(self, x, x_out, want_grad=True):
        """-> values with .ms_ssim, .d_loss_scaled (0-d device tensors) and .grad = d(d_loss_scaled)/d(x_out)"""
        N, C, H, W = self.shape
        assert tuple(x.shape) == self.shape and tuple(x_out.shape) == self.shape
        x, x_out = x.contiguous(), x_out.contiguous()
        grad = torch.empty_like(x_out) if want_grad else None
        scal = torch.empty(16, dtype=torch.float32, device=self.dev)
        check(lib.ic_msssim_loss_grad_f32(ptr(x), ptr(x_out), N, C, H, W, self.K, ptr(self.plan), ptr(grad), ptr(scal), ptr(self.ws),
                                          self.ws_bytes, _lib.current_stream(self.dev)), 'ic_msssim_loss_grad_f32')
        d = _DistortionValues()
        d.ms_ssim, d.d_loss_scaled, d.grad, d.scalars = scal[0], scal[1], grad, scal
        d.mse = d.psnr = None
        return d

    def __call__(self, x, x_out):
        """the autograd form (the plugin call sites compose the loss themselves): d_loss_scaled carries the gradient to x_out"""
        d = _DistortionValues()
        d.d_loss_scaled, d.ms_ssim = _HipDistortionFn.apply(self, x, x_out)
        with torch.no_grad():                      # logged only (train.py:361-366: integer-cast values when not minimised)
            d.mse = Distortions.get_mse_per_img(x, x_out, True).mean()
            d.psnr = Distortions.get_psnr_per_image(x, x_out, True).mean()
        return d


class _HipDistortionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hd, x, x_out):
        v = hd.launch(x, x_out, want_grad=True)
        ctx.save_for_backward(v.grad)
        loss, ms = v.d_loss_scaled.clone(), v.ms_ssim.clone()
        ctx.mark_non_differentiable(ms)              # the RETURNED tensor: ms_ssim is a logged value, d_loss_scaled carries the gradient
        return loss, ms

    @staticmethod
    def backward(ctx, go, _unused):
        grad, = ctx.saved_tensors
        if go is None:                               # only ms_ssim was used downstream: nothing flows to x_out
            return None, None, None
        return None, None, grad * go


class _DistortionValues(object):
    pass


def get_loss(config, ae, pc, d_loss_scaled, bc, heatmap):
    """train.py:303-336.  ae / pc: plugin objects for the regularisation VALUES (None: leave them out -- their gradients
    are folded into the filter-gradient kernels either way).  -> total_loss, H_real, pc_comps, ae_comps"""
    assert config.H_target
    bc_mask = bc * heatmap if heatmap is not None else bc
    H_real = bc.mean()
    H_mask = bc_mask.mean()
    H_soft = 0.5 * (H_mask + H_real)
    pc_loss = float(config.beta) * torch.clamp(H_soft - float(config.H_target), min=0.0)
    zero = torch.zeros((), device=bc.device)
    reg_pc = pc.regularization_loss() if pc is not None else None
    reg_pc = zero if reg_pc is None else reg_pc
    reg_enc = ae.encoder_regularization_loss() if ae is not None else zero
    reg_dec = ae.decoder_regularization_loss() if ae is not None else zero
    pc_comps = [('H_mask', H_mask), ('H_real', H_real), ('pc_loss', pc_loss), ('reg', reg_pc)]
    ae_comps = [('d_loss_scaled', d_loss_scaled), ('reg_enc_dec', reg_enc + reg_dec)]
    total_loss = d_loss_scaled + pc_loss + (reg_pc + reg_enc + reg_dec).detach()
    return total_loss, H_real, pc_comps, ae_comps


class TFAdam(object):
    """tf.train.AdamOptimizer (train.py:339-349 via training_helpers.py:38-48): beta1 0.9, beta2 0.999, eps 1e-8,
    lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t); var -= lr_t * m / (sqrt(v) + eps)  (epsilon outside the bias
    correction, unlike torch.optim.Adam)."""
    kind = 'ADAM'

    def __init__(self, params, grads, lr, beta1=0.9, beta2=0.999, eps=1e-8, flat=None):
        """flat: [(flat_params, flat_grads), ...] -- buffers that params / grads are views of, in one common layout (TrainGraph's
        buckets).  The step is then ONE fused launch per buffer pair (ic_adam_tf_f32) and m / v are views of flat slot buffers;
        without it, seven multi-tensor passes over the tensor lists."""
        self.params, self.grads = list(params), list(grads)
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.flat = None
        if flat:
            self.flat = [(fp, fg, torch.zeros_like(fp), torch.zeros_like(fp)) for fp, fg in flat]

            def slot(p, which):
                for fp, _, fm, fv in self.flat:
                    off = (p.data_ptr() - fp.data_ptr()) // 4
                    if 0 <= off < fp.numel() and p.data_ptr() >= fp.data_ptr():
                        return (fm, fv)[which][off:off + p.numel()].view(p.shape)
                raise ValueError('a parameter is not a view of the flat buffers')
            self.m = [slot(p, 0) for p in self.params]
            self.v = [slot(p, 1) for p in self.params]
        else:
            self.m = [torch.zeros_like(p) for p in self.params]
            self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    def slot_tensors(self, params):
        return [self.m, self.v]

    def step(self, lr=None):
        self.t += 1
        lr = self.lr if lr is None else lr
        lr_t = lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        if self.flat:
            for fp, fg, fm, fv in self.flat:
                check(lib.ic_adam_tf_f32(ptr(fp), ptr(fg), ptr(fm), ptr(fv), fp.numel(), lr_t, self.b1, self.b2, self.eps,
                                         _lib.current_stream(fp.device)), 'ic_adam_tf_f32')
            return
        torch._foreach_mul_(self.m, self.b1)
        torch._foreach_add_(self.m, self.grads, alpha=1.0 - self.b1)
        torch._foreach_mul_(self.v, self.b2)
        torch._foreach_addcmul_(self.v, self.grads, self.grads, value=1.0 - self.b2)
        denom = torch._foreach_sqrt(self.v)
        torch._foreach_add_(denom, self.eps)
        torch._foreach_addcdiv_(self.params, self.m, denom, value=-lr_t)


class TFGradientDescent(object):
    """tf.train.GradientDescentOptimizer (training_helpers.py:42-48, optimizer = SGD): var -= lr * grad.  No slots."""
    kind = 'SGD'

    def __init__(self, params, grads, lr, flat=None):
        self.lr = lr
        self.pairs = [(fp, fg) for fp, fg in flat] if flat else list(zip(params, grads))
        self.t = 0

    def step(self, lr=None):
        self.t += 1
        lr = self.lr if lr is None else lr
        for p_, g_ in self.pairs:
            p_.add_(g_, alpha=-float(lr))

    def slot_tensors(self, params):
        return []


class TFMomentum(object):
    """tf.train.MomentumOptimizer(momentum=config.optimizer_momentum, use_nesterov=True) (training_helpers.py:46-47):
    accum = momentum * accum + grad;  var -= lr * grad + lr * momentum * accum   (TF's Nesterov form of ApplyMomentum)."""
    kind = 'MOMENTUM'

    def __init__(self, params, grads, lr, momentum=0.9, flat=None):
        self.lr, self.momentum = lr, float(momentum)
        self.params = list(params)
        self.pairs = [(fp, fg) for fp, fg in flat] if flat else list(zip(params, grads))
        self.acc = [torch.zeros_like(p_) for p_, _ in self.pairs]
        self.t = 0

    def step(self, lr=None):
        self.t += 1
        lr = float(self.lr if lr is None else lr)
        for (p_, g_), a_ in zip(self.pairs, self.acc):
            a_.mul_(self.momentum).add_(g_)
            p_.add_(g_, alpha=-lr).add_(a_, alpha=-lr * self.momentum)

    def _view(self, p_):
        for (fp, _), a_ in zip(self.pairs, self.acc):
            off = (p_.data_ptr() - fp.data_ptr()) // 4
            if p_.data_ptr() >= fp.data_ptr() and 0 <= off < fp.numel():
                return a_.view(-1)[off:off + p_.numel()].view(p_.shape)
        raise ValueError('a parameter is not a view of the optimiser\'s buffers')

    def slot_tensors(self, params):
        """[(checkpoint suffix index, tensors)]: the one slot `momentum` -> `<var>/<optimizer name>`"""
        return [[self._view(p_) for p_ in params]]


def create_optimizer(config, params, grads, flat):
    """training_helpers.create_optimizer / optimizer_cls (:38-48): config.optimizer in ADAM | SGD | MOMENTUM (Nesterov, momentum =
    config.optimizer_momentum).  The learning rate arrives per step (create_learning_rate_tensor -> learning_rate())."""
    kind = str(getattr(config, 'optimizer', 'ADAM'))
    lr = float(config.lr_initial)
    if kind == 'ADAM':
        return TFAdam(params, grads, lr, flat=flat)
    if kind == 'SGD':
        return TFGradientDescent(params, grads, lr, flat=flat)
    if kind == 'MOMENTUM':
        return TFMomentum(params, grads, lr, momentum=float(config.optimizer_momentum), flat=flat)
    raise ValueError('Invalid optimizer {} (ADAM, SGD, MOMENTUM)'.format(kind))


def get_num_itr_per_epoch(num_images, batch_size, num_crops_per_img):
    """training_helpers.py:51-60: the input pipeline cuts num_crops_per_img crops out of every decoded image, so a batch
    holds batch_size // num_crops_per_img unique images and an epoch is num_images // that many iterations (batch 30 with 8
    crops: num_images // 3).  batch_size is the GLOBAL batch of the config, whatever the number of ranks."""
    return max(int(num_images) // max(int(batch_size) // int(num_crops_per_img), 1), 1)


def learning_rate(config, step, num_itr_per_epoch):
    """training_helpers.py:22-34: FIXED, or staircase exponential decay every decay_interval epochs."""
    lr = float(config.lr_initial)
    if config.lr_schedule == 'FIXED':
        return lr
    decay_steps = max(int(num_itr_per_epoch * config.lr_schedule_decay_interval), 1)
    e = step / decay_steps
    if config.lr_schedule_decay_staircase:
        e = math.floor(e)
    return lr * float(config.lr_schedule_decay_rate) ** e


class Trainer(object):
    """TrainGraph + the two Adam optimisers of get_train_op (AE variables with lr_ae, context model with lr_pc)."""

    def __init__(self, ae_config, pc_config, weights, device='cuda', num_itr_per_epoch=1000, process_group=None, sync_bn=None, wino4=None):
        self.graph = TrainGraph(ae_config, pc_config, weights, device, process_group, sync_bn=sync_bn, wino4=wino4)
        g = self.graph
        ae_names = g.group_names['enc'] + g.group_names['dec']
        pc_names = g.group_names['pc']
        self._ae_names, self._pc_names = ae_names, pc_names
        # get_train_op (train.py:339-349): create_optimizer(ae_config, lr_ae, 'Adam_AE') for the autoencoder's variables,
        # create_optimizer(pc_config, lr_pc, 'Adam_PC') for the context model's -- each config picks its own optimiser class
        self.opt_ae = create_optimizer(ae_config, [g.trainable[n] for n in ae_names], [g.grads[n] for n in ae_names],
                                       flat=[(g.flat_params[k], g.flat_grads[k]) for k in ('enc', 'dec')])
        self.opt_pc = create_optimizer(pc_config, [g.trainable[n] for n in pc_names], [g.grads[n] for n in pc_names],
                                       flat=[(g.flat_params['pc'], g.flat_grads['pc'])])
        self.num_itr_per_epoch = num_itr_per_epoch
        self.global_step = 0

    def apply_gradients(self):
        """get_train_op (train.py:339-349): the two Adam updates on the gradients of the last backward"""
        g = self.graph
        if g.ae_config.train_autoencoder:
            self.opt_ae.step(learning_rate(g.ae_config, self.global_step, self.num_itr_per_epoch))
        if g.ae_config.train_probclass:
            self.opt_pc.step(learning_rate(g.pc_config, self.global_step, self.num_itr_per_epoch))
        self.global_step += 1
        g.version += 1
        g.refresh_pad_value()

    def step(self, x):
        g = self.graph
        out = g.forward_backward(x)
        self.apply_gradients()
        return out

    def state_weights(self, training_state=True):
        """current variables as a checkpoint-style dict name -> numpy array; training_state adds what the reference's Saver
        also writes: global_step and the slots of the two Adam optimisers (get_train_op, train.py:339-349: optimiser names
        Adam_AE / Adam_PC -> slot variables `<var>/Adam_AE`, `<var>/Adam_AE_1`, and the beta powers)."""
        out = OrderedDict((n, t.detach().cpu().numpy()) for n, t in self.graph.params.items())
        if training_state:
            out['global_step'] = np.array(self.global_step, np.int64)
            for opt, tag, names in ((self.opt_ae, 'Adam_AE', self._ae_names), (self.opt_pc, 'Adam_PC', self._pc_names)):
                # slot variables are named after the optimiser's NAME (`Adam_AE` whatever its class, train.py:341,344), uniquified
                # `_1` for the second slot: Adam m, v -> `<var>/Adam_AE`, `<var>/Adam_AE_1`; Momentum's one slot -> `<var>/Adam_AE`
                params = [self.graph.trainable[n] for n in names]
                for si, tensors in enumerate(opt.slot_tensors(params)):
                    for n, t_ in zip(names, tensors):
                        out['{}/{}{}'.format(n, tag, '_1' if si else '')] = t_.detach().cpu().numpy()
                if opt.kind == 'ADAM':
                    # TF-1.x: `beta1_power` / `beta2_power` at the graph root, uniquified per optimiser in creation order
                    # (get_train_op creates Adam_AE first, train.py:339-349).  TF initialises beta1_power to beta1 and multiplies it
                    # once per update (_finish), so after t updates the checkpoint holds beta ** (t + 1)
                    b1n, b2n = BETA_POWER_NAMES[tag]
                    out[b1n] = np.array(opt.b1 ** (opt.t + 1), np.float32)
                    out[b2n] = np.array(opt.b2 ** (opt.t + 1), np.float32)
        return out

    def _adam_steps_from_checkpoint(self, ckpt, tag, opt):
        """updates already applied, from the beta powers TF keeps: the checkpoint holds beta ** (t + 1).  beta2_power is preferred
        (0.999 ** t is a normal float32 until t ~ 87,000; 0.9 ** t goes denormal near t = 830 and reaches zero near 980); beyond
        what either can resolve the global step counts the updates."""
        def read(name, legacy):
            # rounds 1-3 of this repo wrote `<optimiser>/beta1_power` holding beta ** t (no TF checkpoint has that name): no -1 there
            if name in ckpt:
                return float(np.asarray(ckpt[name]).reshape(-1)[0]), 1
            if legacy in ckpt:
                return float(np.asarray(ckpt[legacy]).reshape(-1)[0]), 0
            return None, 0
        b1n, b2n = BETA_POWER_NAMES[tag]
        for (val, minus), base in ((read(b2n, tag + '/beta2_power'), opt.b2), (read(b1n, tag + '/beta1_power'), opt.b1)):
            # a NORMAL float32 carries beta ** (t + 1) to 6e-8 relative, i.e. t to 6e-8 / |log beta| << 1: exact
            if val is not None and 1e-30 < val < 1.0:
                t = int(round(math.log(val) / math.log(base))) - minus
                if t >= 0:
                    return t
            elif val == 1.0 and not minus:
                return 0                                   # legacy checkpoint at step 0: beta ** 0
        return self.global_step

    def restore_training_state(self, ckpt):
        """global_step, Adam moments and step counts from a checkpoint dict (whatever of it is present): the DECAY schedule
        continues where the run stopped instead of restarting at lr_initial, and checkpoints keep counting from there."""
        if 'global_step' in ckpt:
            self.global_step = int(np.asarray(ckpt['global_step']).reshape(-1)[0])
        dev = self.graph.dev
        for opt, tag, names in ((self.opt_ae, 'Adam_AE', self._ae_names), (self.opt_pc, 'Adam_PC', self._pc_names)):
            found = 0
            slots = opt.slot_tensors([self.graph.trainable[n] for n in names])
            for i, n in enumerate(names):
                keys = ['{}/{}{}'.format(n, tag, '_1' if si else '') for si in range(len(slots))]
                if keys and all(k_ in ckpt for k_ in keys):
                    for k_, tensors in zip(keys, slots):
                        tensors[i].copy_(torch.as_tensor(np.asarray(ckpt[k_]), dtype=torch.float32).to(dev).view_as(tensors[i]))
                    found += 1
            if found and opt.kind == 'ADAM':
                opt.t = self._adam_steps_from_checkpoint(ckpt, tag, opt)
            elif found:
                opt.t = self.global_step
        return self.global_step
