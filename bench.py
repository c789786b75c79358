#!/usr/bin/env python
"""bench.py -- throughput of the imgcomp hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input: BASELINE.json configs[1]
(Kodak-shaped image 1x3x512x768, ae_configs/cvpr/low + pc_configs/cvpr/res_shallow, batch 1):
    encode (normalise, 35 convs, importance map, quantiser) -> context-model bit cost for all symbols
    in parallel + bpp -> decode(qhard) (35 convs, de-normalise, clip)          [val.py:85-89 wiring]
Inputs and weights are resident in HBM before the timed region.  Data: seeded synthetic images and
random-init weights (no network for Kodak or the 0515_1103 checkpoint).

Schedule (--in_flight n, default 4): the images of an evaluation set are independent (val.py:157-158 runs one per sess.run), so
n of them are in flight at a time, each a batch-1 step on its own stream with its own network objects and workspace; steps are
issued round-robin and EVERY step is still one image through the whole path.  The launches of one image fill the kernel-boundary
bubbles of the others, and a 3x3 launch no longer has to fill the chip alone (IC_CONV3_IN_FLIGHT: the plan takes the form with
the least CU-time).  --in_flight 1 is one image at a time (what rounds 1-2 reported; kept in the line as `one_image_at_a_time`).
The HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues, 4 unless set: with more streams than
queues, streams that share a queue run one after the other.  bench.py asks for 8 below, before the runtime starts (round 4:
4 images in flight 207.7 Mpix/s on 4 queues, 251.5 on 8; 6 images 236 / 240).

  python bench.py --gpus N --steps K --warmup W            (--mode train: one cfg3 training step per step)
N > 1 is launched by torch.distributed.run, one rank per GPU; the path shards by image (independent
units, no data-path collective), so every rank runs the same per-GPU workload: weak scaling.

Prints ONE JSON line on rank 0 (see the contract in the task statement) with extra objects:
  roofline               -- the dominant kernel (3x3 128->128 conv on the fp32 matrix cores), timed IN-STEP: HIP events around
                            the 32-layer residual stack of the encoder / of the decoder on the step's own activations, n stacks
                            in flight like the step, / launches.  `achieved` = FLOPs the matrix pipe executes (Winograd:
                            16/36 of the direct form's) / that time; `frac` = achieved / 157.3 TFLOP/s, <= 1 by
                            construction; `alone` = one launch at a time (a kernel trace's duration of the launch)
  roofline_context_model -- the same for the masked-3D-conv context model, standalone
  layers_5x5             -- the six 5x5 / stride-2 layers, each alone, against both roofs
  shapes                 -- the north_star's 256x256 shape (batch 1 and 8) through the same step
  cpu_baseline           -- the CPU oracle (torch fp32 restatement of the reference) timed on a bounded sample of the same
                            workload (rank 0, N = 1 only)
"""
import argparse
import ctypes
import json
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')      # one hardware queue per image in flight (+ the default stream); see the docstring

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
# algorithmic work per input pixel, C = 32 (FLOP = 2 MAC, dense, mask-agnostic, halo-free)
FLOP_PER_PX_ENC = 621124.0
FLOP_PER_PX_DEC = 618976.0
FLOP_PER_SYMBOL_PC = 47520.0             # dense; 36,912 counting only the live taps of the causal masks
FLOP_PER_SYMBOL_PC_LIVE = 36912.0
CONV3_FLOP_PER_OUT_PX = 2.0 * 9 * 128 * 128      # per 128-channel output pixel of one 3x3 layer, direct form

FORM_NAMES = {0: 'automatic'}


def conv3_scopes(W, ae_cfg, which):
    root = W.ENC if which == 'enc' else W.DEC
    return [s for s, kind, shape in W.ae_conv_specs(int(ae_cfg.num_chan_bn), int(ae_cfg.arch_param_B), bool(ae_cfg.heatmap))
            if s.startswith(root) and tuple(shape) == (3, 3, 128, 128)]


class Pipeline(object):
    """encode -> (bitcost on the side stream || decode) of one batch, the val.py wiring."""

    def __init__(self, dev, ae_config='low', share='cu_range', seed=0, idle_layers=None, share_weights_with=None):
        import torch
        from imgcomp_cvpr_amd import autoencoder, probclass, config_parser as cp, weights as W, streams
        self.torch, self.dev, self.W = torch, dev, W
        self.ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', ae_config))
        self.pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
        if share_weights_with is not None:
            # ONE model evaluates the images of a set (val.py:74-89 builds the graph once): the pipelines of the images in flight read
            # the same device weights and packed filters (val.Fetcher(share_with=...) does the same), own workspaces and outputs each
            o = share_weights_with
            self.wts, self.ae, self.pc = o.wts, o.ae.sharing_weights(), o.pc.sharing_weights()
        else:
            self.wts = W.synthetic_weights(self.ae_cfg, self.pc_cfg)
            self.ae = autoencoder.get_network_cls(self.ae_cfg)(self.ae_cfg).load_weights(self.wts, dev)
            self.pc = probclass.get_network_cls(self.pc_cfg)(self.pc_cfg, num_centers=self.ae_cfg.num_centers).load_weights(self.wts, dev)
        self.pad_value = float(self.wts['autoencoder/encoder/centers'][0])
        self.serial = share in ('serial', 'auto')    # bitcost, then decode, on one stream
        self.branch = streams.BranchStreams(dev, share=share, idle_layers=idle_layers)
        self.seed = seed

    def set_input(self, N, H, Wd):
        self.N, self.H, self.Wd = N, H, Wd
        self.x_np = self.W.synthetic_image((N, 3, H, Wd), 'natural', seed=self.seed)
        self.x = self.torch.as_tensor(self.x_np).float().to(self.dev)
        self.side = self.branch.context_model_stream(N, H, Wd, int(self.ae_cfg.num_chan_bn))
        self.dec_flags = self.branch.decode_flags(self.side)
        return self

    def capture(self):
        """record one step of this pipeline into a HIP graph (static input, static outputs): a step is then ONE replay -- the
        ~150 launches of an image cost the host one call instead of ~0.35 ms of enqueueing"""
        torch = self.torch
        for _ in range(2):
            self._step_eager()                         # first-use allocations happen outside the capture
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._graph_out = self._step_eager()
        self._graph = g
        return self

    def step(self):
        if getattr(self, '_graph', None) is not None:
            self._graph.replay()
            return self._graph_out
        return self._step_eager()

    def _step_eager(self):
        from imgcomp_cvpr_amd import bits
        torch = self.torch
        cur = torch.cuda.current_stream(self.dev)
        enc = self.ae.encode(self.x, is_training=False)
        if self.serial:
            bc = self.pc.bitcost(enc.qbar, enc.symbols, is_training=False, pad_value=self.pad_value)
            self.last = (enc, bc)       # references only (tests/test_gpu_bench.py compares z / symbols / bits of this very schedule)
            return bits.bitcost_to_bpp(bc, self.x), self.ae.decode(enc.qhard, is_training=False)
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            bc = self.pc.bitcost(enc.qbar, enc.symbols, is_training=False, pad_value=self.pad_value)
            bpp = bits.bitcost_to_bpp(bc, self.x)
        self.last = (enc, bc)
        # per-call plan flag: next to a CU-range side stream the decoder's 3x3 launches leave that stream's CUs alone
        x_out = self.ae.decode(enc.qhard, is_training=False, plan_flags=self.dec_flags)
        cur.wait_stream(self.side)
        return bpp, x_out


class InFlight(object):
    """n independent batch-1 pipelines -- own network objects, workspaces, input image and stream each; step i runs on pipeline
    i % n.  The images of an evaluation set are independent (val.py:157-158 loops over them), so the launches of one image fill
    the kernel-boundary bubbles of the others -- and a launch no longer has to fill the chip alone, so the 3x3 layers run the
    form with the least CU-time (IC_CONV3_IN_FLIGHT).  Every step is still ONE image through the whole path."""

    def __init__(self, torch, first, dev, n, ae_config, seed0, graphs=False):
        self.torch, self.n, self.i = torch, n, 0
        from imgcomp_cvpr_amd import _lib
        shared = os.environ.get('IMGCOMP_BENCH_OWN_WEIGHTS', '0') != '1'       # A/B switch: 1 = every pipeline uploads its own copy (rounds 3-4)
        self.pipes = []
        for k in range(n):
            self.pipes.append(Pipeline(dev, ae_config, 'serial', seed=seed0 + 1000 * k,
                                       share_weights_with=self.pipes[0] if (shared and k) else None).set_input(first.N, first.H, first.Wd))
        for pl in self.pipes:
            # the plan hint: n independent calls of this shape are in flight (IC_CONV3_IN_FLIGHT, include/imgcomp_hip.h)
            pl.ae.plan_flags = first.ae.plan_flags | _lib.CONV3_IN_FLIGHT(n)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
        # set-up, not warm-up: every pipeline allocates its workspaces and output buffers on its first pass
        for pl, s_ in zip(self.pipes, self.streams):
            with torch.cuda.stream(s_):
                pl.step()
        torch.cuda.synchronize(dev)
        if graphs:
            for pl in self.pipes:
                pl.capture()

    def step(self):
        k = self.i % self.n
        self.i += 1
        with self.torch.cuda.stream(self.streams[k]):
            return self.pipes[k].step()


def flight_for_shape(lib, n, h, w, n_default):
    """images in flight for a shape.  VERDICT r5 item 3a asked for enough of them that the 3x3 launches in flight hold >= 512 work-groups
    (8 - 16 at 256 x 256).  Measured in round 6 (tools/flight_sweep.py, Mpix/s of the whole step, n = 1 / 2 / 4 / 6 / 8 / 12 / 15):
        256 x 256  automatic plan  63.8 / 97.1 / 131.5 / 90.6 / 94.8 / 89.6 / 110.4   F(4x4) forced  30.4 / 55.1 / 90.1 / 63.8 / 84.6 / 73.1 / 80.9
        384 x 512  automatic plan 110.7 / 149.1 / 222.2 / 181.1 / 180.3 / 230.2 / 235.2
    More than four streams do not fill the chip better -- they share the hardware queues unevenly (the non-monotonic rows) and a 256 x 256
    launch is bounded by its own 13 us chain, not by the CUs it leaves idle.  So: the benchmark's own count for
    every shape; small images are batched instead (val.py --batch_same_shape: 8 x 256 x 256 per step, 275 Mpix/s)."""
    return int(n_default)


def res_stack_runner(torch, lib, _lib, W, ae, ae_cfg, pipe, enc, which, flags, st):
    """the 32-layer residual stack of the encoder / decoder with its own 32 filters through the library's own launch sequence
    (ic_ae_res_stack_f32 = the res_stack of network.hip that encode / decode run) on the stack's REAL input: the first buffer of the
    autoencoder's workspace still holds it after a call (kept for the global skip) -- h2's output after encode, from_bn's after
    decode.  (Random data would not do: the chip clocks to its power budget, and power follows the data.)  -> (go(stream=None), layers)"""
    N, H, Wd, dev = pipe.N, pipe.H, pipe.Wd, pipe.dev
    h4, w4 = H // 4, Wd // 4
    n4 = N * 128 * h4 * w4
    rs_need = lib.ic_ae_res_stack_workspace_bytes(N, h4, w4)
    if which == 'enc':
        ae.encode(pipe.x, False)
    else:
        ae.decode(enc.qhard, False)
    xin = ae._ws.view(torch.float32)[:n4].clone()
    yo = torch.empty((N, 128, h4, w4), device=dev)
    ws_ = torch.empty(rs_need, dtype=torch.uint8, device=dev)
    tens = []
    for sname in conv3_scopes(W, ae_cfg, which):
        tens += list(ae._plan[sname])
    tab = _lib.ptr_table(tens)
    B = int(ae_cfg.arch_param_B)

    def go(sth=None):
        _lib.check(lib.ic_ae_res_stack_f32(_lib.ptr(xin), tab, B, _lib.ptr(yo), N, h4, w4, _lib.ptr(ws_), rs_need, flags,
                                           st if sth is None else sth))
    go.keep = (xin, yo, ws_, tens, tab)
    go.out = yo
    return go, len(tens) // 3


def timed_concurrent_stacks(torch, lib, _lib, dev, ev, st, gos, reps, warm=2):
    """gos[i] runs on its own stream; HIP events on the main stream bracket all of them -> ms per stack"""
    main = torch.cuda.current_stream(dev)
    strs = [torch.cuda.Stream(device=dev) for _ in gos]
    handles = [ctypes.c_void_p(s_.cuda_stream) for s_ in strs]

    def burst(n):
        for s_ in strs:
            s_.wait_stream(main)
        for _ in range(n):
            for g_, h_ in zip(gos, handles):
                g_(h_)
        for s_ in strs:
            main.wait_stream(s_)
    burst(warm)
    torch.cuda.synchronize(dev)
    _lib.check(lib.ic_event_record(ev[0], st))
    burst(reps)
    _lib.check(lib.ic_event_record(ev[1], st))
    ms = ctypes.c_float()
    _lib.check(lib.ic_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms)))
    return ms.value / (reps * len(gos))


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=50)
    p.add_argument('--warmup', type=int, default=5)
    p.add_argument('--height', type=int, default=512)
    p.add_argument('--width', type=int, default=768)
    p.add_argument('--batch', type=int, default=1)
    p.add_argument('--ae_config', default='low')
    p.add_argument('--mode', default='infer', choices=['infer', 'train'])
    p.add_argument('--share', default='auto', choices=['auto', 'cu_range', 'full_chip', 'serial'],
                   help='how decoder and context model share the chip (imgcomp_cvpr_amd/streams.py); auto = the package default')
    p.add_argument('--idle_layers', type=int, default=None, help="cu_range sharing: 3x3 launches of the decoder that leave the side stream's CUs idle (0 = all; default: sized from the context model's work)")
    p.add_argument('--no_cpu_baseline', action='store_true')
    p.add_argument('--no_extras', action='store_true', help='headline only: no stage split, roofline, extra shapes (rocprofv3 runs)')
    p.add_argument('--pipelined', action='store_true', help='also run the informational three-pipelines-in-flight section')
    p.add_argument('--in_flight', type=int, default=4,
                   help='independent batch-1 images in flight on their own streams (serial arrangement only); 1 = one image at a time')
    p.add_argument('--graphs', type=int, default=0, help='1: every in-flight pipeline replays its step from a captured HIP graph')
    p.add_argument('--calib_copy', action='store_true',
                   help='after the timed steps: a 256 MiB device-to-device copy (rocprofv3 --pmc passes calibrate FETCH_SIZE / WRITE_SIZE on it)')
    p.add_argument('--plan_flags', type=lambda v: int(v, 0), default=0,
                   help='IC_CONV3_* bits OR-ed into every encode / decode call (include/imgcomp_hip.h; 0x80 = IC_CONV3_STACK_KERNEL needs a `make TUNING=1` library)')
    p.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                   help='process-group backend for N > 1 (nccl = RCCL; gloo lets two ranks share one GPU in the tests)')
    p.add_argument('--device', type=int, default=None, help='HIP device index of this rank (default: LOCAL_RANK)')
    p.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                   help='--mode train: weak = 32 crops per GPU; strong = cfg3\'s batch of 32 split over the ranks (train.py:150-153)')
    a = p.parse_args()

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` by itself: become the launcher -- the line INTEGRATION.md section 3 prints, one rank per GPU
        return self_launch(a.gpus)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    assert world == a.gpus, 'WORLD_SIZE {} != --gpus {}'.format(world, a.gpus)
    assert torch.cuda.is_available(), 'bench.py needs a HIP device'
    dev_index = local_rank if a.device is None else a.device
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        if a.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group('gloo')

    if a.mode == 'train':
        return train_main(a, dev, rank, world)

    from imgcomp_cvpr_amd import weights as W, _lib, streams
    lib = _lib.lib
    share = streams.DEFAULT_SHARE if a.share == 'auto' else a.share        # 'auto' = the library default (streams.py: 'serial')
    pipe = Pipeline(dev, a.ae_config, share, seed=rank, idle_layers=a.idle_layers).set_input(a.batch, a.height, a.width)
    pipe.ae.plan_flags = a.plan_flags
    ae, pc, ae_cfg = pipe.ae, pipe.pc, pipe.ae_cfg
    N, H, Wd = a.batch, a.height, a.width
    torch.cuda.synchronize(dev)
    torch.cuda.set_stream(pipe.branch.main)      # CU-range streams are blocking with respect to the legacy default stream

    def barrier(collective):
        # every rank calls run(..., collective=True) the same number of times (once: the contract's timed region); the
        # rank-0-only extras below time with collective=False, i.e. a local synchronize and NO process-group call
        if collective and world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def run(pl, steps, warmup, collective=False):
        for _ in range(warmup):
            pl.step()
        barrier(collective)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = pl.step()
        barrier(collective)
        return time.perf_counter() - t0, out

    n_flight = a.in_flight if (pipe.serial and a.in_flight > 1) else 1
    sched = InFlight(torch, pipe, dev, n_flight, a.ae_config, rank, graphs=bool(a.graphs)) if n_flight > 1 else pipe
    elapsed, (bpp, x_out) = run(sched, a.steps, a.warmup, collective=True)
    elapsed = max_over_ranks(torch, dist, elapsed, dev, world, a.backend)
    ranks_seen = ranks_report(torch, dist, dev, rank, world, a.backend)
    value = N * H * Wd * world * a.steps / elapsed / 1e6
    if a.calib_copy:
        src = torch.randn(64 * 1024 * 1024, device=dev)
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        torch.cuda.synchronize(dev)

    same_stream = pipe.serial or pipe.side is pipe.branch.main
    cus = pipe.branch.idle_cus(N, H, Wd) if not same_stream and pipe.side is not pipe.branch._plain else 0
    extra = {'branch_sharing': share, 'context_model_stream_cus': cus, 'bpp_synthetic': round(float(bpp), 5),
             'images_in_flight': n_flight}
    roofline = roofline_pc = None
    if rank == 0 and not a.no_extras:
        st = _lib.current_stream(dev)
        ev = [ctypes.c_void_p() for _ in range(2)]
        for e in ev:
            _lib.check(lib.ic_event_create(ctypes.byref(e)))

        def timed(fn, reps, warm=2):
            for _ in range(warm):
                fn()
            torch.cuda.synchronize(dev)
            _lib.check(lib.ic_event_record(ev[0], st))
            for _ in range(reps):
                fn()
            _lib.check(lib.ic_event_record(ev[1], st))
            ms = ctypes.c_float()
            _lib.check(lib.ic_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms)))
            return ms.value / reps

        # ---- the same step with ONE image at a time (rounds 1-2 reported this as `value`) ----
        if n_flight > 1:
            dt1, _ = run(pipe, 20, 3)
            extra['one_image_at_a_time'] = {'value': round(N * H * Wd * 20 / dt1 / 1e6, 3), 'unit': 'Mpix/s', 'ms_per_step': round(dt1 / 20 * 1e3, 4)}
        # ---- stage split (each stage alone on the stream) ----
        enc = ae.encode(pipe.x, False)
        ms_enc = timed(lambda: ae.encode(pipe.x, False), 10)
        ms_pc = timed(lambda: pc.bitcost(enc.qbar, enc.symbols, False, pad_value=pipe.pad_value), 20)
        ms_dec = timed(lambda: ae.decode(enc.qhard, False), 10)
        ms_dec_shared = timed(lambda: ae.decode(enc.qhard, False, plan_flags=pipe.dec_flags), 10) if pipe.dec_flags else ms_dec
        extra.update({'ms_encode': round(ms_enc, 4), 'ms_pc_bitcost': round(ms_pc, 4), 'ms_decode': round(ms_dec, 4),
                      'ms_decode_with_step_flags': round(ms_dec_shared, 4),
                      'stage_split_note': 'each stage of ONE image alone on one stream (one-image-at-a-time plan): their sum is the '
                                          'one_image_at_a_time step, not ms_per_step of {} images in flight'.format(n_flight)})

        # ---- dominant kernel, in-step: the 32-layer residual stack with its own 32 filters through the library's own launch
        # sequence (ic_ae_res_stack_f32 = the res_stack of network.hip that encode / decode run), HIP events around it.
        # With n images in flight the step's 3x3 launches overlap across streams, so the stack is timed the same way: n stacks,
        # one per stream, events on the main stream around all of them -> the time in which the chip completes one launch.
        # The same stack alone on one stream (one launch at a time, what a kernel trace shows as the launch's duration) is
        # reported beside it. ----
        h4, w4 = H // 4, Wd // 4
        n4 = N * 128 * h4 * w4
        rs_need = lib.ic_ae_res_stack_workspace_bytes(N, h4, w4)
        step_flags = a.plan_flags | (_lib.CONV3_IN_FLIGHT(n_flight) if n_flight > 1 else 0)

        def res_stack(which, flags):
            return res_stack_runner(torch, lib, _lib, W, ae, ae_cfg, pipe, enc, which, flags, st)

        def timed_concurrent(gos, reps, warm=2):
            return timed_concurrent_stacks(torch, lib, _lib, dev, ev, st, gos, reps, warm)

        def layer_entry(which, flags):
            go, nl = res_stack(which, flags)
            ms_alone = timed(go, 10) / nl
            if n_flight > 1:
                gos = [go] + [res_stack(which, flags)[0] for _ in range(n_flight - 1)]
                ms = timed_concurrent(gos, 6) / nl
            else:
                ms = ms_alone
            flop_direct = CONV3_FLOP_PER_OUT_PX * N * h4 * w4
            form = lib.ic_conv3x3_c128_pick_form(N, h4, w4, flags)
            executed = flop_direct * {0: 1.0, 1: 16.0 / 36.0, 2: 36.0 / 144.0}[form]
            plan = plan_name(lib, _lib, N, h4, w4, flags)
            ent = {'avg_launch_us': round(ms * 1e3, 2), 'layers_timed': nl, 'stacks_in_flight': n_flight,
                   'achieved': round(executed / (ms * 1e-3) / 1e12, 2),
                   'frac': round(executed / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                   'direct_equivalent_tflops': round(flop_direct / (ms * 1e-3) / 1e12, 2),
                   'direct_equivalent_frac': round(flop_direct / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                   'executed_flop_per_launch': executed, 'plan': plan,
                   # one launch at a time: the duration a kernel trace shows for the launch, on the CUs it occupies
                   'alone': {'avg_launch_us': round(ms_alone * 1e3, 2), 'cus': plan['cus'],
                             'frac_of_chip': round(executed / (ms_alone * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                             'frac_of_occupied_cus': round(executed / (ms_alone * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS * 256.0 / max(plan['cus'], 1), 4)}}
            return ent, go

        enc_l, go_enc = layer_entry('enc', step_flags)
        dec_l, go_dec = layer_entry('dec', step_flags | pipe.dec_flags)
        n_idle = (pipe.dec_flags >> 12) & 0x7f
        if pipe.dec_flags & _lib.CONV3_LEAVE_IDLE_CUS and 0 < n_idle < dec_l['layers_timed']:
            # mixed stack: the first n_idle launches keep off the context model's CUs, the rest take the whole chip
            dec_l['plan'] = {'first_launches': n_idle, 'first': dec_l['plan'],
                             'remaining_launches': dec_l['layers_timed'] - n_idle, 'remaining': plan_name(lib, _lib, N, h4, w4, 0)}
        extra['decoder_idle_layers'] = n_idle if pipe.dec_flags & _lib.CONV3_LEAVE_IDLE_CUS else None
        # PMC counters cannot be read from inside this process: rocprofv3 --pmc passes over `bench.py --no_extras --calib_copy`
        # (tools/profile_round.sh) are digested into profiles/rNN_counters.json, keyed by kernel name.  The numbers are only
        # quoted when that file describes THIS step: same input shape and the same 3x3 kernel as the plan that just ran.
        traffic = traffic_src = None
        counters = load_counters(ROOT)
        k3 = enc_l['plan']['kernel']
        if counters and counters.get('input_shape') == [N, 3, H, Wd] and k3 in counters.get('kernels', {}) \
                and counters.get('plan_3x3') == k3:
            ent = counters['kernels'][k3]
            traffic, traffic_src = ent.get('hbm_bytes_per_launch'), counters.get('source')
            extra['conv3x3_counters'] = {k: ent.get(k) for k in ('l2_to_l1_bytes_per_launch', 'valu_per_mfma', 'mfma_busy_us', 'avg_us_rocprof',
                                                                   'mfma_busy_share', 'hbm_read_bytes_per_launch', 'hbm_write_bytes_per_launch')}
        elif counters:
            traffic_src = 'dropped: {} describes shape {} / 3x3 kernel {}, this run is {} / {}'.format(
                counters.get('file'), counters.get('input_shape'), counters.get('plan_3x3'), [N, 3, H, Wd], k3)
        form3 = lib.ic_conv3x3_c128_pick_form(N, h4, w4, step_flags)
        wino = form3 != 0
        from_profiles = None
        if counters and k3 in counters.get('kernels', {}) and counters.get('input_shape') == [N, 3, H, Wd]:
            # the same fractions recomputed from the TRACKED rocprofv3 files alone (no number of this run in them): the launch's
            # duration with nothing beside it, and in the shipped schedule (duration under the tracer / the concurrency it saw)
            ke = counters['kernels'][k3]
            ex_f = enc_l['executed_flop_per_launch']

            def frac_of(us):
                return round(ex_f / (us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4) if us else None
            from_profiles = {'file': counters.get('file'), 'kernel': k3, 'executed_flop_per_launch': ex_f,
                             'alone_us': ke.get('avg_us_rocprof'), 'alone_frac': frac_of(ke.get('avg_us_rocprof')),
                             'work_groups': ke.get('work_groups'),
                             'alone_frac_of_occupied_cus': (round(frac_of(ke.get('avg_us_rocprof')) * 256.0 / ke['work_groups'], 4)
                                                            if ke.get('work_groups') and ke['work_groups'] < 256 and ke.get('avg_us_rocprof') else None),
                             'in_flight_us_under_tracer': ke.get('avg_us_in_flight'), 'concurrency': counters.get('concurrency'),
                             'in_flight_us_over_concurrency': ke.get('avg_us_over_concurrency'),
                             'in_flight_frac': frac_of(ke.get('avg_us_over_concurrency')),
                             'mfma_busy_share': ke.get('mfma_busy_share'),
                             'note': 'frac = executed FLOPs per launch / duration / 157.3 TFLOP/s with durations from the rocprofv3 kernel '
                                     'traces under profiles/ (alone: one image at a time; in flight: the shipped schedule under the tracer, '
                                     'which overlaps streams less than the untraced run `frac` is measured on)'}
        roofline = {'kernel': enc_l['plan']['kernel'] + ' (ic_conv3x3_c128_auto_f32, encoder residual stack, in-step)',
                    'algorithm': {0: 'direct', 1: 'winograd F(2x2,3x3)', 2: 'winograd F(4x4,3x3)'}[form3], 'bound': 'mfma',
                    'achieved': enc_l['achieved'], 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': enc_l['frac'],
                    # the same launch in the FLOPs of the direct form (SURVEY 8(d)'s 294,912 per output pixel): what the layer delivers;
                    # > 1 for the Winograd forms, which execute 16/36 (F(2x2)) or 36/144 (F(4x4)) of them
                    'direct_equivalent_tflops': enc_l['direct_equivalent_tflops'], 'direct_equivalent_frac': enc_l['direct_equivalent_frac'],
                    'traffic': traffic, 'traffic_unit': 'HBM bytes per launch', 'traffic_source': traffic_src,
                    'algorithmic_bytes_per_launch': int(3 * 512 * N * h4 * w4 + {0: 589824, 1: 1048576, 2: 2359296}[form3]),
                    'launches_per_step': 2 * (6 * int(ae_cfg.arch_param_B) + 2),
                    'note': 'frac is against the 2.4 GHz peak; under this load the chip sustains ~1.75 GHz (DESIGN section 3: in-kernel stamps, 0.70 of the '
                            'matrix-pipe cycles).  achieved = FLOPs the matrix pipe executes (Winograd F(2x2): 16/36 of the direct form, F(4x4): 36/144) / the time in which the chip '
                            'completes one launch of the stack: HIP events around {} stacks in flight, one per stream, as in the step; '
                            '`encoder.alone` = the same launch with nothing beside it (its duration in a kernel trace, on plan.cus of 256 CUs)'.format(n_flight),
                    'from_profiles': from_profiles, 'encoder': enc_l, 'decoder': dec_l}
        # The in-flight figure cannot be read from a kernel trace (the tracer serialises the streams); tools/w4_inflight_stamps.py
        # measures the same schedule with in-kernel stamps on a -DW4_LAUNCH_STAMPS build and commits profiles/rNN_inflight_stamps.json:
        #   stacks  the scenario `frac` is measured on (n residual stacks in flight), timed BOTH ways in one run: HIP events as here
        #           (wall time of the burst / launches) and stamps (union of the launches' [first wave start, last store acknowledged]
        #           intervals / launches -- the events figure also pays the gaps in which no launch runs: events_over_stamps);
        #   step    the timed region of this benchmark itself (other kernels share the chip there), stamps only.
        # `frac` of this line must be reproducible from that file: it is compared with the file's events figure of the same scenario;
        # more than 5 % apart, the file's figure replaces it and the live one moves to `frac_live_events`.
        stamps = load_stamps(ROOT)
        if stamps and stamps.get('input_shape') == [N, 3, H, Wd] and stamps.get('images_in_flight') == n_flight and form3 == 2 \
                and 'dominant' in stamps.get('stacks', {}) and 'dominant' in stamps.get('step', {}):
            sk, sp = stamps['stacks'], stamps['step']
            file_frac = sk.get('hip_events_frac') or sk['dominant']['frac']
            ratio = roofline['frac'] / file_frac if file_frac else None
            roofline['from_stamps'] = {
                'file': stamps['file'],
                'stacks_in_flight': {'hip_events_us_per_launch': sk.get('hip_events_us_per_launch'), 'hip_events_frac': sk.get('hip_events_frac'),
                                     'stamps_us_per_launch': sk['dominant']['us_per_launch_under_concurrency'], 'stamps_frac': sk['dominant']['frac'],
                                     'events_over_stamps': sk.get('events_over_stamps')},
                'in_step': {'stamps_us_per_launch': sp['dominant']['us_per_launch_under_concurrency'], 'stamps_frac': sp['dominant']['frac'],
                            'launch_duration_us_mean': sp['kernels'][sp['dominant']['kernel']]['launch_duration_us']['mean'],
                            'concurrency': sp['kernels'][sp['dominant']['kernel']]['concurrency'],
                            'chip_mfma_issue_share': sp.get('chip_mfma_issue_share'), 'shader_clock_ghz': sp.get('shader_clock_ghz_under_load'),
                            'mpix_per_s_wall_with_stamps': sp.get('mpix_per_s_wall')},
                'this_run_frac_over_file_events_frac': round(ratio, 4) if ratio else None,
                'note': 'in-kernel s_memrealtime stamps, no tracer; frac = executed FLOPs per launch / us per launch / 157.3 TFLOP/s'}
            # `frac` / `achieved` stay THIS run's measurement; a disagreement with the committed file is flagged, never substituted
            roofline['from_stamps']['agrees_within_5_percent'] = bool(ratio is not None and abs(ratio - 1.0) <= 0.05)
        sym = N * int(ae_cfg.num_chan_bn) * (H // 8) * (Wd // 8)
        roofline_pc = {'kernel': 'context model, 4 masked conv3d layers + cross-entropy (ic_pc_bitcost_f32), standalone',
                       'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': PEAK_F32_MFMA_TFLOPS,
                       'achieved': round(FLOP_PER_SYMBOL_PC_LIVE * sym / (ms_pc * 1e-3) / 1e12, 2),
                       'frac': round(FLOP_PER_SYMBOL_PC_LIVE * sym / (ms_pc * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                       'dense_algorithmic_tflops': round(FLOP_PER_SYMBOL_PC * sym / (ms_pc * 1e-3) / 1e12, 2),
                       'dense_algorithmic_frac': round(FLOP_PER_SYMBOL_PC * sym / (ms_pc * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                       'ms': round(ms_pc, 4), 'symbols': sym,
                       'note': 'achieved counts the live taps of the causal masks (36,912 FLOP/symbol); SURVEY 8(d) dense figure 47,520 alongside'}
        # ---- the six 5x5 / stride-2 layers around the stacks, each alone on the stream, on the step's own activations ----
        try:
            go_enc()
            go_dec()
            extra['layers_5x5'] = edge_layer_table(torch, lib, _lib, W, ae, ae_cfg, pipe.x, enc.qhard, go_enc.out, go_dec.out, timed, st, N, H, Wd, step_flags)
        except Exception as ex:                                        # informational only
            extra['layers_5x5'] = {'error': str(ex)[:300]}
        for e in ev:
            lib.ic_event_destroy(e)

        # ---- the other sharing arrangement and the north_star's 256x256 shape through the same step() ----
        other = 'full_chip' if share == 'cu_range' else 'cu_range'          # under 'auto' / 'serial': the CU-range arrangement
        try:
            po = Pipeline(dev, a.ae_config, other, seed=rank).set_input(N, H, Wd)
            dt, _ = run(po, 20, 3)
            extra['other_branch_sharing'] = {'share': other, 'value': round(N * H * Wd * 20 / dt / 1e6, 3), 'unit': 'Mpix/s',
                                             'ms_per_step': round(dt / 20 * 1e3, 4)}
            po.branch.close()
        except Exception as ex:                                        # informational only
            extra['other_branch_sharing'] = {'error': str(ex)[:200]}
        shapes = []
        for (n2, h2, w2) in ((1, 256, 256), (8, 256, 256)):
            if (n2, h2, w2) == (N, H, Wd):
                continue
            ps = Pipeline(dev, a.ae_config, share, seed=rank).set_input(n2, h2, w2)
            dt1, _ = run(ps, 30, 5)
            nf2 = flight_for_shape(lib, n2, h2, w2, n_flight)          # enough launches in flight to fill the chip with THIS shape
            k2 = 30 if nf2 <= 6 else 5 * nf2
            sch = InFlight(torch, ps, dev, nf2, a.ae_config, rank, graphs=bool(a.graphs)) if nf2 > 1 else ps
            dt, _ = run(sch, k2, nf2 + 2) if nf2 > 1 else (dt1 * k2 / 30.0, None)
            # executed FLOPs: the 64 3x3 layers (589,824 FLOP per input pixel and network half) in the form the plan runs, the 5x5 layers direct
            fl2 = _lib.CONV3_IN_FLIGHT(nf2) if nf2 > 1 else 0
            f3 = {0: 1.0, 1: 16.0 / 36.0, 2: 36.0 / 144.0}[lib.ic_conv3x3_c128_pick_form(n2, h2 // 4, w2 // 4, fl2)]
            flop = n2 * h2 * w2 * (2 * 589824.0 * f3 + (FLOP_PER_PX_ENC - 589824.0) + (FLOP_PER_PX_DEC - 589824.0))
            ent = {'batch': n2, 'height': h2, 'width': w2, 'value': round(n2 * h2 * w2 * k2 / dt / 1e6, 3), 'unit': 'Mpix/s',
                   'ms_per_step': round(dt / k2 * 1e3, 4), 'images_in_flight': nf2,
                   'one_image_at_a_time': {'value': round(n2 * h2 * w2 * 30 / dt1 / 1e6, 3), 'ms_per_step': round(dt1 / 30 * 1e3, 4)},
                   'executed_frac_of_mfma_peak_whole_step': round(flop * k2 / dt / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                   'plan_3x3': plan_name(lib, _lib, n2, h2 // 4, w2 // 4, fl2)['kernel']}
            shapes.append(ent)
            ps.branch.close()
        extra['shapes'] = shapes
        # BASELINE configs[4]: the high-rate configuration on a 4K frame (one image per GPU; a frame fills the chip by itself:
        # 4050 F(4x4) work-groups per 3x3 launch), one frame at a time
        try:
            p4 = Pipeline(dev, 'hi', share, seed=rank).set_input(1, 2160, 3840)
            dt4, out4 = run(p4, 6, 2)
            extra['cfg5_4k'] = {'workload': 'BASELINE configs[4]: ae_configs/cvpr/hi + pc_configs/cvpr/res_shallow on one 3840x2160 frame, encode + bitcost + decode',
                                'value': round(2160 * 3840 * 6 / dt4 / 1e6, 3), 'unit': 'Mpix/s', 'ms_per_step': round(dt4 / 6 * 1e3, 3),
                                'plan_3x3': plan_name(lib, _lib, 1, 540, 960, 0)['kernel']}
            p4.branch.close()
            del p4
        except Exception as ex:                                        # informational only
            extra['cfg5_4k'] = {'error': str(ex)[:300]}
        if a.pipelined:
            extra['in_flight_sweep'] = pipelined_section(torch, dev, a, N, H, Wd, pipe)
        if world == 1:
            try:
                extra['train'] = train_object(torch, dist, dev)
            except Exception as ex:                                    # informational in this line; --mode train is the contract form
                extra['train'] = {'error': str(ex)[:300]}

    # ---- CPU baseline: the oracle on the host cores (rank 0, N == 1 only) ----
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not a.no_extras:
        from oracle import oracle as O
        # torch's CPU conv kernels oversubscribe badly beyond a few dozen threads on the 256-core host
        # (16 threads is the fastest setting measured on the MI355X box)
        cores = min(os.cpu_count() or 1, 16)
        torch.set_num_threads(cores)
        sample = pipe.x_np[:1]
        with torch.no_grad():
            O.validate_forward(sample[:, :, :64, :64], pipe.wts, ae_cfg.as_dict(), torch.float32)     # warm-up (primitive caches)
            runs, t1 = 0, time.perf_counter()
            while runs < 3 or (time.perf_counter() - t1 < 10.0 and runs < 200):
                O.validate_forward(sample, pipe.wts, ae_cfg.as_dict(), torch.float32)
                runs += 1
            dt = (time.perf_counter() - t1) / runs
        cpu = {'value': round(sample.shape[2] * sample.shape[3] / dt / 1e6, 4), 'unit': 'Mpix/s', 'cores': cores,
               'kind': 'port', 'sample': '{} x image 3x{}x{} through the torch-CPU fp32 oracle (encode + bitcost + '
               'decode), {} torch threads, {:.2f} s per image'.format(runs, sample.shape[2], sample.shape[3], cores, dt)}

    if rank == 0:
        C = int(ae_cfg.num_chan_bn)
        flop_step = N * H * Wd * (FLOP_PER_PX_ENC + FLOP_PER_PX_DEC + FLOP_PER_SYMBOL_PC * C / 64.0)
        if same_stream:
            sched_text = ('encode, context-model bitcost, decode(qhard) of an image in that order on one stream, every launch on the whole '
                          'chip (val.py:85-89 evaluates bitcost and reconstruction in one session.run); ' +
                          ('one image at a time' if n_flight == 1 else
                           '{} independent batch-1 images in flight, one stream each, steps issued round-robin (val.py:157-158 loops over '
                           'independent images): each step is one image, the launches of one image fill the kernel-boundary bubbles '
                           'of the others'.format(n_flight)))
        else:
            sched = ('a stream limited to the {} CUs the decoder leaves idle'.format(cus) if cus else
                     'a second stream next to the decoder (which fills the chip)')
            sched_text = ('one image at a time; bitcost and decode of that image run concurrently (val.py:85-89 evaluates both in '
                          'one session.run), the bitcost on ' + sched)
        out = {
            'metric': 'Megapixels/s encode+pc-logits (and decode) per node',
            'value': round(value, 3), 'unit': 'Mpix/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(elapsed / a.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            # rounds 1-2 quoted `value` one image at a time; since round 3 it is the throughput of the evaluation loop with
            # several independent images in flight (still one image per step) -- the latency-style figure is `one_image_at_a_time`
            'value_definition': ('throughput, {} independent batch-1 images in flight on one GPU; NOT comparable with BENCH_r01/r02 '
                                 '(one image at a time: see one_image_at_a_time)'.format(n_flight) if n_flight > 1 else 'one image at a time'),
            'config': {'workload': 'BASELINE configs[1]: Kodak-shaped image {}x3x{}x{} per GPU per step, '
                                   'ae_configs/cvpr/{} + pc_configs/cvpr/res_shallow, encode + parallel '
                                   'context-model bitcost + decode(qhard); random-init weights'.format(
                                       N, H, Wd, a.ae_config),
                       'batch_per_gpu': N, 'height': H, 'width': Wd, 'parallelism': 'image-sharded x{}'.format(world),
                       'schedule': sched_text},
            'model_tflops_per_s': round(flop_step * world * a.steps / elapsed / 1e12, 2),
            'roofline': roofline, 'roofline_context_model': roofline_pc, 'cpu_baseline': cpu,
        }
        out.update(extra)
        # the driver's record keeps the contract keys, `config`, `roofline` and `cpu_baseline` of this line: the like-for-like figure
        # with the reference's one-sess.run-per-image loop (val.py:157-158) rides in `config` (numbers only) so that it survives too
        one = extra.get('one_image_at_a_time')
        out['config']['images_in_flight'] = n_flight
        out['ranks'] = ranks_seen
        out['config']['one_image_at_a_time_mpix_s'] = one['value'] if one else (round(value, 3) if n_flight == 1 else None)
        out['config']['one_image_at_a_time_ms_per_step'] = one['ms_per_step'] if one else (round(elapsed / a.steps * 1e3, 4) if n_flight == 1 else None)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()                 # rank 0 is still timing the stage split / dominant kernel: leave together
        dist.destroy_process_group()


def self_launch(n):
    """re-run this command line under torch.distributed.run: one process per GPU on this node, rendezvous on 127.0.0.1 (the
    container's host name may not resolve), a free port.  The ranks' output is this process's output; returns their exit code."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf inter-process handles (RCCL, peer-mapped BatchNorm exchange)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc:
        sys.exit(rc)
    return rc


def edge_layer_table(torch, lib, _lib, W, ae, ae_cfg, x, qhard, stack_out_enc, stack_out_dec, timed, st, N, H, Wd, step_flags=0):
    """h1, h2, to_bn (autoencoder.py:222,223,237) and from_bn, h12, h13 (:251,264,265) one by one through the generic C-ABI
    entry points, each fed the tensor it sees inside the step.  Per layer: time, algorithmic FLOPs (dense, 2 FLOP per
    MAC) against the fp32 MFMA peak, algorithmic bytes (input + output activation once) against the 8 TB/s HBM peak."""
    E, D = W.ENC, W.DEC
    C = int(ae_cfg.num_chan_bn)
    Cb = C + (1 if bool(ae_cfg.heatmap) else 0)
    dev = x.device
    P = _lib.ptr
    half = torch.empty((N, 64, H // 2, Wd // 2), device=dev)
    quar = torch.empty((N, 128, H // 4, Wd // 4), device=dev)
    half_d = torch.empty_like(half)
    quar_d = torch.empty_like(quar)
    bott = torch.empty((N, Cb, H // 8, Wd // 8), device=dev)
    xo = torch.empty_like(x)
    pl = ae._plan
    px = float(N * H * Wd)
    layers = [
        ('h1', 'conv5s2_cin3_mfma_kernel', lambda: lib.ic_conv2d_bn_act_f32(P(x), P(pl[E + '/h1'][0]), P(pl[E + '/h1'][1]), P(pl[E + '/h1'][2]), None, None,
                                                                           P(half), N, 3, H, Wd, 64, 5, 5, 2, 1, None, None, st),
         2.0 * 75 * 64 * px / 4, 4.0 * (3 * px + 64 * px / 4)),
        ('h2', 'conv_mfma_kernel', lambda: lib.ic_conv2d_mfma_bn_act_f32(P(half), P(pl[E + '/h2'][0]), P(pl[E + '/h2'][1]), P(pl[E + '/h2'][2]), P(quar),
                                                                        N, 64, H // 2, Wd // 2, 128, 5, 5, 2, 0, 1, st),
         2.0 * 25 * 64 * 128 * px / 16, 4.0 * (64 * px / 4 + 128 * px / 16)),
        ('to_bn', 'conv_mfma_kernel', lambda: lib.ic_conv2d_mfma_bn_act_f32(P(stack_out_enc), P(pl[E + '/to_bn'][0]), P(pl[E + '/to_bn'][1]), P(pl[E + '/to_bn'][2]),
                                                                           P(bott), N, 128, H // 4, Wd // 4, Cb, 5, 5, 2, 0, 0, st),
         2.0 * 25 * 128 * Cb * px / 64, 4.0 * (128 * px / 16 + Cb * px / 64)),
        ('from_bn', 'deconv3_mfma_kernel', lambda: lib.ic_deconv2d_bn_act_f32(P(qhard), P(pl[D + '/from_bn'][0]), P(pl[D + '/from_bn'][1]), P(pl[D + '/from_bn'][2]),
                                                                             P(quar_d), N, C, H // 8, Wd // 8, 128, 3, 3, 1, None, None, 0, st),
         2.0 * 9 * C * 128 * px / 64, 4.0 * (C * px / 64 + 128 * px / 16)),
        ('h12', 'deconv5_mfma_kernel', lambda: lib.ic_conv2d_mfma_bn_act_f32(P(stack_out_dec), P(pl[D + '/h12'][0]), P(pl[D + '/h12'][1]), P(pl[D + '/h12'][2]),
                                                                            P(half_d), N, 128, H // 4, Wd // 4, 64, 5, 5, 2, 1, 1, st),
         2.0 * 25 * 128 * 64 * px / 16, 4.0 * (128 * px / 16 + 64 * px / 4)),
        ('h13', 'deconv5_cout3_mfma_kernel', lambda: lib.ic_deconv2d_bn_act_f32(P(half_d), P(pl[D + '/h13'][0]), P(pl[D + '/h13'][1]), P(pl[D + '/h13'][2]),
                                                                               P(xo), N, 64, H // 2, Wd // 2, 3, 5, 5, 0, None, None, 0, st),
         2.0 * 25 * 64 * 3 * px / 4, 4.0 * (64 * px / 4 + 3 * px)),
    ]
    # h2 / h12 in the form the step runs when the 3x3 stack of the same call runs F(4x4): ONE 3x3 convolution over / to phases on
    # the F(4x4) kernel (csrc/conv3x3_wino4.hip); FLOPs stay the dense 5x5 figure (the form executes 36 / 100 of them)
    h4, w4 = H // 4, Wd // 4
    in_step_w4 = bool(getattr(ae, '_edge_both', False)) and lib.ic_wino4_conv5s2_supported(N, h4, w4) == 1 and \
        not (step_flags & _lib.CONV5_NO_WINO4) and (bool(step_flags & _lib.CONV5_WINO4) or lib.ic_conv3x3_c128_pick_form(N, h4, w4, step_flags) == 2
                                                    or lib.ic_wino4_conv5s2_workgroups(N, h4, w4, 0) >= 160)
    if bool(getattr(ae, '_edge_both', False)) and lib.ic_wino4_conv5s2_supported(N, h4, w4) == 1:
        half_p = torch.empty((N, 256, h4, w4), device=dev)
        _lib.check(lib.ic_conv2d_bn_act_f32(P(x), P(pl[E + '/h1'][0]), P(pl[E + '/h1'][1]), P(pl[E + '/h1'][2]), None, None, P(half), N, 3, H, Wd, 64, 5, 5, 2, 1,
                                            None, None, st), 'h1')
        _lib.check(lib.ic_space_to_depth2_f32(P(half), P(half_p), N, 64, H // 2, Wd // 2, st))
        off2 = lib.ic_conv2d_mfma_packed_floats(5, 5, 64, 128, 2, 0) * 4
        off12 = lib.ic_conv2d_mfma_packed_floats(5, 5, 128, 64, 2, 1) * 4
        w2 = ctypes.c_void_p(pl[E + '/h2'][0].data_ptr() + off2)
        w12 = ctypes.c_void_p(pl[D + '/h12'][0].data_ptr() + off12)
        layers.insert(2, ('h2 as F(4x4) over phases', 'wino4_3x3_kernel<256, 128>',
                          lambda: lib.ic_wino4_conv5s2_c64_c128_bn_act_f32(P(half_p), w2, P(pl[E + '/h2'][1]), P(pl[E + '/h2'][2]), P(quar), N, h4, w4, 1, 0, st),
                          2.0 * 25 * 64 * 128 * px / 16, 4.0 * (64 * px / 4 + 128 * px / 16)))
        layers.insert(6, ('h12 as F(4x4) to phases', 'wino4_3x3_kernel<128, 256>',
                          lambda: lib.ic_wino4_deconv5s2_c128_c64_bn_act_f32(P(stack_out_dec), w12, P(pl[D + '/h12'][1]), P(pl[D + '/h12'][2]), P(half_d), N, h4, w4, 1, 0, st),
                          2.0 * 25 * 128 * 64 * px / 16, 4.0 * (128 * px / 16 + 64 * px / 4)))
    out, total = [], 0.0
    for name, kernel, fn, flop, nbytes in layers:
        _lib.check(fn(), name)                       # also produces the next layer's input (half -> h2, half_d -> h13)
        us = timed(fn, 20, warm=3) * 1e3
        runs = (' as F(4x4)' in name) if (name.split(' ')[0] in ('h2', 'h12') and in_step_w4) else (' as F(4x4)' not in name)
        if runs:
            total += us
        tf, gbs = flop / us / 1e6, nbytes / us / 1e3
        out.append({'layer': name, 'kernel': kernel, 'us': round(us, 2), 'algorithmic_tflops': round(tf, 1),
                    'frac_of_mfma_peak': round(tf / PEAK_F32_MFMA_TFLOPS, 3), 'algorithmic_bytes': int(nbytes),
                    'hbm_gb_per_s': round(gbs, 0), 'frac_of_hbm_peak': round(gbs / 8000.0, 3),
                    'bound': 'mfma' if flop / (PEAK_F32_MFMA_TFLOPS * 1e6) > nbytes / 8e6 else 'hbm', 'form_of_the_step': bool(runs),
                    # the F(4x4)-over-phases form executes 36 of every 100 dense multiplies: its frac_of_mfma_peak (dense FLOPs) can pass 1
                    'executed_fraction_of_dense_flops': 0.36 if ' as F(4x4)' in name else 1.0})
    return {'layers': out, 'total_us': round(total, 1),
            'note': 'each layer alone on the stream on the tensor it sees in the step; FLOPs dense (SURVEY 8(d)), bytes = input + output once; '
                    'total_us sums the forms the step runs (form_of_the_step)'}


def load_counters(root):
    """the newest profiles/rNN_counters.json (tools/profile_digest.py), or None"""
    import glob
    files = sorted(glob.glob(os.path.join(root, 'profiles', 'r[0-9][0-9]_counters.json')))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            c = json.load(f)
        c['file'] = os.path.relpath(files[-1], root)
        return c
    except (IOError, OSError, ValueError):
        return None


def load_stamps(root):
    """the newest profiles/rNN_inflight_stamps.json (tools/w4_inflight_stamps.py: in-kernel launch stamps, no tracer), or None"""
    import glob
    files = sorted(glob.glob(os.path.join(root, 'profiles', 'r[0-9][0-9]_inflight_stamps.json')))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            c = json.load(f)
        c['file'] = os.path.relpath(files[-1], root)
        return c
    except (IOError, OSError, ValueError):
        return None


def ranks_report(torch, dist, dev, rank, world, backend):
    """self-verification of a multi-GPU run (every rank calls it once, after the timed region): an all-reduce (SUM) of a one-hot rank
    mask must come back all ones -- every rank of the world took part in a collective over this backend --, and an all-gather of
    (HIP device index, PCI bus id hash) shows that the ranks sit on distinct devices.  -> dict on every rank."""
    info = {'world_size': world, 'backend': ('rccl (torch nccl)' if backend == 'nccl' else backend) if world > 1 else None}
    if world == 1:
        info.update({'rccl_ranks_seen': 1, 'devices': [torch.cuda.get_device_name(dev) + ' #{}'.format(dev.index)]})
        return info
    cdev = dev if backend == 'nccl' else 'cpu'
    mask = torch.zeros(world, dtype=torch.int32, device=cdev)
    mask[rank] = 1
    dist.all_reduce(mask, op=dist.ReduceOp.SUM)
    props = torch.cuda.get_device_properties(dev)
    ident = torch.tensor([dev.index, int(getattr(props, 'pci_bus_id', -1)), int(getattr(props, 'pci_device_id', -1))], dtype=torch.int64, device=cdev)
    allid = [torch.zeros_like(ident) for _ in range(world)]
    dist.all_gather(allid, ident)
    ids = [tuple(int(v) for v in t.tolist()) for t in allid]
    info.update({'rccl_ranks_seen': int((mask == 1).sum().item()), 'dist_world_size': dist.get_world_size(),
                 'devices': ['rank {}: hip device {} pci bus {} dev {}'.format(r, *i) for r, i in enumerate(ids)],
                 'distinct_devices': len(set(ids))})
    return info


def max_over_ranks(torch, dist, elapsed, dev, world, backend):
    if world == 1:
        return elapsed
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def plan_name(lib, _lib, N, h4, w4, flags):
    """which kernel(s) ic_conv3x3_c128_auto_f32 launches for this shape and these flags (the library's own plan query)."""
    form = lib.ic_conv3x3_c128_pick_form(N, h4, w4, flags)
    if form == 0:
        return {'kernel': 'conv3x3_c128_kernel', 'cus': 256, 'form': 'direct'}
    if form == 2:
        wgs = int(lib.ic_wino4_3x3_c128_workgroups(N, h4, w4))
        return {'kernel': 'wino4_3x3_kernel<128, 128>', 'cus': min(256, wgs), 'form': 'winograd F(4x4,3x3)', 'work_groups': wgs}
    pl = (ctypes.c_longlong * 5)()
    _lib.check(lib.ic_wino3x3_c128_plan(N, h4, w4, flags, pl))
    names = []
    if pl[0]:
        names.append('wino3x3_c128_shared_kernel' if w4 % 2 == 0 else 'wino3x3_c128_kernel')
    if pl[1]:
        names.append('wino3x3_c128_tn_kernel<{}>'.format(int(pl[2])))
    if pl[3]:
        names.append('wino3x3_c128_t16_kernel')
    if pl[4]:
        names.append('wino3x3_c128_ksplit_kernel')
    return {'kernel': ' + '.join(names), 'cus': int(lib.ic_wino3x3_c128_workgroups(N, h4, w4, flags)),
            'tile_groups': {'whole_k': int(pl[0]), 'segment_jobs': int(pl[1]), 'nb': int(pl[2]), 't16': int(pl[3]), 'ksplit': int(pl[4])}}


def pipelined_section(torch, dev, a, N, H, Wd, pipe):
    """Extra, NOT the contract value: the same step with 2 / 3 independent batch-1 images in flight on their own streams (what a
    val.py loop over the 24 Kodak images can do: the images are independent).  Kernels of one image fill the launch-boundary
    bubbles of the other."""
    from imgcomp_cvpr_amd import bits
    out = {}
    for ns in (2, 3, 4, 6):
        try:
            pipes = [Pipeline(dev, a.ae_config, 'serial', seed=i).set_input(N, H, Wd) for i in range(ns)]
            streams_ = [torch.cuda.Stream(device=dev) for _ in range(ns)]

            def one(pl):
                e = pl.ae.encode(pl.x, is_training=False)
                b = pl.pc.bitcost(e.qbar, e.symbols, is_training=False, pad_value=pl.pad_value)
                bits.bitcost_to_bpp(b, pl.x)
                return pl.ae.decode(e.qhard, is_training=False)
            torch.cuda.synchronize(dev)
            for i in range(2 * ns):
                with torch.cuda.stream(streams_[i % ns]):
                    one(pipes[i % ns])
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            n_img = 12 * ns
            for i in range(n_img):
                with torch.cuda.stream(streams_[i % ns]):
                    one(pipes[i % ns])
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t1
            out['{}_images_in_flight'.format(ns)] = {'value': round(N * H * Wd * n_img / dt / 1e6, 3), 'unit': 'Mpix/s',
                                                      'ms_per_image': round(dt / n_img * 1e3, 4), 'images': n_img}
            for pl in pipes:
                pl.branch.close()
        except Exception as ex:                                       # informational only
            out['{}_images_in_flight'.format(ns)] = {'error': str(ex)[:200]}
    out['note'] = 'independent batch-1 images on their own streams on one GPU; not the contract value'
    return out


def train_main(a, dev, rank, world):
    """--mode train: BASELINE configs[2] -- ae_configs/cvpr/med + res_shallow, random 128x128 crops, one full training step per
    bench step: forward in training mode, MS-SSIM loss, hand-written backward, RCCL gradient all-reduce of the three flat
    buckets, two Adam updates.  --scaling weak: 32 crops per GPU (the global batch grows with N); --scaling strong: cfg3's
    batch of 32 split over the ranks, what train.py does with --batch_size 32 (reference train.py:150-153 feeds one batch)."""
    import torch
    import torch.distributed as dist
    GLOBAL, H, Wd = 32, 128, 128
    if a.scaling == 'strong':
        assert GLOBAL % world == 0, 'strong scaling splits a batch of 32: --gpus must divide it'
        N = GLOBAL // world
    else:
        N = GLOBAL
    elapsed, out, tr = train_steps_timed(torch, dist, dev, rank, world, N, H, Wd, a.steps, a.warmup, a.backend)
    ranks_seen = ranks_report(torch, dist, dev, rank, world, a.backend)
    if rank == 0:
        from imgcomp_cvpr_amd import _lib, weights as W
        try:
            roof = train_roofline(torch, _lib.lib, _lib, W, tr, dev, N, H, Wd, elapsed / a.steps * 1e3)
        except Exception as ex:                                    # informational
            roof = {'error': str(ex)[:300]}
        print(json.dumps({
            'metric': 'training images/s (cfg3: cvpr/med + res_shallow, 128x128 crops, {})'.format(
                'batch 32 per GPU' if a.scaling == 'weak' else 'global batch 32 split over the GPUs'),
            'value': round(N * world * a.steps / elapsed, 2), 'unit': 'img/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(elapsed / a.steps * 1e3, 3), 'higher_is_better': True, 'scaling': a.scaling, 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[2]: train step, ae_configs/cvpr/med + pc_configs/cvpr/res_shallow, {}x3x{}x{} '
                                   'per GPU, MS-SSIM loss, two Adam optimisers, data-parallel gradient all-reduce'.format(N, H, Wd),
                       'batch_per_gpu': N, 'global_batch': N * world, 'parallelism': 'dp{}'.format(world),
                       'cross_replica_batchnorm': tr.graph._bn_world() > 1},
            'ranks': ranks_seen, 'roofline': roof,
            'mpix_per_s': round(N * H * Wd * world * a.steps / elapsed / 1e6, 3),
            'last_step': {k: round(float(v), 5) for k, v in out.items()}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def train_steps_timed(torch, dist, dev, rank, world, N, H, Wd, steps, warmup, backend='nccl'):
    """`warmup` + `steps` cfg3 training steps on one fixed synthetic batch of N crops per rank, timed like the contract's region
    (barrier + synchronize on both sides, max over the ranks) -> (seconds for `steps`, the last step's scalars, the Trainer)"""
    from imgcomp_cvpr_amd import config_parser as cp, weights as W, training
    ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'med'))
    pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
    tr = training.Trainer(ae_cfg, pc_cfg, W.synthetic_weights(ae_cfg, pc_cfg), dev, num_itr_per_epoch=1000)
    x = torch.as_tensor(W.synthetic_image((N, 3, H, Wd), 'natural', seed=rank)).float().to(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
    out = None
    for _ in range(warmup):
        out = tr.step(x)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = tr.step(x)
    barrier()
    return max_over_ranks(torch, dist, time.perf_counter() - t0, dev, world, backend), out, tr


def train_step_flops(W, graph, N, H, Wd):
    """FLOPs one cfg3 training step EXECUTES on the matrix cores (2 FLOP per multiply-add), from the layer table of the graph that just
    ran: per convolution forward + data gradient (none for h1: the image needs none) + filter gradient, each in the form it runs --
    3x3 128 -> 128 layers: Winograd F(4x4) executes 36 / 144 of the direct multiplies, F(2x2) 16 / 36 (graph._w3_f4 says which per
    direction; their filter gradients are taken in the F(2x2) domain: 16 / 36); every other layer direct; the context model's four
    masked conv3d layers on their live taps (36,912 FLOP per symbol for k = 24 ... scaled from the filter shapes), x 3 passes.
    -> (executed, direct-form equivalent)"""
    C, B = int(graph.C), int(graph.B)
    f4 = getattr(graph, '_w3_f4', (False, False))
    executed = direct = 0.0
    size = {'h1': (H // 2, Wd // 2), 'h2': (H // 4, Wd // 4), 'to_bn': (H // 8, Wd // 8), 'from_bn': (H // 4, Wd // 4),
            'h12': (H // 2, Wd // 2), 'h13': (H, Wd)}
    for scope, kind, shape in W.ae_conv_specs(C, B, bool(graph.heatmap)):
        kh, kw, a_, b_ = shape
        oh, ow = size.get(scope.rsplit('/', 1)[-1], (H // 4, Wd // 4))
        taps = kh * kw * a_ * b_
        if kind == 'deconv' or kh == 5:
            # stride 2: a 5x5 / 2 convolution computes its output at (H/2 x W/2) positions; the transposed one meets every INPUT position
            pos = oh * ow if kind == 'conv' else (oh // 2) * (ow // 2)
        else:
            pos = oh * ow
        one = 2.0 * taps * pos * N
        passes = 2 if scope.endswith('/h1') else 3
        direct += passes * one
        if (kh, kw, a_, b_) == (3, 3, 128, 128) and kind == 'conv':
            executed += one * ((0.25 if f4[0] else 16.0 / 36.0) + (0.25 if f4[1] else 16.0 / 36.0) + 16.0 / 36.0)
        else:
            executed += passes * one
    # context model: live taps of the causal masks (first layer's mask excludes the centre: 13 of 18 taps live, the others 14)
    k, L = int(graph.k), int(graph.L)
    sym = N * C * (H // 8) * (Wd // 8)
    live = 2.0 * (13 * 1 * k + 14 * k * k + 14 * k * k + 14 * k * L)
    dense = 2.0 * 18 * (1 * k + k * k + k * k + k * L)
    executed += 3 * live * sym
    direct += 3 * dense * sym
    return executed, direct


def train_roofline(torch, lib, _lib, W, tr, dev, N, H, Wd, ms_per_step):
    """roofline object of the training step: executed FLOPs per step / step time / fp32 MFMA peak for the whole step,
    and the dominant BACKWARD kernel -- the 3x3 128 -> 128 filter gradient in the Winograd domain (ic_conv3x3_c128_wgrad_f32: producer +
    slice reduction, 64 calls per step) -- timed live with HIP events on the step's own shape."""
    executed, direct = train_step_flops(W, tr.graph, N, H, Wd)
    st = _lib.current_stream(dev)
    h4, w4 = H // 4, Wd // 4
    x = torch.relu(torch.randn((N, 128, h4, w4), device=dev))
    dy = torch.randn((N, 128, h4, w4), device=dev) * 0.1
    w = torch.randn((3, 3, 128, 128), device=dev) * 0.05
    dw = torch.empty_like(w)
    need = lib.ic_conv3x3_c128_wgrad_workspace_bytes(N, h4, w4)
    ws = torch.empty(max(int(need), 16), dtype=torch.uint8, device=dev)
    ev = [ctypes.c_void_p() for _ in range(2)]
    for e in ev:
        _lib.check(lib.ic_event_create(ctypes.byref(e)))

    def call():
        _lib.check(lib.ic_conv3x3_c128_wgrad_f32(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), N, h4, w4, _lib.ptr(w), 1e-4, _lib.ptr(ws), need, st))
    for _ in range(5):
        call()
    torch.cuda.synchronize(dev)
    _lib.check(lib.ic_event_record(ev[0], st))
    for _ in range(30):
        call()
    _lib.check(lib.ic_event_record(ev[1], st))
    ms = ctypes.c_float()
    _lib.check(lib.ic_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms)))
    for e in ev:
        lib.ic_event_destroy(e)
    us = ms.value / 30 * 1e3
    wg_direct = 2.0 * 9 * 128 * 128 * N * h4 * w4
    wg_exec = wg_direct * 16.0 / 36.0
    n3 = 2 * (6 * int(tr.graph.B) + 2)
    return {'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': PEAK_F32_MFMA_TFLOPS,
            'achieved': round(executed / (ms_per_step * 1e-3) / 1e12, 2),
            'frac': round(executed / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            'executed_flop_per_step': executed, 'direct_equivalent_flop_per_step': direct,
            'direct_equivalent_frac': round(direct / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            'forms_3x3': {'forward': 'F(4x4)' if tr.graph._w3_f4[0] else 'F(2x2)', 'data_gradient': 'F(4x4)' if tr.graph._w3_f4[1] else 'F(2x2)',
                          'filter_gradient': 'F(2x2) domain'},
            'dominant_backward_kernel': {
                'kernel': 'wino3x3_c128_wgrad8_kernel + wino_wgrad_reduce_kernel (ic_conv3x3_c128_wgrad_f32), alone on the stream',
                'calls_per_step': n3, 'avg_call_us': round(us, 2), 'executed_flop_per_call': wg_exec,
                'achieved': round(wg_exec / us / 1e6, 2), 'frac': round(wg_exec / us / 1e6 / PEAK_F32_MFMA_TFLOPS, 4),
                'share_of_step': round(n3 * us * 1e-3 / ms_per_step, 3)},
            'note': 'whole step: executed FLOPs (Winograd forms counted at what they execute, filter gradients included) / step time / 157.3; '
                    'the element-wise passes of a training step (BatchNorm statistics / apply / backward, optimiser, MS-SSIM) execute no matrix FLOPs and '
                    'are what keeps the fraction below the inference step\'s'}


def train_object(torch, dist, dev, steps=12, warmup=4):
    """the N = 1 inference line's `train` object: BASELINE configs[2] (cfg3) on this GPU -- images/s, ms per step and the last
    step's loss terms (MS-SSIM must lie in (0, 1]; d_loss_scaled = K (1 - MS-SSIM)), so that the driver's BENCH file carries a
    training number too.  `python bench.py --mode train` prints the same measurement as a full contract line."""
    elapsed, out, tr = train_steps_timed(torch, dist, dev, 0, 1, 32, 128, 128, steps, warmup)
    K = float(tr.graph.ae_config.K_ms_ssim)
    ent = {'workload': 'BASELINE configs[2]: one training step, ae_configs/cvpr/med + pc_configs/cvpr/res_shallow, 32x3x128x128 crops, '
                       'MS-SSIM loss, two Adam optimisers; one fixed synthetic batch',
           'value': round(32 * steps / elapsed, 2), 'unit': 'img/s', 'ms_per_step': round(elapsed / steps * 1e3, 3),
           'steps': steps, 'warmup': warmup, 'last_step': {k: round(float(v), 5) for k, v in out.items()},
           'checks': {'ms_ssim_in_unit_interval': bool(0.0 < out['ms_ssim'] <= 1.0),
                      'd_loss_is_K_times_one_minus_ms_ssim': bool(abs(out['d_loss_scaled'] - K * (1.0 - out['ms_ssim'])) < 2e-3 + 1e-6 * K)}}
    try:
        from imgcomp_cvpr_amd import _lib, weights as W
        ent['roofline'] = train_roofline(torch, _lib.lib, _lib, W, tr, dev, 32, 128, 128, elapsed / steps * 1e3)
    except Exception as ex:                                        # informational
        ent['roofline'] = {'error': str(ex)[:300]}
    del tr
    torch.cuda.empty_cache()
    return ent


if __name__ == '__main__':
    main()
