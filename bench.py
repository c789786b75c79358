#!/usr/bin/env python
"""bench.py -- throughput of the imgcomp hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input: BASELINE.json configs[1]
(Kodak-shaped image 1x3x512x768, ae_configs/cvpr/low + pc_configs/cvpr/res_shallow, batch 1):
    encode (normalise, 35 convs, importance map, quantiser) -> context-model bit cost for all symbols
    in parallel + bpp -> decode(qhard) (35 convs, de-normalise, clip)          [val.py:85-89 wiring]
Inputs and weights are resident in HBM before the timed region.  Data: seeded synthetic image and
random-init weights (no network for Kodak or the 0515_1103 checkpoint).

  python bench.py --gpus N --steps K --warmup W
N > 1 is launched by torch.distributed.run, one rank per GPU; the path shards by image (independent
units, no data-path collective), so every rank runs the same per-GPU workload: weak scaling.

Prints ONE JSON line on rank 0 (see the contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (3x3 128->128 conv on the fp32 matrix cores): algorithmic FLOP per
                  launch / average launch duration measured with HIP events on the launch stream
  cpu_baseline -- the CPU oracle (torch fp32 restatement of the reference, all host cores) timed on a
                  bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
# algorithmic work per input pixel, C = 32 (FLOP = 2 MAC, dense, mask-agnostic, halo-free)
FLOP_PER_PX_ENC = 621124.0
FLOP_PER_PX_DEC = 618976.0
FLOP_PER_SYMBOL_PC = 47520.0
CONV3_FLOP_PER_OUT_PX = 2.0 * 9 * 128 * 128      # per 128-channel output pixel of one 3x3 layer


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--warmup', type=int, default=3)
    p.add_argument('--height', type=int, default=512)
    p.add_argument('--width', type=int, default=768)
    p.add_argument('--batch', type=int, default=1)
    p.add_argument('--ae_config', default='low')
    p.add_argument('--no_cpu_baseline', action='store_true')
    a = p.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    assert world == a.gpus, 'WORLD_SIZE {} != --gpus {}'.format(world, a.gpus)
    assert torch.cuda.is_available(), 'bench.py needs a HIP device'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)

    from imgcomp_cvpr_amd import autoencoder, probclass, bits, config_parser as cp, weights as W, _lib
    lib = _lib.lib
    ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', a.ae_config))
    pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
    wts = W.synthetic_weights(ae_cfg, pc_cfg)
    ae = autoencoder.get_network_cls(ae_cfg)(ae_cfg).load_weights(wts, dev)
    pc = probclass.get_network_cls(pc_cfg)(pc_cfg, num_centers=ae_cfg.num_centers).load_weights(wts, dev)
    N, H, Wd = a.batch, a.height, a.width
    x_np = W.synthetic_image((N, 3, H, Wd), 'natural', seed=rank)
    x = torch.as_tensor(x_np).float().to(dev)
    pad_value = float(wts['autoencoder/encoder/centers'][0])

    # The context model and the decoder both hang off the encoder output and do not depend on each other (val.py:85-89):
    # the bitcost goes on a second HIP stream restricted to the CUs the decoder's one-work-group-per-CU 3x3 launches
    # leave idle (imgcomp_cvpr_amd/streams.py; a Kodak map: 64 of 256 CUs), the same arrangement val.py runs.
    from imgcomp_cvpr_amd import streams
    branch = streams.BranchStreams(dev)
    side = branch.context_model_stream(N, H, Wd)
    torch.cuda.synchronize(dev)
    torch.cuda.set_stream(branch.main)      # CU-range streams are blocking with respect to the legacy default stream

    def step():
        cur = torch.cuda.current_stream(dev)
        enc = ae.encode(x, is_training=False)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            bc = pc.bitcost(enc.qbar, enc.symbols, is_training=False, pad_value=pad_value)
            bpp = bits.bitcost_to_bpp(bc, x)
        branch.reserve_idle_cus(side is not branch._plain)     # the decoder's 3x3 launches leave the side stream's CUs alone
        x_out = ae.decode(enc.qhard, is_training=False)
        branch.reserve_idle_cus(False)
        cur.wait_stream(side)
        return bpp, x_out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        bpp, x_out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    pixels_per_step = N * H * Wd * world
    value = pixels_per_step * a.steps / elapsed / 1e6

    # ---- stage split and the dominant kernel, HIP events on the launch stream (rank 0) ----
    extra = {'context_model_stream_cus': branch.idle_cus(N, H, Wd) if side is not branch._plain else 0}
    roofline = None
    if rank == 0:
        st = _lib.current_stream(dev)
        ev = [ctypes.c_void_p() for _ in range(2)]
        for e in ev:
            _lib.check(lib.ic_event_create(ctypes.byref(e)))

        def timed(fn, reps):
            fn()
            torch.cuda.synchronize(dev)
            _lib.check(lib.ic_event_record(ev[0], st))
            for _ in range(reps):
                fn()
            _lib.check(lib.ic_event_record(ev[1], st))
            ms = ctypes.c_float()
            _lib.check(lib.ic_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms)))
            return ms.value / reps

        enc = ae.encode(x, False)
        ms_enc = timed(lambda: ae.encode(x, False), 5)
        ms_pc = timed(lambda: pc.bitcost(enc.qbar, enc.symbols, False, pad_value=pad_value), 5)
        ms_dec = timed(lambda: ae.decode(enc.qhard, False), 5)
        extra.update({'ms_encode': round(ms_enc, 4), 'ms_pc_bitcost': round(ms_pc, 4), 'ms_decode': round(ms_dec, 4),
                      'bpp_synthetic': round(float(bpp), 5)})
        # dominant kernel: the shape the residual stacks run at, (N,128,H/4,W/4)
        h4, w4 = H // 4, Wd // 4
        xin = torch.randn((N, 128, h4, w4), device=dev)
        res = torch.randn((N, 128, h4, w4), device=dev)
        yout = torch.empty_like(xin)
        wpk, sc, sh = ae._plan['autoencoder/encoder/res_block_enc_0/enc_0_1/conv2']
        def conv():
            _lib.check(lib.ic_conv3x3_c128_auto_f32(_lib.ptr(xin), _lib.ptr(wpk), _lib.ptr(sc), _lib.ptr(sh),
                                                    _lib.ptr(res), None, _lib.ptr(yout), N, h4, w4, 0, st))
        ms_conv = timed(conv, 64)
        flop = CONV3_FLOP_PER_OUT_PX * N * h4 * w4
        achieved = flop / (ms_conv * 1e-3) / 1e12
        wino = lib.ic_conv3x3_c128_pick_algo(N, h4, w4) == 1
        # 'achieved' counts the ALGORITHMIC (direct-form) multiply-adds of the layer; the Winograd
        # form issues 16/36 of them to the matrix cores -- 'executed_*' is what the MFMA pipe actually did.
        executed = achieved * (16.0 / 36.0 if wino else 1.0)
        # HBM-side bytes per launch: PMC counters cannot be read from inside this process; they are collected with
        # rocprofv3 --pmc on the same kernel and shape (tools/pmc_wino.sh) and committed under profiles/
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_conv3x3_traffic.json')) as f:
                tj = json.load(f)
            if tj['shape'] == [N, 128, h4, w4]:
                e = tj['winograd F(2x2,3x3)' if wino else 'direct']
                traffic, traffic_src = e['fetch_bytes'] + e['write_bytes'], tj['source']
        except (IOError, OSError, KeyError, ValueError):
            pass
        groups = N * (-(-h4 // 4)) * (-(-w4 // 32))
        t16 = wino and lib.ic_wino3x3_c128_workgroups(N, h4, w4) >= 256 > groups
        roofline = {'kernel': (('wino3x3_c128_t16_kernel' if t16 else 'wino3x3_c128_shared_kernel') if wino else 'conv3x3_c128_kernel') + ' (ic_conv3x3_c128_auto_f32)',
                    'algorithm': 'winograd F(2x2,3x3)' if wino else 'direct', 'bound': 'mfma',
                    'achieved': round(achieved, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4), 'traffic': traffic, 'traffic_unit': 'bytes per launch',
                    'traffic_source': traffic_src, 'algorithmic_bytes_per_launch': int(3 * 512 * N * h4 * w4 + (1048576 if wino else 589824)),
                    'executed_tflops': round(executed, 2), 'executed_frac': round(executed / PEAK_F32_MFMA_TFLOPS, 4),
                    'avg_launch_us': round(ms_conv * 1e3, 2), 'flop_per_launch': flop,
                    'launches_per_step': 2 * (6 * int(ae_cfg.arch_param_B) + 2)}
        for e in ev:
            lib.ic_event_destroy(e)
        # Extra, NOT the contract value: the same step with three independent batch-1 pipelines in flight (one stream and
        # one set of workspaces each).  A Kodak-sized 3x3 launch fills 768 of the 1024 SIMDs; kernels of the other
        # pipelines take the rest.  `value` above stays the strictly sequential single-stream number.
        try:
            pipes = [(ae, pc, torch.cuda.Stream(device=dev))]
            for _ in range(2):
                ae2 = autoencoder.get_network_cls(ae_cfg)(ae_cfg).load_weights(wts, dev)
                pc2 = probclass.get_network_cls(pc_cfg)(pc_cfg, num_centers=ae_cfg.num_centers).load_weights(wts, dev)
                pipes.append((ae2, pc2, torch.cuda.Stream(device=dev)))

            def one(aei, pci):
                e = aei.encode(x, is_training=False)
                b = pci.bitcost(e.qbar, e.symbols, is_training=False, pad_value=pad_value)
                bits.bitcost_to_bpp(b, x)
                return aei.decode(e.qhard, is_training=False)
            torch.cuda.synchronize(dev)
            for i in range(6):
                with torch.cuda.stream(pipes[i % 3][2]):
                    one(*pipes[i % 3][:2])
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            n_img = 36
            for i in range(n_img):
                with torch.cuda.stream(pipes[i % 3][2]):
                    one(*pipes[i % 3][:2])
            torch.cuda.synchronize(dev)
            dt3 = time.perf_counter() - t1
            extra['pipelined_3_streams'] = {'value': round(N * H * Wd * n_img / dt3 / 1e6, 3), 'unit': 'Mpix/s',
                                            'ms_per_image': round(dt3 / n_img * 1e3, 4), 'images': n_img,
                                            'note': 'three independent batch-1 pipelines in flight on one GPU; not the contract value'}
        except Exception as ex:                                       # informational only
            extra['pipelined_3_streams'] = {'error': str(ex)[:200]}

    # ---- CPU baseline: the oracle on the host cores (rank 0, N == 1 only) ----
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import oracle as O
        # torch's CPU conv kernels oversubscribe badly beyond a few dozen threads on the 256-core host
        # (tools/cpu_threads.py: 16 threads is the fastest setting measured on the MI355X box)
        cores = min(os.cpu_count() or 1, 16)
        torch.set_num_threads(cores)
        sample = x_np[:1]
        with torch.no_grad():
            O.validate_forward(sample[:, :, :64, :64], wts, ae_cfg.as_dict(), torch.float32)     # warm-up (primitive caches)
            runs, t1 = 0, time.perf_counter()
            while runs < 3 or (time.perf_counter() - t1 < 10.0 and runs < 200):
                O.validate_forward(sample, wts, ae_cfg.as_dict(), torch.float32)
                runs += 1
            dt = (time.perf_counter() - t1) / runs
        cpu = {'value': round(sample.shape[2] * sample.shape[3] / dt / 1e6, 4), 'unit': 'Mpix/s', 'cores': cores,
               'kind': 'port', 'sample': '{} x image 3x{}x{} through the torch-CPU fp32 oracle (encode + bitcost + '
               'decode), {} torch threads, {:.2f} s per image'.format(runs, sample.shape[2], sample.shape[3], cores, dt)}

    if rank == 0:
        C = int(ae_cfg.num_chan_bn)
        flop_step = N * H * Wd * (FLOP_PER_PX_ENC + FLOP_PER_PX_DEC + FLOP_PER_SYMBOL_PC * C / 64.0)
        out = {
            'metric': 'Megapixels/s encode+pc-logits (and decode) per node',
            'value': round(value, 3), 'unit': 'Mpix/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(elapsed / a.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: Kodak-shaped image {}x3x{}x{} per GPU per step, '
                                   'ae_configs/cvpr/{} + pc_configs/cvpr/res_shallow, encode + parallel '
                                   'context-model bitcost + decode(qhard); random-init weights'.format(
                                       N, H, Wd, a.ae_config),
                       'batch_per_gpu': N, 'height': H, 'width': Wd, 'parallelism': 'image-sharded x{}'.format(world),
                       'schedule': 'one image at a time; bitcost and decode of that image run concurrently (val.py:85-89 '
                                   'evaluates both in one session.run), the bitcost on {}'.format(
                                       'a stream limited to the {} CUs the decoder leaves idle'.format(extra['context_model_stream_cus'])
                                       if extra.get('context_model_stream_cus') else 'a second stream')},
            'model_tflops_per_s': round(flop_step * world * a.steps / elapsed / 1e12, 2),
            'roofline': roofline, 'cpu_baseline': cpu,
        }
        out.update(extra)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()                 # rank 0 is still timing the stage split / dominant kernel: leave together
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
