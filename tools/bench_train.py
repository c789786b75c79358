#!/usr/bin/env python
"""Training-step throughput (BASELINE configs[2]: ae_configs/cvpr/med + res_shallow, random 128x128 crops, batch 32).
  python tools/bench_train.py [--batch 32] [--crop 128] [--steps 10]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... tools/bench_train.py   (global batch split over ranks)
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import config_parser as cp, training, weights as W


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--batch', type=int, default=32)
    p.add_argument('--crop', type=int, default=128)
    p.add_argument('--steps', type=int, default=10)
    p.add_argument('--warmup', type=int, default=2)
    p.add_argument('--distortion', default='ms_ssim')
    p.add_argument('--bn_chunk', type=int, default=0)
    a = p.parse_args()
    if a.bn_chunk:
        from imgcomp_cvpr_amd import _lib as _l
        _l.lib.ic_bn_set_tuning(a.bn_chunk)
    rank = int(os.environ.get('RANK', '0')); world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=dev)
    ae, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'med'))
    pc, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
    ae.distortion_to_minimize = a.distortion
    tr = training.Trainer(ae, pc, W.synthetic_weights(ae, pc), dev)
    nb = a.batch // world
    x = torch.as_tensor(W.synthetic_image((nb, 3, a.crop, a.crop), 'natural', rank)).float().to(dev)
    for _ in range(a.warmup):
        out = tr.step(x)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = tr.step(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    if rank == 0:
        flop = a.batch * a.crop * a.crop * (621124 + 618976 + 23760) * 3.0
        print(json.dumps({'metric': 'training step, Megapixels/s', 'value': round(a.batch * a.crop * a.crop / dt / 1e6, 3),
                          'ms_per_step': round(dt * 1e3, 3), 'img_per_s': round(a.batch / dt, 1), 'n_gpus': world,
                          'global_batch': a.batch, 'crop': a.crop, 'model_tflops_per_s (fwd+bwd ~ 3x fwd)': round(flop / dt / 1e12, 1),
                          'last': {k: round(v, 4) for k, v in out.items()}}))


if __name__ == '__main__':
    main()
