"""-m gpu: bench.py as the driver launches it -- including N > 1, which round 2 shipped with a rank-asymmetric barrier that
would have hung every multi-GPU point.  gpurun boxes have ONE GPU, so the two ranks of these tests share cuda:0 and the process
group is gloo (`--backend gloo --device 0`); the collective sequence of every rank is the same as under RCCL."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(nproc, port, extra, timeout=600):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    if nproc == 1:
        cmd = [sys.executable, 'bench.py', '--gpus', '1'] + extra
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1',
               '--master-port', str(port), 'bench.py', '--gpus', str(nproc), '--backend', 'gloo', '--device', '0'] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, universal_newlines=True)
    assert p.returncode == 0, 'bench.py rc {}\n{}\n{}'.format(p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, 'expected ONE JSON line on rank 0, got {}: {}'.format(len(lines), p.stdout[-2000:])
    return json.loads(lines[0])


def test_bench_two_ranks_inference(cuda):
    """`bench.py --gpus 2`: returns, rank 0 prints one line, the rank-0-only extras (stage split, roofline, other shapes) run
    without any process-group call while rank 1 waits at the final barrier."""
    out = _launch(2, 29561, ['--steps', '3', '--warmup', '1', '--height', '128', '--width', '192'])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['scaling'] == 'weak' and out['value'] > 0
    assert out['roofline'] is not None and 0 < out['roofline']['frac'] <= 1
    assert out['cpu_baseline'] is None                              # N = 1 only
    assert len(out['shapes']) == 2
    # the line verifies its own world: every rank took part in a collective, and where each of them ran
    assert out['ranks']['rccl_ranks_seen'] == 2 and out['ranks']['dist_world_size'] == 2 and len(out['ranks']['devices']) == 2


def test_bench_two_ranks_training_weak_and_strong(cuda):
    """`bench.py --mode train --gpus 2`: the data-parallel step (bucketed gradient all-reduce, cross-replica BatchNorm) under
    both scalings; strong = cfg3's batch of 32 split 2 x 16."""
    weak = _launch(2, 29562, ['--mode', 'train', '--steps', '2', '--warmup', '1'])
    assert weak['n_gpus'] == 2 and weak['scaling'] == 'weak' and weak['config']['batch_per_gpu'] == 32
    strong = _launch(2, 29563, ['--mode', 'train', '--steps', '2', '--warmup', '1', '--scaling', 'strong'])
    assert strong['scaling'] == 'strong' and strong['config']['batch_per_gpu'] == 16 and strong['config']['global_batch'] == 32
    assert all(v == v for v in strong['last_step'].values())          # finite losses


def test_bench_launches_itself_for_more_than_one_gpu(cuda):
    """`python bench.py --gpus 2` WITHOUT the torch.distributed.run wrapper (round 3 died on `assert world == a.gpus`): bench.py
    becomes the launcher, the two ranks share cuda:0 over gloo, rank 0 prints the one line with n_gpus = 2."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--backend', 'gloo', '--device', '0', '--steps', '3', '--warmup', '1',
           '--height', '128', '--width', '192', '--no_extras']
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, universal_newlines=True)
    assert p.returncode == 0, 'rc {}\n{}\n{}'.format(p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['value'] > 0 and out['steps'] == 3


def test_bench_single_rank_contract(cuda):
    """the default launch: one JSON line carrying the contract's keys plus roofline / cpu_baseline"""
    out = _launch(1, 0, ['--steps', '3', '--warmup', '1', '--height', '128', '--width', '192'])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in out, k
    assert out['cpu_baseline']['kind'] == 'port' and out['cpu_baseline']['value'] > 0
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'from_profiles')) <= set(out['roofline'])
    # the training number the driver's BENCH file carries (cfg3 on this GPU) with its own sanity checks
    tr = out['train']
    assert 'error' not in tr, tr
    assert tr['value'] > 0 and tr['checks']['ms_ssim_in_unit_interval'] and tr['checks']['d_loss_is_K_times_one_minus_ms_ssim']


@pytest.mark.parametrize('n_flight,rounds', [(4, 16), (8, 26)])
def test_in_flight_schedule_is_bit_identical_to_serial(cuda, n_flight, rounds):
    """The schedule BENCH's `value` is measured on (bench.InFlight, default flags: n Kodak-sized pipelines on n streams, F(4x4) 3x3
    layers with two work-groups of DIFFERENT grids per CU, h2 / h12 in their <256,128> / <128,256,SHUF> forms, context model and edge
    kernels co-resident) gives, for every image and every step, exactly the z / symbols / bit costs / bpp / x_out the SAME pipeline
    gives alone on one stream with the same plan flags (val.py:157-158: one image per sess.run is what is being reproduced).
    Round 4's failure (wrong lanes under shared SIMDs; root cause: profiles/r05_w4_rootcause.md) lived exactly here and no test looked.
    Comparisons are enqueued on the pipelines' own streams (mismatch counters on the device), so the images stay in flight:
    n_flight x rounds steps = 64 / 208 steps, each with 64 launches of the 128 -> 128 kernel and one of each 5x5-as-phases form."""
    sys.path.insert(0, ROOT)
    import bench
    from imgcomp_cvpr_amd import _lib
    first = bench.Pipeline(cuda, 'low', 'serial', seed=0).set_input(1, 512, 768)
    sched = bench.InFlight(torch, first, cuda, n_flight, 'low', 0)
    # the plan really is the one under test: F(4x4) for the Kodak map with n launches in flight
    assert int(_lib.lib.ic_conv3x3_c128_pick_form(1, 128, 192, sched.pipes[0].ae.plan_flags)) == 2
    refs = []
    for pl, st in zip(sched.pipes, sched.streams):
        torch.cuda.synchronize()
        with torch.cuda.stream(st):                      # alone on the chip: nothing else is enqueued anywhere
            bpp, x_out = pl.step()
        torch.cuda.synchronize()
        enc, bc = pl.last
        refs.append([t.clone() for t in (enc.z, enc.symbols, enc.qhard, bc, bpp.reshape(-1), x_out)])
    assert not torch.equal(refs[0][5], refs[1][5])       # different images per pipeline
    mism = [torch.zeros((), dtype=torch.int64, device=cuda) for _ in sched.pipes]
    for _ in range(rounds):
        for k in range(n_flight):
            bpp, x_out = sched.step()
            pl = sched.pipes[k]
            with torch.cuda.stream(sched.streams[k]):
                enc, bc = pl.last
                for got, want in zip((enc.z, enc.symbols, enc.qhard, bc, bpp.reshape(-1), x_out), refs[k]):
                    mism[k] += (got != want).sum()
    torch.cuda.synchronize()
    bad = [int(m) for m in mism]
    assert bad == [0] * n_flight, 'values that differ from the serial run, per image over {} steps each: {}'.format(rounds, bad)
