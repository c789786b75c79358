"""Soak of the shipped F(4x4) kernel under the conditions of round 4's failure: many launches, two work-groups per CU from one grid and
from different grids, every output compared bit for bit.   python tools/probes/soak.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from imgcomp_cvpr_amd import _lib
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device('cuda:0')
for n_flight in (4, 8):
    first = bench.Pipeline(dev, 'low', 'serial', seed=0).set_input(1, 512, 768)
    sched = bench.InFlight(torch, first, dev, n_flight, 'low', 0)
    refs = []
    for pl, st in zip(sched.pipes, sched.streams):
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            bpp, x_out = pl.step()
        torch.cuda.synchronize()
        enc, bc = pl.last
        refs.append([t.clone() for t in (enc.z, enc.symbols, bc, x_out)])
    mism = [torch.zeros((), dtype=torch.int64, device=dev) for _ in sched.pipes]
    t0 = time.perf_counter()
    for _ in range(rounds):
        for k in range(n_flight):
            bpp, x_out = sched.step()
            pl = sched.pipes[k]
            with torch.cuda.stream(sched.streams[k]):
                enc, bc = pl.last
                for got, want in zip((enc.z, enc.symbols, bc, x_out), refs[k]):
                    mism[k] += (got != want).sum()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = rounds * n_flight
    print('%d images in flight: %d steps = %d launches of wino4_3x3_kernel<128,128> (+ %d each of <256,128> and <128,256>) in %.1f s; values that differ from the serial run: %s'
          % (n_flight, steps, 64 * steps, steps, dt, [int(m) for m in mism]), flush=True)
