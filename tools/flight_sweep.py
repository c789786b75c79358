"""the bench step (encode + context model + decode of one batch per step) with n independent pipelines in flight, per shape and plan:
how many images in flight fill the chip for small images.   python tools/flight_sweep.py [HxWxB ...]"""
import os, sys, json, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from imgcomp_cvpr_amd import _lib

dev = torch.device('cuda:0')
shapes = [tuple(int(v) for v in s.split('x')) for s in sys.argv[1:]] or [(256, 256, 1), (384, 512, 1), (512, 768, 1)]
for (h, w, b) in shapes:
    first = bench.Pipeline(dev, 'low', 'serial', seed=0).set_input(b, h, w)
    res = {}
    for plan, flags in (('auto', 0), ('f2x2', _lib.CONV3_NO_WINO4), ('f4x4', _lib.CONV3_WINO4)):
        for n in (1, 2, 4, 6, 8, 12, 15):
            first.ae.plan_flags = flags
            sch = bench.InFlight(torch, first, dev, n, 'low', 0) if n > 1 else first
            k = max(30, 6 * n)
            for _ in range(n + 2):
                sch.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                sch.step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res['{} n={}'.format(plan, n)] = round(b * h * w * k / dt / 1e6, 1)
            form = _lib.lib.ic_conv3x3_c128_pick_form(b, h // 4, w // 4, flags | (_lib.CONV3_IN_FLIGHT(n) if n > 1 else 0))
            res['{} n={} form'.format(plan, n)] = int(form)
            del sch
    print(json.dumps({'shape': [b, 3, h, w], 'mpix_s': res}), flush=True)
