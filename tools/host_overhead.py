"""How long does the HOST need to enqueue one step (encode + bitcost + bpp + decode)?  If that is close to the GPU time of the step,
small images are host-bound and more images in flight cannot help.   python tools/host_overhead.py [H W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda:0')
H, W = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (256, 256)
p = bench.Pipeline(dev, 'low', 'serial', seed=0).set_input(1, H, W)
for _ in range(5):
    p.step()
torch.cuda.synchronize()
# short bursts: the host must not run into a full launch queue (200 steps queued back to back measure the GPU, not the host)
n, host, total = 4, [], []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        p.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) / n * 1e6)
    total.append((t2 - t0) / n * 1e6)
host.sort(); total.sort()
print('{}x{}: host enqueue {:.1f} us per step (median of 20 bursts of {}), GPU-complete {:.1f} us per step (one stream)'.format(
    H, W, host[10], n, total[10]))
