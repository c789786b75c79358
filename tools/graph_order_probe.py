"""Is work enqueued behind a HIP graph launch ordered after the graph's last node?  (training hazard, round 4)

Captures a dependent chain of n small kernels (y = y * 1 + 1, n times) with torch.cuda.CUDAGraph, then -- with the host far ahead
of the device -- replays it and immediately enqueues a consumer of the result on the same stream (and, second variant, on
another stream behind an event).  A correctly ordered launch yields exactly `n` every time."""
import sys

import torch


def probe(n, reps=20, busy=64, numel=1 << 20, matmul=False, backward=False):
    dev = torch.device('cuda', 0)
    buf = torch.zeros(numel, device=dev)
    out = torch.zeros(numel, device=dev)
    A = torch.eye(256, device=dev)
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))

    def body():
        y = buf
        if backward:
            y = y.detach().requires_grad_(True)
            z = y
            for _ in range(n):
                z = z * 1.0 + 1.0
            g, = torch.autograd.grad(z.sum(), y)
            out.copy_(z.detach() + g - 1.0)
            return
        for i in range(n):
            y = y * 1.0 + 1.0
            if matmul and i % 8 == 0:
                y = (y.view(-1, 256) @ A).view(-1)
        out.copy_(y)
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        body()
    torch.cuda.synchronize()
    big = torch.randn(4096, 4096, device=dev)
    bad_same = bad_other = 0
    side = torch.cuda.Stream(dev)
    results = []
    for r in range(reps):
        for _ in range(busy):                       # put the host ahead of the device
            big2 = big @ big
        out.zero_()
        g.replay()
        results.append(out.clone())                  # consumer on the same stream, right behind the launch
    torch.cuda.synchronize()
    bad_same = sum(int(not bool((t == n).all())) for t in results)
    results = []
    for r in range(reps):
        for _ in range(busy):
            big2 = big @ big
        out.zero_()
        g.replay()
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            results.append(out.clone())              # consumer on another stream behind an event
        torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    bad_other = sum(int(not bool((t == n).all())) for t in results)
    return bad_same, bad_other


if __name__ == '__main__':
    print(torch.__version__, torch.version.hip)
    for kw in ({}, {'matmul': True}, {'backward': True}):
        for n in (8, 64, 256, 600, 1500):
            print(kw, 'chain', n, '-> wrong results (same stream, other stream) of 20:', probe(n, **kw), flush=True)
