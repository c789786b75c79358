"""How long the HOST is held by an upload of one Kodak-sized uint8 image while the device has work queued: from pageable memory on the
compute stream (waits for the queued work), from page-locked / registered memory, from pageable memory on a stream of its own
(what val.Fetcher.enqueue does).  Round 4: 9.5 / 2.8 ms (blocking / non_blocking pageable), 0.08 ms pinned, 0.30 ms side stream."""
import time, torch
dev = torch.device('cuda')
img = torch.randint(0, 255, (1, 3, 512, 768), dtype=torch.uint8)
busy = torch.randn(4096, 4096, device=dev)


def t(fn, n=30, load=True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        if load:
            for _ in range(3):
                busy @ busy          # ~ms of queued device work
        fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return th / n * 1e3


for load in (False, True):
    print('device busy' if load else 'device idle')
    print('  pageable .to:              %.3f ms host' % t(lambda: img.to(dev), load=load))
    print('  pageable .to non_blocking: %.3f ms host' % t(lambda: img.to(dev, non_blocking=True), load=load))
    pin = torch.empty_like(img).pin_memory()
    print('  copy into pin_memory():    %.3f ms host' % t(lambda: pin.copy_(img), load=False))
    print('  pinned .to non_blocking:   %.3f ms host' % t(lambda: pin.to(dev, non_blocking=True), load=load))
    s2 = torch.cuda.Stream()

    def on_side():
        with torch.cuda.stream(s2):
            return img.to(dev)
    print('  pageable .to on a side stream: %.3f ms host' % t(on_side, load=load))
