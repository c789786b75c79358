#!/usr/bin/env python
"""Evidence for the power-limited matrix-core ceiling: sample the shader clock and socket power (rocm-smi) while one kernel
runs back to back -- the Kodak 3x3 Winograd layer (768 of 1024 SIMDs busy) against the h2 layer (all SIMDs busy).

    python tools/clock_probe.py
"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib as L

lib = L.lib
dev = torch.device('cuda:0'); st = L.current_stream()
g = torch.Generator(device='cpu').manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)


def smi():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=20).stdout
    except Exception as ex:          # noqa: BLE001
        return 'rocm-smi failed: %s' % ex
    sclk = re.findall(r'sclk clock level:?\s*\S*:?\s*\(?(\d+)Mhz\)?', out)
    pw = re.findall(r'Power \(W\):\s*([\d.]+)', out)
    return 'sclk %s MHz  power %s W' % (','.join(sclk[:1]) or '?', ','.join(pw[:1]) or '?')


def hammer(fn, seconds, label):
    stop = [False]
    samples = []

    def sampler():
        time.sleep(0.5)
        while not stop[0]:
            samples.append(smi())
            time.sleep(0.3)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        for _ in range(200):
            fn()
        torch.cuda.synchronize(); n += 200
    dt = time.time() - t0
    stop[0] = True; th.join()
    print('%-28s %.1f us per launch | %s' % (label, dt / n * 1e6, ' | '.join(samples[:4])))


# 3x3 Winograd layer, Kodak map
x = r(1, 128, 128, 192); y = torch.empty_like(x); res = r(1, 128, 128, 192)
wt = r(3, 3, 128, 128) * 0.05
ww = torch.empty(lib.ic_wino3x3_c128_packed_floats(), device=dev)
L.check(lib.ic_pack_wino3x3_c128_f32(L.ptr(wt), L.ptr(ww), 0, st))
sc, sh = torch.ones(128, device=dev), torch.zeros(128, device=dev)
for ks, label in ((0, '3x3 whole-K (768 SIMDs)'), (1, '3x3 K-split (1024 SIMDs)')):
    lib.ic_wino3x3_c128_set_tuning(2, ks)
    hammer(lambda: L.check(lib.ic_wino3x3_c128_bn_act_f32(L.ptr(x), L.ptr(ww), L.ptr(sc), L.ptr(sh), L.ptr(res), None, L.ptr(y), 1, 128, 192, 1, st)), 4.0, label)
lib.ic_wino3x3_c128_set_tuning(2, -1)
# h2
w = r(5, 5, 64, 128) * 0.05
wp = torch.empty(lib.ic_conv2d_mfma_packed_floats(5, 5, 64, 128, 2, 0), device=dev)
L.check(lib.ic_pack_conv2d_mfma_f32(L.ptr(w), L.ptr(wp), 5, 5, 64, 128, 2, 0, st))
x2 = r(1, 64, 256, 384); y2 = torch.empty(1, 128, 128, 192, device=dev)
hammer(lambda: L.check(lib.ic_conv2d_mfma_bn_act_f32(L.ptr(x2), L.ptr(wp), L.ptr(sc), L.ptr(sh), L.ptr(y2), 1, 64, 256, 384, 128, 5, 5, 2, 0, 1, st)), 4.0, 'h2 (1024 SIMDs, 3 waves each)')
print('idle: ' + smi())
