"""val.validate() with the PNG decoding taken out (images handed over from memory): what the loop itself sustains -- round 4: 331 images/s
one at a time, 481 with 2 or 4 in flight (189 Mpix/s): host-bound at ~2.1 ms per image; with decoding on 8 threads 414 (tools/val_throughput.py)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from imgcomp_cvpr_amd import val, config_parser as cp, weights as W
n = 192
ae, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
pc, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
wts = W.synthetic_weights(ae, pc)
imgs = {}
with tempfile.TemporaryDirectory() as d:
    paths = []
    for i in range(n):
        p = os.path.join(d, 'img{:03d}.png'.format(i)); paths.append(p)
        imgs[p] = np.clip(W.synthetic_image((1, 3, 512, 768), 'natural', seed=i % 8)[0], 0, 255).astype(np.uint8)
    val.load_image_chw = lambda p, pad: imgs[p]           # no decoding at all: the loop's own ceiling
    flags = val.OutputFlags(save_ours=False, ckpt_step=-1, real_bpp=False)
    for in_flight in (1, 2, 4, 6):
        out = os.path.join(d, 'o%d' % in_flight); os.makedirs(out)
        val.validate(ae, pc, wts, paths[:8], out, flags, verbose=False, in_flight=in_flight, loader_threads=1)
        t0 = time.perf_counter()
        val.validate(ae, pc, wts, paths, out, flags, verbose=False, in_flight=in_flight, loader_threads=1)
        dt = time.perf_counter() - t0
        print('no decoding, in flight %d: %.1f images/s = %.1f Mpix/s' % (in_flight, n / dt, n * 512 * 768 / dt / 1e6), flush=True)
