"""val.py's own loop on 24 Kodak-sized PNGs (synthetic images, synthetic weights): images per second end to end -- PNG decode on the
host, encode + context model + decode + MS-SSIM / PSNR on the device, measures.csv -- for the decode-ahead thread pool and the number
of images in flight.      python tools/val_throughput.py [n_images]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from imgcomp_cvpr_amd import val, config_parser as cp, weights as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ae, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
pc, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
wts = W.synthetic_weights(ae, pc)
with tempfile.TemporaryDirectory() as d:
    paths = []
    for i in range(n):
        x = W.synthetic_image((1, 3, 512, 768), 'natural', seed=i)[0]
        p = os.path.join(d, 'img{:02d}.png'.format(i))
        Image.fromarray(np.transpose(np.clip(x, 0, 255).astype(np.uint8), (1, 2, 0))).save(p)
        paths.append(p)
    flags = val.OutputFlags(save_ours=False, ckpt_step=-1, real_bpp=False)
    t0 = time.perf_counter()
    for p in paths:
        val.load_image_chw(p, 8)
    print('PNG decode alone: %.1f ms per image' % ((time.perf_counter() - t0) / n * 1e3))
    for in_flight, threads in ((1, 1), (4, 1), (1, 8), (4, 8), (4, 16)):
        out = os.path.join(d, 'out_{}_{}'.format(in_flight, threads))
        os.makedirs(out)
        val.validate(ae, pc, wts, paths[:4], out, flags, verbose=False, in_flight=in_flight, loader_threads=threads)      # set-up pass
        t0 = time.perf_counter()
        avg = val.validate(ae, pc, wts, paths, out, flags, verbose=False, in_flight=in_flight, loader_threads=threads)
        dt = time.perf_counter() - t0
        print('in flight %d, loader threads %2d: %.1f images/s = %.1f Mpix/s   (%s)' % (in_flight, threads, n / dt, n * 512 * 768 / dt / 1e6, avg), flush=True)
