"""val.py's own loop on 24 Kodak-sized PNGs (synthetic images, synthetic weights): images per second end to end -- PNG decode on the
host, encode + context model + decode + MS-SSIM / PSNR on the device, measures.csv -- for the decode-ahead thread pool and the number
of images in flight.      python tools/val_throughput.py [n_images]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from imgcomp_cvpr_amd import val, config_parser as cp, weights as W
import imgcomp_cvpr_amd
imgcomp_cvpr_amd.ask_for_hardware_queues(8)        # as val.main() does before the HIP runtime starts (without it: 4 queues, 407 instead of 496 images/s)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
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
    for in_flight, threads in ((1, 1), (4, 1), (4, 8), (4, 16), (4, 24)):
        out = os.path.join(d, 'out_{}_{}'.format(in_flight, threads))
        os.makedirs(out)
        val.validate(ae, pc, wts, paths[:4], out, flags, verbose=False, in_flight=in_flight, loader_threads=threads)      # set-up pass
        t0 = time.perf_counter()
        avg = val.validate(ae, pc, wts, paths, out, flags, verbose=False, in_flight=in_flight, loader_threads=threads)
        dt = time.perf_counter() - t0
        print('in flight %d, loader threads %2d: %.1f images/s = %.1f Mpix/s   (%s)' % (in_flight, threads, n / dt, n * 512 * 768 / dt / 1e6, avg), flush=True)
    # the loop alone: images already decoded (host uint8 in), what the decoders would have to keep up with
    imgs = [val.load_image_chw(p, 8) for p in paths]
    f = val.Fetcher(ae, pc, wts, 'cuda', plan_flags=0)
    fs = [f] + [val.Fetcher(ae, pc, wts, 'cuda', share_with=f) for _ in range(3)]
    from collections import deque
    for rep in range(2):
        pend = deque()
        t0 = time.perf_counter()
        for k, im in enumerate(imgs):
            if len(pend) == 4:
                ff, h = pend.popleft(); ff.collect(h)
            pend.append((fs[k % 4], fs[k % 4].enqueue(im)))
        while pend:
            ff, h = pend.popleft(); ff.collect(h)
        dt = time.perf_counter() - t0
    print('decoded images (host uint8 in), 4 in flight: %.1f images/s = %.1f Mpix/s' % (n / dt, n * 512 * 768 / dt / 1e6), flush=True)
# a directory of small images (256 x 256): consecutive same-shape images as one batch of 8 (val.py --batch_same_shape) against one image per step
with tempfile.TemporaryDirectory() as d:
    paths = []
    for i in range(4 * n):
        x = W.synthetic_image((1, 3, 256, 256), 'natural', seed=1000 + i)[0]
        p = os.path.join(d, 'img{:03d}.png'.format(i))
        Image.fromarray(np.transpose(np.clip(x, 0, 255).astype(np.uint8), (1, 2, 0))).save(p)
        paths.append(p)
    flags = val.OutputFlags(save_ours=False, ckpt_step=-1, real_bpp=False)
    for batch in (1, 8):
        out = os.path.join(d, 'out_b{}'.format(batch))
        os.makedirs(out)
        val.validate(ae, pc, wts, paths[:32], out, flags, verbose=False, in_flight=4, loader_threads=val.default_loader_threads(), batch_same_shape=batch)
        t0 = time.perf_counter()
        avg = val.validate(ae, pc, wts, paths, out, flags, verbose=False, in_flight=4, loader_threads=val.default_loader_threads(), batch_same_shape=batch)
        dt = time.perf_counter() - t0
        print('256 x 256 PNGs, batch_same_shape %d: %.1f images/s = %.1f Mpix/s   (%s)' % (batch, len(paths) / dt, len(paths) * 65536 / dt / 1e6, avg), flush=True)
