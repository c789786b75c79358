"""cProfile of val.validate() on 48 Kodak-sized PNGs (4 in flight, 8 loader threads): where the host's time per image goes."""
import os, sys, tempfile, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from imgcomp_cvpr_amd import val, config_parser as cp, weights as W
ae, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
pc, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
wts = W.synthetic_weights(ae, pc)
with tempfile.TemporaryDirectory() as d:
    paths = []
    for i in range(48):
        x = W.synthetic_image((1, 3, 512, 768), 'natural', seed=i)[0]
        p = os.path.join(d, 'img{:02d}.png'.format(i))
        Image.fromarray(np.transpose(np.clip(x, 0, 255).astype(np.uint8), (1, 2, 0))).save(p)
        paths.append(p)
    flags = val.OutputFlags(save_ours=False, ckpt_step=-1, real_bpp=False)
    out = os.path.join(d, 'out'); os.makedirs(out)
    val.validate(ae, pc, wts, paths[:4], out, flags, verbose=False)
    pr = cProfile.Profile(); pr.enable()
    val.validate(ae, pc, wts, paths, out, flags, verbose=False)
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(14)
