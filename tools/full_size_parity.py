"""BASELINE configs[1] at its full size against the float64 oracle (too slow for the test-suite: ~1 min of CPU): one Kodak-shaped image
through the plans the benchmark's step runs -- F(4x4) for the 3x3 layers and h2 / h12 (as with images in flight) and the one-at-a-time
plan -- z, heatmap, symbol flips, bit cost and bpp, x_out.      python tools/full_size_parity.py [ae_config H W]
(`hi 2160 3840` = BASELINE configs[4] on a whole 4K frame: ~2 min of float64 oracle on 16 host threads)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from imgcomp_cvpr_amd import autoencoder, probclass, bits, config_parser as cp, weights as W, _lib
from oracle import oracle as O

torch.set_num_threads(16)
dev = torch.device('cuda:0')
AE, H, Wd = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ('low', 512, 768)
ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', AE))
pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
wts = W.synthetic_weights(ae_cfg, pc_cfg)
ae = autoencoder.get_network_cls(ae_cfg)(ae_cfg).load_weights(wts, dev)
pc = probclass.get_network_cls(pc_cfg)(pc_cfg, num_centers=ae_cfg.num_centers).load_weights(wts, dev)
x = W.synthetic_image((1, 3, H, Wd), 'natural', seed=0)
t0 = time.time()
with torch.no_grad():
    ref = O.encode(torch.as_tensor(x).double(), wts, ae_cfg.as_dict())
    centers = wts['autoencoder/encoder/centers']
    rb, _ = O.bitcost(ref.qhard, ref.symbols, wts, float(centers[0]))
    ref_xo = O.decode(ref.qhard, wts, ae_cfg.as_dict())
print('float64 oracle: %.0f s' % (time.time() - t0), flush=True)


def rel(a, b):
    return float((a.double().cpu() - b).abs().max()) / max(1.0, float(b.abs().max()))


xd = torch.as_tensor(x).float().to(dev)
for name, flags in (('images-in-flight plan (F(4x4) 3x3 layers, h2, h12)', _lib.CONV3_IN_FLIGHT(4)), ('one-at-a-time plan (F(2x2) 3x3 layers, F(4x4) h2 / h12)', 0),
                    ('all direct', _lib.CONV3_DIRECT | _lib.CONV5_NO_WINO4)):
    enc = ae.encode(xd, False, plan_flags=flags)
    bc = pc.bitcost(enc.qbar, enc.symbols, False, pad_value=pc.auto_pad_value(ae))
    xo = ae.decode(dev_q := ref.qhard.float().to(dev), False, plan_flags=flags)
    torch.cuda.synchronize()
    flips = int((enc.symbols.cpu() != ref.symbols).sum())
    same = (enc.symbols.cpu() == ref.symbols)
    bpp = float(bits.bitcost_to_bpp(bc, xd))
    print('%s:\n   z %.2e  heatmap %.2e  symbol flips %d of %d  x_out (reference symbols) %.2e  bpp %.6f (oracle %.6f)' % (
        name, rel(enc.z, ref.z), rel(enc.heatmap, ref.heatmap), flips, ref.symbols.numel(), rel(xo, ref_xo), bpp,
        O.bitcost_to_bpp(rb, torch.as_tensor(x))), flush=True)
