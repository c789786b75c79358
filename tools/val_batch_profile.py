"""where a batched val.py step spends its time: sections of Fetcher.enqueue / collect for one batch of 8 256 x 256 images, each followed by
a device synchronize (host wall clock), against 8 single-image steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from imgcomp_cvpr_amd import val, config_parser as cp, weights as W, bits, metrics, _lib

ae_c, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
pc_c, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
wts = W.synthetic_weights(ae_c, pc_c)
f = val.Fetcher(ae_c, pc_c, wts, 'cuda', plan_flags=_lib.CONV3_IN_FLIGHT(4))
imgs = [W.synthetic_image((1, 3, 256, 256), 'natural', seed=i)[0] for i in range(8)]
sync = torch.cuda.synchronize


def timed(label, fn, reps=5):
    fn(); sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    sync()
    print('%-46s %8.3f ms' % (label, (time.perf_counter() - t0) / reps * 1e3), flush=True)
    return out


timed('8 single-image steps (enqueue + collect)', lambda: [f.collect(f.enqueue(im)) for im in imgs])
timed('1 batched step of 8 (enqueue + collect)', lambda: f.collect(f.enqueue(imgs)))
xb = torch.as_tensor(np.stack(imgs)).to('cuda')
x = xb.float()
enc = timed('  encode batch 8', lambda: f.ae.encode(x, is_training=False))
bc = timed('  bitcost batch 8', lambda: f.pc.bitcost(enc.qbar, enc.symbols, is_training=False, pad_value=f.pc.auto_pad_value(f.ae)))
xo = timed('  decode batch 8', lambda: f.ae.decode(enc.qhard, is_training=False))
xo8 = xo.to(torch.uint8)
timed('  8 x bitcost_to_bpp on slices', lambda: [bits.bitcost_to_bpp(bc[i:i + 1], x[i:i + 1]) for i in range(8)])
timed('  8 x val_metrics_device on slices', lambda: [metrics.val_metrics_device(xb[i:i + 1], xo8[i:i + 1], f._metrics_ws) for i in range(8)])
timed('  page-locked allocation of a batch buffer', lambda: torch.empty((8, 3, 256, 256), dtype=torch.uint8, pin_memory=True))
timed('  page-locked allocation of a Kodak image', lambda: torch.empty((3, 512, 768), dtype=torch.uint8, pin_memory=True))
x1 = x[:1].contiguous()
timed('  encode batch 1', lambda: f.ae.encode(x1, is_training=False))

