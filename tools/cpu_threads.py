"""host-side probe: how does the torch-CPU oracle scale with threads on this box?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import config_parser as cp, weights as W
from oracle import oracle as O
ae, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
pc, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
w = W.synthetic_weights(ae, pc)
x = W.synthetic_image((1, 3, 256, 256), 'natural', 0)
print('cores', os.cpu_count())
for th in (8, 16, 32, 64, 128):
    if th > (os.cpu_count() or 1):
        break
    torch.set_num_threads(th)
    with torch.no_grad():
        O.validate_forward(x[:, :, :64, :64], w, ae.as_dict())
        t = time.perf_counter(); O.validate_forward(x, w, ae.as_dict()); dt = time.perf_counter() - t
    print('threads', th, 'sec %.2f' % dt, 'Mpix/s %.4f' % (256 * 256 / dt / 1e6), flush=True)
