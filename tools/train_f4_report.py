"""cfg3 training step (32 x 3 x 128 x 128, cvpr/med + res_shallow, MS-SSIM) with the 3x3 layers in F(2x2) / F(4x4) per direction:
gradient errors of the tensors tests/test_gpu_configs.py checks against the float64 autograd of the oracle, and the step time.
   python tools/train_f4_report.py [modes...]      modes: 0 fwd bwd 1   -> gpurun_out/train_f4_report.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from imgcomp_cvpr_amd import training, config_parser as cp, weights as W
from oracle import train_oracle as T
from tests import util
from tests.test_gpu_configs import CFG3_GRAD_RTOL, CFG3_GRAD_RTOL_DEFAULT

modes = sys.argv[1:] or ['0', 'fwd', 'bwd', '1']
dev = torch.device('cuda:0')
ae, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'med'))
pc, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
ae.H_target = 0.5
wts = W.synthetic_weights(ae, pc)
x = W.synthetic_image((32, 3, 128, 128), 'natural', 7)
torch.set_num_threads(16)
total, comps, p = T.train_loss(x, wts, ae.as_dict(), pc.as_dict(), torch.float64)
total.backward()
names = [n for n in p if p[n].grad is not None]
out = {}
xd = torch.as_tensor(x).float().to(dev)
for m in modes:
    g = training.TrainGraph(ae, pc, wts, dev, wino4=m)
    g.forward_backward(xd)
    torch.cuda.synchronize()
    flips = int((g.last['symbols'].cpu() != comps['symbols']).sum())
    errs = {n: util.rel_err(g.grads[n], p[n].grad) for n in names if n in g.grads}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
    over = {n: (e, CFG3_GRAD_RTOL.get(n, None)) for n, e in errs.items() if n in CFG3_GRAD_RTOL and e > CFG3_GRAD_RTOL[n]}
    zerr = util.rel_err(g.last['z'], comps['z'].detach())
    for _ in range(3):
        g.forward_backward(xd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        g.forward_backward(xd)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 8 * 1e3
    out[m] = {'forward_backward_ms': round(ms, 3), 'symbol_flips': flips, 'z_rel_err': zerr, 'worst': worst,
              'checked_tensors': {n: errs[n] for n in CFG3_GRAD_RTOL if n in errs}, 'over_their_test_bound': over,
              'max_over_all_tensors': max(errs.values()), 'median_over_all_tensors': float(np.median(list(errs.values())))}
    print(m, json.dumps(out[m])[:1500], flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/train_f4_report.json', 'w'), indent=1)
