"""A/B of the context model's two inference passes on the bench image's real symbols: channels-last (probclass_cl.hip, default)
against planar (IC_PC_PLANAR=1) -- bit-identity of bits and logits, and time per pass.   python tools/pc_ab.py [reps] [H W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import autoencoder, probclass, config_parser as cp, weights as W
dev = torch.device('cuda:0')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
H, Wd = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 768)
ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
wts = W.synthetic_weights(ae_cfg, pc_cfg)
ae = autoencoder.get_network_cls(ae_cfg)(ae_cfg).load_weights(wts, dev)
pc = probclass.get_network_cls(pc_cfg)(pc_cfg, num_centers=ae_cfg.num_centers).load_weights(wts, dev)
x = torch.as_tensor(W.synthetic_image((1, 3, H, Wd), 'natural', seed=0)).float().to(dev)
enc = ae.encode(x, is_training=False)
pad = float(wts['autoencoder/encoder/centers'][0])


def run(planar):
    os.environ['IC_PC_PLANAR'] = '1' if planar else '0'
    bc = pc.bitcost(enc.qbar, enc.symbols, False, pad_value=pad).clone()
    lg = pc.logits_all(enc.qbar, pad_value=pad).clone() if hasattr(pc, 'logits_all') else None
    torch.cuda.synchronize()
    for _ in range(5):
        pc.bitcost(enc.qbar, enc.symbols, False, pad_value=pad)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        pc.bitcost(enc.qbar, enc.symbols, False, pad_value=pad)
    torch.cuda.synchronize()
    return bc, lg, (time.perf_counter() - t0) / reps * 1e6


for rnd in range(2):
    bc_cl, lg_cl, us_cl = run(False)
    bc_pl, lg_pl, us_pl = run(True)
    print('round {}: channels-last {:.1f} us   planar {:.1f} us   bits identical: {}   sum bits {:.3f}'.format(
        rnd, us_cl, us_pl, bool(torch.equal(bc_cl, bc_pl)), float(bc_cl.sum())), flush=True)
    if not torch.equal(bc_cl, bc_pl):
        d = (bc_cl - bc_pl).abs()
        print('   max |diff| {:.3e} at {}  mismatches {}'.format(float(d.max()), int(d.argmax()), int((d > 0).sum())))
