"""The context model alone, back to back, on the symbols a real encoder produces for bench.py's image -- the target of the
standalone rocprofv3 trace of the round (tools/profile_round.sh).   python tools/run_pc.py [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import autoencoder, probclass, config_parser as cp, weights as W
dev = torch.device('cuda:0')
ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'low'))
pc_cfg, _ = cp.parse(cp.builtin_config_path('pc_configs', 'cvpr', 'res_shallow'))
wts = W.synthetic_weights(ae_cfg, pc_cfg)
ae = autoencoder.get_network_cls(ae_cfg)(ae_cfg).load_weights(wts, dev)
pc = probclass.get_network_cls(pc_cfg)(pc_cfg, num_centers=ae_cfg.num_centers).load_weights(wts, dev)
x = torch.as_tensor(W.synthetic_image((1, 3, 512, 768), 'natural', seed=0)).float().to(dev)       # bench.py: rank 0's image
enc = ae.encode(x, is_training=False)
pad = float(wts['autoencoder/encoder/centers'][0])
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 50):
    pc.bitcost(enc.qbar, enc.symbols, False, pad_value=pad)
torch.cuda.synchronize()
