"""BASELINE configs[4]: a few steps of the hi + res_shallow path on one 3840 x 2160 frame (for rocprofv3: tools/profile.sh trace cfg5 python tools/run_cfg5.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda:0')
p = bench.Pipeline(dev, 'hi', 'serial').set_input(1, 2160, 3840)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    p.step()
torch.cuda.synchronize()
