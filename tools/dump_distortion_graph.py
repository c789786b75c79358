"""dump the captured distortion graph (training._GraphedDistortion) as DOT and print its shape: nodes, edges, roots, leaves, forks"""
import os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from imgcomp_cvpr_amd import config_parser as cp, training

_orig = torch.cuda.CUDAGraph


class _Dbg(_orig):
    def __init__(self, *a, **k):
        super().__init__()
        self.enable_debug_mode()


torch.cuda.CUDAGraph = _Dbg
ae_cfg, _ = cp.parse(cp.builtin_config_path('ae_configs', 'cvpr', 'med'))
gd = training._GraphedDistortion(ae_cfg, (32, 3, 128, 128), torch.device('cuda', 0))
path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/dist_graph.dot'
gd.graph.debug_dump(path)
txt = open(path).read()
edges = re.findall(r'"?([\w.]+)"?\s*->\s*"?([\w.]+)"?', txt)
nodes = set(a for e in edges for a in e)
outd, ind = {}, {}
for a, b in edges:
    outd[a] = outd.get(a, 0) + 1
    ind[b] = ind.get(b, 0) + 1
print('nodes', len(nodes), 'edges', len(edges), 'roots', sum(1 for n in nodes if n not in ind), 'leaves', sum(1 for n in nodes if n not in outd),
      'forks', sum(1 for v in outd.values() if v > 1), 'joins', sum(1 for v in ind.values() if v > 1))
