"""the two forms of the Winograd filter-gradient kernel on the cfg3 layer: time per call (kernel + slice reduction) and a checksum of dW
   (bit-identical by construction: same K order per accumulator).   IMGCOMP_HIP_LIB=... python tools/probes/wgrad_ab.py"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from imgcomp_cvpr_amd import _lib as L
dev = torch.device('cuda:0')
N, H, W = 32, 32, 32
g = torch.Generator().manual_seed(2)
x = torch.randn((N, 128, H, W), generator=g).to(dev); dy = torch.randn((N, 128, H, W), generator=g).to(dev)
w = torch.randn((3, 3, 128, 128), generator=g).to(dev)
need = L.lib.ic_conv3x3_c128_wgrad_workspace_bytes(N, H, W)
ws = torch.empty(need, dtype=torch.uint8, device=dev); dw = torch.empty_like(w)
def go(): L.check(L.lib.ic_conv3x3_c128_wgrad_f32(L.ptr(x), L.ptr(dy), L.ptr(dw), N, H, W, L.ptr(w), 0.25, L.ptr(ws), need, L.current_stream(dev)))
for _ in range(5): go()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): go()
e1.record(); torch.cuda.synchronize()
ref = torch.einsum('nchw,ndhw->cd', x.double(), dy.double())          # centre tap of dW: sum over positions of x[ci] * dy[co]
err = float((dw[1, 1].double() - (ref + 0.25 * w[1, 1].double())).abs().max() / ref.abs().max())
print('%s: %.1f us per call, crc32(dW) %08x, centre tap rel err %.2e' % (os.path.basename(L.LIB_PATH), e0.elapsed_time(e1) / 50 * 1e3, zlib.crc32(dw.cpu().numpy().tobytes()), err))
