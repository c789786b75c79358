"""Where does a build of the F(4x4) kernel return wrong values under full load, and in which lanes / channels / positions?
   IMGCOMP_HIP_LIB=<variant .so> python tools/w4_rootcause.py [launches] [N H W]
Reference = element-wise median of three launches of the SAME build (the failure is rare and never hits a place twice), so no
second library is needed.  Two regimes: one launch of N maps (two work-groups per CU from one grid), and N = 1 launches on four
streams (two work-groups per CU from different grids: what val.py / bench.py run)."""
import os, sys, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imgcomp_cvpr_amd import _lib as L
lib = L.lib
dev = torch.device('cuda:0')
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 60
N, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (8, 128, 192)
g = torch.Generator().manual_seed(1)
x = (torch.relu(torch.randn((N, 128, H, W), generator=g)) * 1.5).to(dev)
w = (torch.randn((3, 3, 128, 128), generator=g) * 0.03).to(dev)
sc = (torch.rand(128, generator=g) * 0.6 + 0.5).to(dev)
sh = (torch.randn(128, generator=g) * 0.1).to(dev)
r1 = torch.randn((N, 128, H, W), generator=g).to(dev)
wp4 = torch.empty(lib.ic_wino4_3x3_c128_packed_floats(), device=dev)
L.check(lib.ic_pack_wino4_3x3_c128_f32(L.ptr(w), L.ptr(wp4), 0, L.current_stream(dev)))


def run(xi, ri, y, n):
    L.check(lib.ic_wino4_3x3_c128_bn_act_f32(L.ptr(xi), L.ptr(wp4), L.ptr(sc), L.ptr(sh), L.ptr(ri), None, L.ptr(y), n, H, W, 1, 0, L.current_stream(dev)))


def describe(bad_idx, tag):
    c = collections.Counter
    n_, ch, yy, xx = (bad_idx[:, k].tolist() for k in range(4))
    rep = {
        'tile_in_segment': dict(sorted(c((v % 64) // 4 for v in xx).items())),
        'wave(c//16%4)': dict(sorted(c((v // 16) % 4 for v in ch).items())),
        'half(c//64)': dict(sorted(c(v // 64 for v in ch).items())),
        'kq((c%16)//4)': dict(sorted(c((v % 16) // 4 for v in ch).items())),
        'reg(c%4)': dict(sorted(c(v % 4 for v in ch).items())),
        'y%4': dict(sorted(c(v % 4 for v in yy).items())), 'x%4': dict(sorted(c(v % 4 for v in xx).items())),
    }
    ev = collections.defaultdict(set)          # (image, tile row, tile col) -> channels hit
    for a, b_, c_, d in zip(n_, ch, yy, xx): ev[(a, c_ // 4, d // 4)].add(b_)
    rep['events(tiles hit)'] = len(ev)
    rep['channels per event'] = dict(sorted(c(len(v) for v in ev.values()).items()))
    rep['first events'] = [[list(k), sorted(v)[:20]] for k, v in list(ev.items())[:6]]
    print(tag, json.dumps(rep), flush=True)


# regime 1: one grid
ys = [torch.empty_like(x) for _ in range(3)]
for y in ys: run(x, r1, y, N)
torch.cuda.synchronize()
ref = torch.stack(ys).median(dim=0).values
bad_launches, bad_total, all_bad = 0, 0, []
y = torch.empty_like(x)
for it in range(launches):
    run(x, r1, y, N)
    d = (y != ref)
    nb = int(d.sum())
    if nb:
        bad_launches += 1; bad_total += nb
        if len(all_bad) < 40: all_bad.append(d.nonzero().cpu())
print('ONE GRID  N={} {}x{}: bad launches {} / {}   bad elements {}'.format(N, H, W, bad_launches, launches, bad_total), flush=True)
if all_bad: describe(torch.cat(all_bad), 'ONE GRID ')
# regime 2: four streams, one map each
streams = [torch.cuda.Stream() for _ in range(4)]
x1 = [x[i:i + 1].contiguous() for i in range(4)]; rr = [r1[i:i + 1].contiguous() for i in range(4)]
yo = [torch.empty_like(x1[0]) for _ in range(4)]
bad_launches, bad_total, all_bad = 0, 0, []
for it in range(launches):
    torch.cuda.synchronize()
    for rep in range(4):
        for i, s in enumerate(streams):
            with torch.cuda.stream(s): run(x1[i], rr[i], yo[i], 1)
    torch.cuda.synchronize()
    for i in range(4):
        d = (yo[i] != ref[i:i + 1])
        nb = int(d.sum())
        if nb:
            bad_launches += 1; bad_total += nb
            if len(all_bad) < 40: all_bad.append(d.nonzero().cpu())
print('4 STREAMS N=1 {}x{}: bad launches {} / {}   bad elements {}'.format(H, W, bad_launches, 4 * launches, bad_total), flush=True)
if all_bad: describe(torch.cat(all_bad), '4 STREAMS')
