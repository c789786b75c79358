"""does the F(4x4) kernel return the same values launch after launch at full load? (the check that found the packed-fp32 failure; N H W from argv as in wino4_check.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'wino4_check.py')).read().split("torch.set_num_threads(16)")[0])
ref = ref64(1, ())
ys = [run4(1, ()) for _ in range(3)]
torch.cuda.synchronize()
print('deterministic:', bool(torch.equal(ys[0], ys[1])), bool(torch.equal(ys[1], ys[2])))
e = (ys[0].double().cpu() - ref).abs()
print('max err', float(e.max()))
bad = e > 1e-3
print('bad fraction', float(bad.float().mean()))
print('bad per channel (first 32):', bad.float().mean(dim=(0, 2, 3))[:32].numpy().round(2))
print('bad per row (first 16):', bad.float().mean(dim=(0, 1, 3))[:16].numpy().round(2))
print('bad per col (first 80):', bad.float().mean(dim=(0, 1, 2))[:80].numpy().round(2))
print('bad per image:', bad.float().mean(dim=(1, 2, 3)).numpy().round(3))
idx = bad.nonzero()
print('bad elements', idx.shape[0])
print(idx[:40].numpy().tolist())
for k in range(1, 3):
    e2 = (ys[k].double().cpu() - ref).abs() > 1e-3
    print('run', k, 'bad', int(e2.sum()), e2.nonzero()[:8].numpy().tolist())
for k in range(3):
    e2 = (ys[k].double().cpu() - ref)
    bi = (e2.abs() > 1e-3).nonzero()
    for t in bi[:12]:
        n_, c_, y_, x_ = t.tolist()
        print('run', k, t.tolist(), 'got', float(ys[k][n_, c_, y_, x_]), 'ref', float(ref[n_, c_, y_, x_]), 'diff', float(e2[n_, c_, y_, x_]),
              'neighbours diff', [round(float(e2[n_, c_, y_, x_ + d]), 6) for d in (1, 2, 3)], 'row+1', round(float(e2[n_, c_, y_ + 1, x_]), 6))
import collections
e2 = ((ys[2].double().cpu() - ref).abs() > 1e-3).nonzero()
print('run 2 bad total', e2.shape[0])
print('by image', collections.Counter(e2[:, 0].tolist()))
print('by tile row (y//4), y%4', collections.Counter(((e2[:, 2] // 4).tolist())), collections.Counter((e2[:, 2] % 4).tolist()))
print('by channel tile, c%4', collections.Counter((e2[:, 1] // 16).tolist()), collections.Counter((e2[:, 1] % 4).tolist()))
print('by segment (x//64), tile in seg (x%64//4), x%4', collections.Counter((e2[:, 3] // 64).tolist()), collections.Counter((e2[:, 3] % 64 // 4).tolist()), collections.Counter((e2[:, 3] % 4).tolist()))
