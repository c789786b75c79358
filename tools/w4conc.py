"""F(4x4) launches of one work-group per CU next to an unrelated kernel stream: does ANY co-resident wave break it? (it does not -- only a second wave of this kernel with packed fp32 ops did)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'wino4_check.py')).read().split("torch.set_num_threads(16)")[0])
# N = 1 launches (192 work-groups: one per CU) next to an unrelated kernel stream: does ANY co-resident wave break it?
ref = ref64(1, ())
s2 = torch.cuda.Stream()
big = torch.randn(4096, 4096, device=dev)
bad_tot = 0
for it in range(30):
    with torch.cuda.stream(s2):
        for _ in range(3):
            big @ big
    y = run4(1, ())
    torch.cuda.synchronize()
    bad_tot += int(((y.double().cpu() - ref).abs() > 1e-3).sum())
print('N = 1 beside a GEMM stream: bad elements over 30 launches:', bad_tot)
# two of my launches on two streams (each 192 work-groups): pairs share CUs
s3 = torch.cuda.Stream()
bad_tot = 0
for it in range(30):
    with torch.cuda.stream(s2):
        ya = run4(1, ())
    with torch.cuda.stream(s3):
        yb = run4(1, ())
    torch.cuda.synchronize()
    bad_tot += int(((ya.double().cpu() - ref).abs() > 1e-3).sum()) + int(((yb.double().cpu() - ref).abs() > 1e-3).sum())
print('two N = 1 launches on two streams: bad elements over 30 pairs:', bad_tot)
