"""determinism of the F(2x2) forms at full load (2 work-groups per CU for the NB = 1 segment form)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'wino4_check.py')).read().split("torch.set_num_threads(16)")[0])
ref = ref64(1, ())
for name, fl in (('seg1', L.CONV3_WINO_SEG1), ('seg3', L.CONV3_WINO_SEG3), ('wholek', L.CONV3_WINO_WHOLEK)):
    bad = 0
    same = True
    first = None
    for it in range(12):
        y = run2(1, (), fl)
        torch.cuda.synchronize()
        if first is None:
            first = y.clone()
        same = same and bool(torch.equal(first, y))
        bad += int(((y.double().cpu() - ref).abs() > 1e-3).sum())
    print(name, 'deterministic over 12 launches:', same, ' bad elements:', bad, flush=True)
