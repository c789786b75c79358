set +e
python - > gpurun_out/train_ab.log 2>&1 <<'PY'
import sys, os, torch
sys.path.insert(0, os.getcwd())
import bench
from imgcomp_cvpr_amd import training
dev = torch.device('cuda:0')
for rnd in range(3):
    for fuse in (True, False):
        training.TrainGraph.FUSE_BN_STATS = fuse
        el, out, tr = bench.train_steps_timed(torch, None, dev, 0, 1, 32, 128, 128, 12, 4)
        print('FUSE_BN_STATS', fuse, 'ms per step %.3f' % (el / 12 * 1e3), {k: round(float(v), 5) for k, v in out.items()}, flush=True)
        del tr
PY
timeout 600 python -m pytest tests -m gpu -x -q -k "epilogue_batchnorm or val_batches" > gpurun_out/gputest7.log 2>&1
tail -3 gpurun_out/gputest7.log | head -1
python tools/val_batch_profile.py > gpurun_out/val_batch_profile.log 2>&1
tools/profile.sh trace tr7 python bench.py --mode train --steps 6 --warmup 2 > /dev/null 2>&1
cp $(find gpurun_out/prof_tr7 -name "*kernel_stats.csv") gpurun_out/train7_kernel_stats.csv; rm -rf gpurun_out/prof_tr7
