mkdir -p gpurun_out/r5b
python -m pytest tests -m gpu -x -q -k "cfg1_256 or cfg5_tiles or pin_reference or pinned_kodak or hip_ms_ssim or ms_ssim_loss" 2>&1 | tail -8 > gpurun_out/r5b/new_tests.log
IMGCOMP_HIP_LIB=$PWD/imgcomp_cvpr_amd/csrc/variants/lib_ls.so python tools/w4_inflight_stamps.py --out gpurun_out/r5b/inflight_stamps.json > gpurun_out/r5b/stamps.log 2>&1
python bench.py > gpurun_out/r5b/bench.json 2> gpurun_out/r5b/bench.err
cat gpurun_out/r5b/new_tests.log; tail -3 gpurun_out/r5b/stamps.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/r5b/bench.json'))
print(d['value'], d['ms_per_step'], d['config'].get('one_image_at_a_time_mpix_s'), d['roofline']['frac'], d['roofline'].get('from_stamps'))
print(d['train']['ms_per_step'], [ (s['batch'],s['value'],s['one_image_at_a_time']['value']) for s in d['shapes']], d['cfg5_4k'].get('value'))
PY
