#!/bin/bash
# A/B: bench.py with the 16x16-job Winograd kernel off / on (IMGCOMP_WINO_T16)
for v in 0 1; do
  IMGCOMP_WINO_T16=$v python bench.py --steps 20 --warmup 5 | tail -1 > gpurun_out/b_t16_$v.json
done
python - <<'PY'
import json
for v in (0, 1):
    d = json.load(open('gpurun_out/b_t16_%d.json' % v))
    print('t16', v, d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['ms_encode'], d['ms_decode'], d['context_model_stream_cus'], d['pipelined_3_streams']['value'])
PY
