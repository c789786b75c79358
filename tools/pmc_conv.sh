#!/bin/bash
# PMC passes over the conv3x3 micro-benchmark (separate passes: SQ has 8 slots, TCC 4).
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_conv
mkdir -p $OUT
ARGS="--variants ${1:-2} --pads 0 --reps 3"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" \
         "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/p$i -o p$i --output-format csv -- python $R/tools/bench_conv3x3.py $ARGS > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $C  ($f)"
  python - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
if not f:
    print('no csv'); sys.exit()
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    k = r['Kernel_Name']
    if 'conv3x3' not in k: continue
    key = (k[:60], r['Counter_Name'])
    agg.setdefault(key, []).append(float(r['Counter_Value']))
for (k, c), v in agg.items():
    print('%-62s %-28s n=%d mean=%.4g' % (k, c, len(v), sum(v) / len(v)))
PY
done
