#!/bin/bash
# HBM traffic of the 3x3 layer at the Kodak residual-stack shape: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (TCC has 4 counter slots: FETCH_SIZE costs 3, WRITE_SIZE 2), kernel trace only -- MI355X_MICROARCH.md "HBM" / "PMC slots".
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_wino
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU"; do
  tag=$(echo $C | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $C -d $OUT/$tag -o p --output-format csv -- python $R/tools/bench_wino.py --reps 4 > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
if not f:
    print('no csv'); sys.exit()
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'wino3x3' not in k and 'conv3x3_c128' not in k: continue
    agg.setdefault((k.split('(')[0][:50], r['Counter_Name']), []).append(float(r['Counter_Value']))
for (k, c), v in agg.items():
    print('%-52s %-26s launches=%d mean=%.6g' % (k, c, len(v), sum(v) / len(v)))
PY
done
