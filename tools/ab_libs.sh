#!/bin/bash
# run one kbench command against several builds of the library (tools/build_variants.sh), interleaved rounds
# usage: tools/ab_libs.sh "<kbench args>" base abl1 abl2 ...
args=$1; shift
for rnd in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then lib=""; else lib="$(pwd)/imgcomp_cvpr_amd/csrc/variants/lib_$v.so"; fi
    echo "== $v (round $rnd)"
    IMGCOMP_HIP_LIB=$lib python tools/kbench.py $args 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('  ', d.get('shape'), {k: d['us_median'][k] for k in d['us_median']})
"
  done
done
