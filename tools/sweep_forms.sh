#!/bin/bash
# bench.py headline for each 3x3 form x images in flight (A/B inside one gpurun call):  tools/sweep_forms.sh "0 3 8" "1 4 6"
forms=${1:-"0 3"}; flights=${2:-"4 6"}; extra=${3:-}
for f in $forms; do for n in $flights; do
  python bench.py --no_extras --in_flight $n --warmup 20 --plan_flags $f $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plan_flags', sys.argv[1], 'in_flight', d['images_in_flight'], 'Mpix/s', d['value'], 'ms', d['ms_per_step'])" $f
done; done
