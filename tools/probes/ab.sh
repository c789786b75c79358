# A/B of library variants on the bench step: tools/probes/ab.sh <out-name> <variant|shipped> ...   (bench.py --no_extras, 3 runs each)
out=gpurun_out/$1; shift
mkdir -p $(dirname $out)
for v in "$@"; do
  lib=""; [ "$v" != shipped ] && lib=$PWD/imgcomp_cvpr_amd/csrc/variants/lib_$v.so
  for rep in 1 2 3; do
    r=$(IMGCOMP_HIP_LIB=$lib python bench.py --no_extras --steps 80 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")
    echo "$v in_flight4 $r" >> $out
  done
  r=$(IMGCOMP_HIP_LIB=$lib python bench.py --no_extras --steps 40 --warmup 5 --in_flight 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")
  echo "$v one_at_a_time $r" >> $out
  r=$(IMGCOMP_HIP_LIB=$lib python bench.py --no_extras --steps 40 --warmup 5 --in_flight 1 --plan_flags 0x0b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")
  echo "$v one_at_a_time_f4 $r" >> $out
done
cat $out
