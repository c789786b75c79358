for v in shipped ablB ablAB ablT ablABT; do
  lib=""; [ "$v" != shipped ] && lib=$PWD/imgcomp_cvpr_amd/csrc/variants/lib_$v.so
  a=$(IMGCOMP_HIP_LIB=$lib python tools/wino4_check.py 8 128 192 2>&1 | tail -1 | sed 's/F(2x2).*//')
  b=$(IMGCOMP_HIP_LIB=$lib python bench.py --no_extras --steps 80 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")
  c=$(IMGCOMP_HIP_LIB=$lib python bench.py --no_extras --steps 40 --warmup 5 --in_flight 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")
  echo "$v | $a | in flight $b | one at a time $c" | tee -a gpurun_out/abl.txt
done
