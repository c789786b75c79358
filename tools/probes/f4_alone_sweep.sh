# one image at a time: the automatic 3x3 plan against F(4x4) forced, over image sizes (bench.py --in_flight 1)
mkdir -p gpurun_out
for hw in "384 512" "512 512" "448 768" "512 768" "640 768" "768 768" "768 1024"; do
  set -- $hw
  for pf in 0 0x0b; do
    r=$(python bench.py --no_extras --steps 30 --warmup 5 --in_flight 1 --plan_flags $pf --height $1 --width $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "$1x$2 plan_flags=$pf $r" | tee -a gpurun_out/f4_alone_sweep.txt
  done
done
