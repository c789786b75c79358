#!/bin/bash
# All rocprofv3 passes of a round, on the GPU box: gpurun -- tools/profile_round.sh
# -> gpurun_out/prof_*/ ; tools/profile_digest.py turns them into the files committed under profiles/.
set +e
t=tools/profile.sh
timeout 320 $t trace bench python bench.py --steps 30 --warmup 5 --no_extras
for form in seg3 wholek; do
  timeout 320 $t pmc l_${form}_fetch "FETCH_SIZE" python tools/run_layer.py --form $form --calib
  timeout 320 $t pmc l_${form}_write "WRITE_SIZE" python tools/run_layer.py --form $form --calib
  timeout 320 $t pmc l_${form}_l2 "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" python tools/run_layer.py --form $form --calib
  timeout 320 $t pmc l_${form}_ea "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" python tools/run_layer.py --form $form --calib
  timeout 320 $t pmc l_${form}_sq "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" python tools/run_layer.py --form $form
done
timeout 320 $t trace pc python tools/run_pc.py 40
timeout 320 $t trace train python bench.py --mode train --steps 6 --warmup 2
ls gpurun_out | head -40
