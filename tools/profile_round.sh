#!/bin/bash
# All rocprofv3 passes of a round, on the GPU box:   gpurun -- tools/profile_round.sh
# -> gpurun_out/prof_*/ ; tools/profile_digest.py rNN turns them into the files committed under profiles/.
# Every pass profiles THE SAME command -- the step bench.py times, in the shipped schedule -- so one trace gives every kernel's
# average duration and each counter pass gives every kernel's counters (3x3 layer, the context model's kernels, the six
# 5x5 / stride-2 layers).  --calib_copy appends a 256 MiB device copy: FETCH_SIZE / WRITE_SIZE are calibrated on it per pass.
set +e
t=tools/profile.sh
STEP="python bench.py --steps 20 --warmup 5 --no_extras --calib_copy"
# the step as shipped (4 images in flight; under the tracer the streams overlap less than untraced: the digest reports the
# concurrency it saw), the SAME kernels one image at a time (the plan hint of the in-flight schedule forced: each launch's
# duration with nothing beside it -- what bench.py reports as `alone`), and the one-at-a-time schedule of rounds 1-2
ALONE="$STEP --in_flight 1 --plan_flags 0x200000"
timeout 320 $t trace bench $STEP
timeout 320 $t trace alone $ALONE
timeout 320 $t trace bench1 $STEP --in_flight 1
STEP=$ALONE          # counter passes serialise the launches anyway: profile the step's kernels one image at a time
timeout 320 $t pmc b_sq "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" $STEP
timeout 320 $t pmc b_fetch "FETCH_SIZE" $STEP
timeout 320 $t pmc b_write "WRITE_SIZE" $STEP
timeout 320 $t pmc b_l2 "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" $STEP
# the context model standalone, on the REAL encoder output of the bench image (what bench.py's roofline_context_model times)
timeout 320 $t trace pc python tools/run_pc.py 40
# the training step (BASELINE configs[2])
timeout 320 $t trace train python bench.py --mode train --steps 6 --warmup 2
# the in-flight schedule WITHOUT a tracer: in-kernel launch stamps (needs the -DW4_LAUNCH_STAMPS build, made in the build container:
#   tools/build_variants.sh conv3x3_wino4.hip ls="-fno-slp-vectorize -DW4_LAUNCH_STAMPS")
if [ -f imgcomp_cvpr_amd/csrc/variants/lib_ls.so ]; then
  IMGCOMP_HIP_LIB=$PWD/imgcomp_cvpr_amd/csrc/variants/lib_ls.so timeout 300 python tools/w4_inflight_stamps.py --out gpurun_out/inflight_stamps.json > gpurun_out/inflight_stamps.log 2>&1
  tail -2 gpurun_out/inflight_stamps.log
fi
ls gpurun_out | head -40
