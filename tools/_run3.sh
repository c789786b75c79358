set +e
export ST=$PWD/imgcomp_cvpr_amd/csrc/variants/lib_st.so
python tools/w4_wg8.py > gpurun_out/wg8f.log 2>&1
for i in 1 2 3; do python bench.py --no_extras --steps 60 >> gpurun_out/b3_base.log 2>&1; python bench.py --no_extras --steps 60 --plan_flags 0x8000000 >> gpurun_out/b3_wg8f.log 2>&1; done
for sh in "8 128 192" "1 540 960"; do
  IMGCOMP_HIP_LIB=$ST W4_FLAGS=0x8000000 python tools/w4prof.py $sh >> gpurun_out/w4prof_wg8f.log 2>&1
done
C="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
tools/profile.sh pmc sq_wg8f "$C" python tools/run_layer.py --form w4wg8 --n 8 > /dev/null 2>&1
python tools/pmc_summary.py $(find gpurun_out/prof_sq_wg8f -name "*counter_collection.csv") 2>&1 | head -12 > gpurun_out/pmc_wg8f.txt
rm -rf gpurun_out/prof_sq_wg8f
