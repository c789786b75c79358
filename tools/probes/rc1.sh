mkdir -p gpurun_out/rc
for v in slp slp_pkpost4 slp_pkpre4 slp_war4 slp_mpost2 slp_mpre2 slp_dspre4 slp_dpppre4 slp_spread0; do
  echo "=== $v" >> gpurun_out/rc/rc1.log
  IMGCOMP_HIP_LIB=$PWD/imgcomp_cvpr_amd/csrc/variants/lib_$v.so timeout 300 python tools/w4_rootcause.py 40 >> gpurun_out/rc/rc1.log 2>&1
done
echo "=== shipping" >> gpurun_out/rc/rc1.log
timeout 300 python tools/w4_rootcause.py 40 >> gpurun_out/rc/rc1.log 2>&1
tail -c 6000 gpurun_out/rc/rc1.log
