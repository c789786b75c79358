mkdir -p gpurun_out/rc
for v in slp0 slp0_guard slp0_gvalu slp0_gmem slp0_guard4 slp0_guardpre slp0_pre noslp0; do
  echo "=== $v" >> gpurun_out/rc/rc2.log
  IMGCOMP_HIP_LIB=$PWD/imgcomp_cvpr_amd/csrc/variants/lib_$v.so timeout 300 python tools/w4_rootcause.py 30 2>&1 | cut -c1-1500 >> gpurun_out/rc/rc2.log
done
cat gpurun_out/rc/rc2.log
