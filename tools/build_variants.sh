#!/bin/bash
# Tuning builds of libimgcomp_hip.so that differ in ONE translation unit (A/B runs on a GPU box through IMGCOMP_HIP_LIB):
#   tools/build_variants.sh <file.hip> <name>=<flags> [<name>=<flags> ...]
# e.g. tools/build_variants.sh conv3x3_wino_tn.hip abl1="-DTN_ABL=1" prof="-DWN_PROF"
# -> imgcomp_cvpr_amd/csrc/variants/lib_<name>.so (ignored by git, shipped by gpurun)
set -e
cd "$(dirname "$0")/../imgcomp_cvpr_amd/csrc"
make -j8 >/dev/null
src=$1; shift
mkdir -p variants
base=${src%.hip}
others=$(ls *.o | grep -v "^${base}.o$")
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -c $src -o variants/${base}_${name}.o &
done
wait
for spec in "$@"; do
  name=${spec%%=*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others variants/${base}_${name}.o -o variants/lib_${name}.so
  rm -f variants/${base}_${name}.o
  echo built variants/lib_${name}.so
done
