#!/bin/bash
# Builds of libimgcomp_hip.so whose conv3x3_wino4 device code is the compiler's assembly with wait states patched in
# (tools/w4_isa_patch.py): tools/w4_isa_variants.sh "<extra hipcc flags>" name=rule[,rule] ...   (rule "none" = unpatched)
# -> imgcomp_cvpr_amd/csrc/variants/lib_<name>.so
set -e
LL=/opt/rocm/lib/llvm/bin
here=$(cd "$(dirname "$0")" && pwd)
cd "$here/../imgcomp_cvpr_amd/csrc"
make -j8 >/dev/null
mkdir -p variants
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function"
extra=$1; shift
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc $F $extra --cuda-device-only -S conv3x3_wino4.hip -o $tmp/dev.s 2>/dev/null
others=$(ls *.o | grep -v "^conv3x3_wino4.o$")
for spec in "$@"; do
  name=${spec%%=*}; rule=${spec#*=}
  if [ "$rule" = none ]; then cp $tmp/dev.s $tmp/$name.s; else python3 $here/w4_isa_patch.py $tmp/dev.s $tmp/$name.s $rule; fi
  $LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $tmp/$name.s -o $tmp/$name.dev.o
  $LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared $tmp/$name.dev.o -o $tmp/$name.out
  $LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$tmp/$name.out -output=$tmp/$name.hipfb
  /opt/rocm/bin/hipcc $F $extra --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $tmp/$name.hipfb -c conv3x3_wino4.hip -o $tmp/$name.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $tmp/$name.o -o variants/lib_$name.so
  echo built variants/lib_$name.so
done
rm -rf $tmp
