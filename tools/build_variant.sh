#!/bin/bash
# Build an experimental variant of the library next to the product one:
#   tools/build_variant.sh NAME file1,file2 -DMACRO=1 ...      (files = the .hip sources, without extension, the macros apply to)
# -> build_variants/libNAME.so (git-ignored; travels to the GPU box with gpurun): the named sources recompiled with the extra flags,
# every other object taken from the product build (csrc/.obj).  Used with tools/ab_bench.py / tools/ab_configs.sh.
set -e
name=${1:?usage: build_variant.sh NAME file1,file2 [hipcc flags]}; files=${2:?files}; shift 2
cd "$(dirname "$0")/../tacotronv2_wavernn_chinese_amd/csrc"
make -s -j8
mkdir -p ../../build_variants .obj_var/$name
objs=""
for o in .obj/*.o; do
  b=$(basename $o .o)
  if [[ ",$files," == *",$b,"* ]]; then
    extra=""; [ "$b" = loop_batch_cs ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed $extra "$@" -c $b.hip -o .obj_var/$name/$b.o
    objs="$objs .obj_var/$name/$b.o"
  else
    objs="$objs $o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -no-hip-rt $objs -o ../../build_variants/lib$name.so
ls -la ../../build_variants/lib$name.so
