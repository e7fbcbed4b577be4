#!/bin/bash
# Build an experimental variant of the library next to the product one:  tools/build_variant.sh NAME -DMACRO=1 ...
# -> build_variants/libNAME.so (git-ignored; travels to the GPU box with gpurun).  Used with tools/ab_bench.py.
set -e
name=${1:?usage: build_variant.sh NAME [hipcc flags]}; shift
cd "$(dirname "$0")/../tacotronv2_wavernn_chinese_amd/csrc"
mkdir -p ../../build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -no-hip-rt -Wno-unused-result "$@" \
  api.hip prologue.hip loop_simple.hip loop_team2.hip loop_batch.hip loop_deepmind.hip loop_dm_team.hip epilogue.hip losses.hip train.hip train_team.hip \
  -o ../../build_variants/lib$name.so 2>&1 | grep -E "error" || true
ls -la ../../build_variants/lib$name.so
