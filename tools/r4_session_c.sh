#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${TAG:-r4c}
REPS="1" BENCH_ARGS="--kernel batch_cs" bash tools/ab_configs.sh "4" product build_variants/libcs_g2d5.so build_variants/libcs_g2d7.so build_variants/libcs_g2d8.so build_variants/libcs_g2d10.so build_variants/libcs_g2d12.so 2>&1 | tee gpurun_out/${T}_ab.txt
for lib in "" build_variants/libcs_yield.so build_variants/libcs_g2d8.so; do timeout 200 python tools/phase_profile.py 4 32 41 batch_cs $lib | grep -E "wave0|wave4|us/step"; done 2>&1 | tee gpurun_out/${T}_phases.txt
