#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${TAG:-r4i}
REPS="1 2" bash tools/ab_configs.sh "1" product build_variants/libt2_ps1.so build_variants/libt2_ps2.so build_variants/libt2_ps4.so 2>&1 | tee gpurun_out/${T}_ab.txt
REPS="1" BENCH_ARGS="--kernel batch_cs" bash tools/ab_configs.sh "2 4" product build_variants/libcs_g2d6.so build_variants/libcs_g2d9.so build_variants/libcs_g2d12.so build_variants/libcs_g1.so 2>&1 | tee -a gpurun_out/${T}_ab.txt
for c in "2 64" "4 32"; do timeout 200 python tools/phase_profile.py $c 41 batch_cs | grep -E "wave0|wave4|us/step"; done 2>&1 | tee gpurun_out/${T}_phases.txt
