#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${TAG:-r4n}
REPS="1 2" BENCH_ARGS="--kernel batch_cs" bash tools/ab_configs.sh "2 4" product $(ls build_variants/libcs_*.so) 2>&1 | tee gpurun_out/${T}_ab.txt
for lib in $(ls build_variants/libcs_*.so); do timeout 200 python tools/phase_profile.py 2 64 41 batch_cs $lib | grep -E "wave0|wave4|us/step"; done 2>&1 | tee gpurun_out/${T}_phases.txt
