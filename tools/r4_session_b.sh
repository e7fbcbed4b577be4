#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${TAG:-r4b}
REPS="1" BENCH_ARGS="--kernel batch_cs" bash tools/ab_configs.sh "4" product $(ls build_variants/libcs_*.so) 2>&1 | tee gpurun_out/${T}_ab.txt
REPS="1" BENCH_ARGS="--kernel batch_cs --batch 32" bash tools/ab_configs.sh "2" product $(ls build_variants/libcs_*.so) 2>&1 | tee -a gpurun_out/${T}_ab.txt
