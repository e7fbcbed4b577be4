#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${TAG:-r4x}
REPS="1 2" BENCH_ARGS="--kernel batch_cs" bash tools/ab_configs.sh "${CFGS:-2}" product $(ls build_variants/libcs_*.so) 2>&1 | tee gpurun_out/${T}_ab.txt
