#!/bin/bash
# round 6, GPU session 4: the swizzled landing (CS_PUT2 = 2) changes WHO a sentinel lane waits for (both granules of a lane's pair now come from one
# producer wave): the early-look threshold re-tuned for it
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
REPS="1 2 3" bash tools/ab_configs.sh "4" build_variants/libput1.so product build_variants/libel16.so build_variants/libel24.so build_variants/libel32.so build_variants/libel48.so build_variants/libel0.so > gpurun_out/r6s4_ab_el.txt 2>&1
cat gpurun_out/r6s4_ab_el.txt
