#!/bin/bash
# round 6, GPU session 5: CS_EARLY_LOOK re-swept on the SHIPPED granule placement (round 4 found 1-16 within 1 %; the kernel changed since)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
REPS="1 2 3" bash tools/ab_configs.sh "4 2" product build_variants/libel2.so build_variants/libel4.so build_variants/libel12.so build_variants/libel16.so build_variants/libel24.so > gpurun_out/r6s5_ab_el.txt 2>&1
cat gpurun_out/r6s5_ab_el.txt
