#!/bin/bash
# round-2 GPU session K: opaque per-thread LDS bases (one VGPR per buffer family instead of ~50 hoisted addresses), conditioning prefetch at R = 8
# AGPR shuffles per step), early-issued x3 / fc1 gathers at R = 8 too, merged x2|h1' gather at R = 4
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/r2k_*
for cfg in "2 64" "2 32"; do set -- $cfg
  WRNN_TEAM_PROF=1 timeout 120 python bench.py --config $1 --batch $2 --frames 41 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2k_prof_c$1_b$2.err
done
timeout 200 python bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2k_bench_c2.json 2> gpurun_out/r2k_bench_c2.err
timeout 200 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2k_bench_c4.json 2> gpurun_out/r2k_bench_c4.err
timeout 600 python -m pytest tests -x -q -m gpu --durations=3 > gpurun_out/r2k_pytest_gpu.log 2>&1
echo "rc pytest_gpu $?" >> gpurun_out/r2k_summary.log
cat gpurun_out/r2k_summary.log; tail -3 gpurun_out/r2k_pytest_gpu.log
