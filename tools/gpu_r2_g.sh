#!/bin/bash
# round-2 GPU session G: R = 4 batch kernel with the conditioning loads one step ahead and the x3 / fc1 gathers requested mid shadow work
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/r2g_*
for cfg in "2 32" "4 32"; do set -- $cfg
  WRNN_TEAM_PROF=1 timeout 120 python bench.py --config $1 --batch $2 --frames 41 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2g_prof_c$1_b$2.json 2> gpurun_out/r2g_prof_c$1_b$2.err
done
timeout 200 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2g_bench_c4.json 2> gpurun_out/r2g_bench_c4.err
timeout 200 python bench.py --config 2 --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2g_bench_c2_b32.json 2> gpurun_out/r2g_bench_c2_b32.err
timeout 600 python -m pytest tests -x -q -m gpu --durations=5 > gpurun_out/r2g_pytest_gpu.log 2>&1
echo "rc pytest_gpu $?" >> gpurun_out/r2g_summary.log
cat gpurun_out/r2g_summary.log; tail -3 gpurun_out/r2g_pytest_gpu.log
