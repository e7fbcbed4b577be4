#!/bin/bash
# round-2 GPU session H: h2' = x3 - x2 written in place of x2 (LDS counter sync inside the workgroup) -> W_hh2 reads one vector
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/r2h_*
for cfg in "2 64" "2 32"; do set -- $cfg
  WRNN_TEAM_PROF=1 timeout 120 python bench.py --config $1 --batch $2 --frames 41 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2h_prof_c$1_b$2.json 2> gpurun_out/r2h_prof_c$1_b$2.err
done
timeout 200 python bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2h_bench_c2.json 2> gpurun_out/r2h_bench_c2.err
timeout 200 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2h_bench_c4.json 2> gpurun_out/r2h_bench_c4.err
timeout 600 python -m pytest tests -x -q -m gpu --durations=5 > gpurun_out/r2h_pytest_gpu.log 2>&1
echo "rc pytest_gpu $?" >> gpurun_out/r2h_summary.log
cat gpurun_out/r2h_summary.log; tail -3 gpurun_out/r2h_pytest_gpu.log
