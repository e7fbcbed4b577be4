#!/bin/bash
# round-2 GPU session C: diagnostics -- fine-grained phase cycles of the batch kernel, failing BASELINE-size tests with tracebacks,
# team2 after the LDS plane padding
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/r2c_*
for cfg in "2 64" "2 32" "4 32"; do set -- $cfg
  WRNN_TEAM_PROF=1 timeout 120 python bench.py --config $1 --batch $2 --frames 41 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2c_prof_c$1_b$2.json 2> gpurun_out/r2c_prof_c$1_b$2.err
  timeout 120 python bench.py --config $1 --batch $2 --frames 41 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2c_bench_c$1_b$2.json 2> gpurun_out/r2c_bench_c$1_b$2.err
done
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2c_bench_c1.json 2> gpurun_out/r2c_bench_c1.err
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "team2 and (free_running or teacher_forced or mol)" --durations=3 > gpurun_out/r2c_team2.log 2>&1
echo "rc team2 $?" >> gpurun_out/r2c_summary.log
timeout 700 python -m pytest tests/test_gpu_baseline_sizes.py -v -s --tb=short --timeout=300 --durations=0 -k "philox or config2 or config4 or 9bit" > gpurun_out/r2c_baseline.log 2>&1
echo "rc baseline $?" >> gpurun_out/r2c_summary.log
cat gpurun_out/r2c_summary.log
