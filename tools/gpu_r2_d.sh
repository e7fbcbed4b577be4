#!/bin/bash
# round-2 GPU session D: batch kernel with direct AGPR operands; BASELINE-size parity (torch Philox replay); bench lines
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/r2d_*
timeout 420 python -m pytest tests/test_gpu_baseline_sizes.py -v -s --tb=long --durations=0 > gpurun_out/r2d_baseline.log 2>&1
echo "rc baseline $?" >> gpurun_out/r2d_summary.log
timeout 240 python -m pytest tests/test_gpu_parity.py -q -k "batch and (free_running or mol or many_rows or edge)" --durations=3 > gpurun_out/r2d_batch_short.log 2>&1
echo "rc batch_short $?" >> gpurun_out/r2d_summary.log
timeout 100 python -m pytest tests/test_forward_loss.py -q > gpurun_out/r2d_forward.log 2>&1
echo "rc forward $?" >> gpurun_out/r2d_summary.log
for cfg in "2 64" "2 32" "4 32"; do set -- $cfg
  WRNN_TEAM_PROF=1 timeout 120 python bench.py --config $1 --batch $2 --frames 41 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2d_prof_c$1_b$2.json 2> gpurun_out/r2d_prof_c$1_b$2.err
done
timeout 200 python bench.py --steps 3 --warmup 1 > gpurun_out/r2d_bench_c1.json 2> gpurun_out/r2d_bench_c1.err
timeout 200 python bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2d_bench_c2.json 2> gpurun_out/r2d_bench_c2.err
timeout 200 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2d_bench_c4.json 2> gpurun_out/r2d_bench_c4.err
timeout 200 python bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2d_bench_c3.json 2> gpurun_out/r2d_bench_c3.err
cat gpurun_out/r2d_summary.log
