#!/bin/bash
# round-2 GPU session B: restructured batch kernel (no spills), forward/loss, any-dims, BASELINE-size parity with timings
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/r2b_*
timeout 500 python -m pytest tests/test_gpu_parity.py -q -x -k "batch" --durations=5 > gpurun_out/r2b_batch_short.log 2>&1
echo "rc batch_short $?" >> gpurun_out/r2b_summary.log
timeout 300 python -m pytest tests/test_forward_loss.py tests/test_gpu_any_dims.py -q --durations=5 > gpurun_out/r2b_forward_anydims.log 2>&1
echo "rc forward_anydims $?" >> gpurun_out/r2b_summary.log
WRNN_TEAM_PROF=1 timeout 200 python bench.py --config 2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2b_bench_c2_prof.json 2> gpurun_out/r2b_bench_c2_prof.err
echo "rc bench2prof $?" >> gpurun_out/r2b_summary.log
timeout 200 python bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2b_bench_c2.json 2> gpurun_out/r2b_bench_c2.err
echo "rc bench2 $?" >> gpurun_out/r2b_summary.log
WRNN_TEAM_PROF=1 timeout 200 python bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2b_bench_c4_prof.json 2> gpurun_out/r2b_bench_c4_prof.err
timeout 200 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2b_bench_c4.json 2> gpurun_out/r2b_bench_c4.err
echo "rc bench4 $?" >> gpurun_out/r2b_summary.log
timeout 900 python -m pytest tests/test_gpu_baseline_sizes.py -q -s --durations=0 > gpurun_out/r2b_baseline_sizes.log 2>&1
echo "rc baseline_sizes $?" >> gpurun_out/r2b_summary.log
cat gpurun_out/r2b_summary.log; tail -3 gpurun_out/r2b_batch_short.log; tail -8 gpurun_out/r2b_forward_anydims.log
