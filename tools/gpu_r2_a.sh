#!/bin/bash
# round-2 GPU session A: MFMA probe, batch-kernel parity on the short goldens, BASELINE-size parity, bench lines
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
./bench_micro/mfma4_probe > gpurun_out/mfma4_probe.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "batch" > gpurun_out/r2a_batch_short.log 2>&1
echo "rc batch_short $?" >> gpurun_out/r2a_summary.log
timeout 900 python -m pytest tests/test_gpu_baseline_sizes.py -q -s > gpurun_out/r2a_baseline_sizes.log 2>&1
echo "rc baseline_sizes $?" >> gpurun_out/r2a_summary.log
timeout 300 python bench.py --steps 3 --warmup 1 > gpurun_out/r2a_bench_c1.json 2> gpurun_out/r2a_bench_c1.err
echo "rc bench1 $?" >> gpurun_out/r2a_summary.log
WRNN_TEAM_PROF=1 timeout 300 python bench.py --config 2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2a_bench_c2.json 2> gpurun_out/r2a_bench_c2.err
echo "rc bench2 $?" >> gpurun_out/r2a_summary.log
timeout 300 python bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2a_bench_c4.json 2> gpurun_out/r2a_bench_c4.err
echo "rc bench4 $?" >> gpurun_out/r2a_summary.log
tail -3 gpurun_out/r2a_batch_short.log; tail -5 gpurun_out/r2a_baseline_sizes.log; cat gpurun_out/r2a_summary.log; cat gpurun_out/mfma4_probe.log
