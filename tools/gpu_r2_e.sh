#!/bin/bash
# round-2 GPU session E: BASELINE-size parity file (oracle capped at 16 threads) + rocprofv3 profiles
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/r2e_*
timeout 450 python -m pytest tests/test_gpu_baseline_sizes.py -v -s --tb=long --durations=0 > gpurun_out/r2e_baseline.log 2>&1
echo "rc baseline $?" >> gpurun_out/r2e_summary.log
timeout 500 bash tools/profile_r2.sh > gpurun_out/r2e_profile.log 2>&1
echo "rc profile $?" >> gpurun_out/r2e_summary.log
cat gpurun_out/r2e_summary.log
