#!/bin/bash
# round-2 GPU session F: what the driver runs at round end -- the whole -m gpu suite (timed) and smoke()
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/r2f_*
true
echo "rc smoke $?" >> gpurun_out/r2f_summary.log
timeout 1000 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/r2f_pytest_gpu.log 2>&1
echo "rc pytest_gpu $?" >> gpurun_out/r2f_summary.log
cat gpurun_out/r2f_summary.log; tail -3 gpurun_out/r2f_smoke.log; tail -25 gpurun_out/r2f_pytest_gpu.log | head -20
