#!/bin/bash
# round 6, GPU session 2: new parity tests, the fold legs in bench context, A/B of the CS_SPREAD meeting point
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests/test_gpu_fold_latency.py -m gpu -q -x -s --durations=10 > gpurun_out/r6s2_pytest.log 2>&1; echo "rc pytest $?"
grep -E "^\[parity|passed|failed|rror" gpurun_out/r6s2_pytest.log | tail -30
timeout 600 python tools/fold_in_bench.py > gpurun_out/r6s2_fold_in_bench.txt 2>&1; echo "rc fold_in_bench $?"
REPS="1 2 3" bash tools/ab_configs.sh "2" product build_variants/libnomeet.so > gpurun_out/r6s2_ab_meet.txt 2>&1
cat gpurun_out/r6s2_ab_meet.txt
