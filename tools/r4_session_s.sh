#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${TAG:-r4s}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_forward_loss.py tests/test_gpu_ragged.py -m gpu -q -x -k "batch_cs or mol or MOL" > gpurun_out/${T}_pytest.log 2>&1
echo "rc pytest $?"; grep -E "passed|failed|error|Error" gpurun_out/${T}_pytest.log | tail -5
REPS="1 2" BENCH_ARGS="--kernel batch_cs" bash tools/ab_configs.sh "4" product $(ls build_variants/libcs_*.so) 2>&1 | tee gpurun_out/${T}_ab.txt
timeout 200 python tools/phase_profile.py 4 32 41 batch_cs | grep -E "wave0|wave4|us/step" 2>&1 | tee gpurun_out/${T}_phases.txt
