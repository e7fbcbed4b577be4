#!/bin/bash
# round-4 session: the critical/shadow batch kernel -- parity subset, A/B against the single-wave batch kernel, phase cycles
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${TAG:-r4a}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_forward_loss.py tests/test_gpu_ragged.py tests/test_gpu_baseline_sizes.py -m gpu -q -x -k "batch_cs" -s > gpurun_out/${T}_pytest.log 2>&1
echo "rc pytest $?"; grep -E "^\[parity|passed|failed|error|Error" gpurun_out/${T}_pytest.log | tail -30
REPS="1 2" bash tools/ab_kernels.sh "4 2" batch batch_cs 2>&1 | tee gpurun_out/${T}_ab.txt
for spec in "4 32" "2 32"; do set -- $spec; timeout 200 python tools/phase_profile.py $1 $2 41 batch_cs; timeout 200 python tools/phase_profile.py $1 $2 41 batch; done 2>&1 | tee gpurun_out/${T}_phases.txt
