#!/bin/bash
# GPU-box session for the training step: its tests + a rocprofv3 kernel-trace of a few iterations at the reference batch size.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${TAG:-tr}
timeout ${TEST_TIMEOUT:-240} python -m pytest tests/test_train_step.py -m gpu -q -x -s > gpurun_out/${T}_pytest.log 2>&1
echo "rc pytest $?"; grep -E "^\[train|passed|failed|Error|error" gpurun_out/${T}_pytest.log | tail -20
if [ "${PROFILE:-0}" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${T}_prof -o kt -- python -m pytest $GRAFT_REPO_ROOT/tests/test_train_step.py -m gpu -q -x -k optimizer > $GRAFT_REPO_ROOT/gpurun_out/${T}_prof.log 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find gpurun_out/${T}_prof -name "*kernel_stats*" | head -1); [ -n "$f" ] && head -25 $f
fi
