#!/bin/bash
# One GPU-box session of round 5: parity tests that touch the batch kernels (product library), then A/B of build variants, then phase cycles.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'TAG=s1 LIBS="product build_variants/libr4base.so ..." bash tools/r5_session.sh'
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${TAG:-s}
if [ -n "$PYTEST_SEL" ]; then
  # PYTEST_K: a -k expression with '+' for spaces (the command line travels through two shells), e.g. batch_cs+or+config2
  if [ -n "$PYTEST_K" ]; then KARGS=(-k "${PYTEST_K//+/ }"); else KARGS=(); fi
  timeout ${TEST_TIMEOUT:-900} python -m pytest $PYTEST_SEL -m gpu -q --durations=8 -p no:cacheprovider --timeout ${PER_TEST:-150} "${KARGS[@]}" ${PYTEST_ARGS} > gpurun_out/${T}_pytest.log 2>&1
  echo "rc pytest $?"; grep -E "^\[parity|passed|failed|error|FAILED|ERROR|Timeout" gpurun_out/${T}_pytest.log | tail -40
fi
if [ -n "$LIBS" ]; then
  REPS="${REPS:-1 2}" bash tools/ab_configs.sh "${CFGS:-2 4}" $LIBS 2>&1 | tee gpurun_out/${T}_ab.txt
fi
for spec in $PHASE; do   # e.g. PHASE="2:64 4:32" (product) or "2:64:build_variants/libx.so"
  IFS=: read c b lib <<< "$spec"
  timeout 120 python tools/phase_profile.py $c $b 41 batch_cs $lib 2>&1 | tee -a gpurun_out/${T}_phase.txt
done
