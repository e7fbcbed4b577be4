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
# DUMPCMP="product build_variants/libx.so ...": the labels of configs[2] / [4] at 41 frames (fixed seeds) must be IDENTICAL across builds whose arithmetic is the same
if [ -n "$DUMPCMP" ]; then
  for c in 2 4; do
    for lib in $DUMPCMP; do
      n=$(basename $lib .so)
      if [ "$lib" = product ]; then cmd="python bench.py"; else cmd="python tools/ab_bench.py $lib"; fi
      timeout 200 $cmd --config $c --frames 41 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs --dump /tmp/dump_${c}_${n}.npz > /dev/null 2>&1
    done
    python - <<PY | tee -a gpurun_out/${T}_dumpcmp.txt
import numpy as np
libs = "$DUMPCMP".split()
names = [l.split('/')[-1].replace('.so', '') for l in libs]
ref = np.load('/tmp/dump_${c}_%s.npz' % names[0])
for n in names[1:]:
    try:
        d = np.load('/tmp/dump_${c}_%s.npz' % n)
        same = np.array_equal(d['labels'], ref['labels']) and np.array_equal(d['samples'], ref['samples'])
        print('config ${c}: %s vs %s: %s (%d label mismatches)' % (n, names[0], 'IDENTICAL' if same else 'DIFFERENT', int((d['labels'] != ref['labels']).sum())))
    except Exception as e:
        print('config ${c}: %s: no dump (%r)' % (n, e))
PY
  done
fi
