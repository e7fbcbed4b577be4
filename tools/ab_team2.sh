#!/bin/bash
# A/B of config 1 (latency kernel) in ONE session on one box: the round-2 tree (build_variants/r2tree, ABI 3), the current
# library, and the experimental variants under build_variants/.  Prints us/step of each, twice (order effects).
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${TAG:-ab}
run() { # name, command...
  n=$1; shift
  "$@" > gpurun_out/${T}_$n.json 2> gpurun_out/${T}_$n.err
  python - <<PY
import json
try:
    for l in open('gpurun_out/${T}_$n.json'):
        if l.startswith('{'):
            d = json.loads(l); print('%-10s %9.2f ksamples/s  %.4f us/step' % ('$n', d['value'], d['config']['us_per_step']))
except Exception as e:
    print('$n', 'failed', e)
PY
}
for rep in 1 2; do
  [ -d build_variants/r2tree ] && run r2_$rep python build_variants/r2tree/bench.py --config 1 --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline
  run cur_$rep python bench.py --config 1 --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-extra-configs
  for v in ${VARIANTS}; do
    run ${v}_$rep python tools/ab_bench.py build_variants/lib$v.so --config 1 --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-extra-configs
  done
done
