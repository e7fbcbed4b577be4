#!/bin/bash
# A/B of the batch kernel (configs 2 and 4) in ONE session: current library vs variants under build_variants/.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${TAG:-abb}
run() { n=$1; shift; "$@" > gpurun_out/${T}_$n.json 2> gpurun_out/${T}_$n.err
  python - <<PY
import json
try:
    for l in open('gpurun_out/${T}_$n.json'):
        if l.startswith('{'):
            d = json.loads(l); print('%-14s %9.1f ksamples/s  %.4f us/step  frac %.4f' % ('$n', d['value'], d['config']['us_per_step'], d['roofline']['frac']))
except Exception as e:
    print('$n', 'failed', e)
PY
}
for rep in 1 2; do
  for c in 2 4; do
    run cur_c${c}_$rep python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline
    for v in ${VARIANTS}; do
      run ${v}_c${c}_$rep python tools/ab_bench.py build_variants/lib$v.so --config $c --steps 2 --warmup 1 --no-cpu-baseline
    done
  done
done
