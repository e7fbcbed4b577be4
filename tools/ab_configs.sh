#!/bin/bash
# A/B of library builds on bench configs in ONE GPU session:  [BENCH_ARGS="--kernel batch_cs"] tools/ab_configs.sh "2 4" libA.so libB.so ...   ("product" = the shipped one)
cd "$(dirname "$0")/.." || exit 1
cfgs=$1; shift
for rep in ${REPS:-1 2}; do
  for c in $cfgs; do
    for lib in "$@"; do
      if [ "$lib" = product ]; then cmd="python bench.py"; else cmd="python tools/ab_bench.py $lib"; fi
      timeout 200 $cmd --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('config $c rep $rep %-28s %9.1f ksamples/s  %.4f us/step' % ('$lib'.split('/')[-1], d['value'], d['config']['us_per_step']))
"
    done
  done
done
