#!/bin/bash
# A/B of loop kernels of the SHIPPED library on bench configs in one GPU session:  tools/ab_kernels.sh "2 4" batch batch_cs
cd "$(dirname "$0")/.." || exit 1
cfgs=$1; shift
for rep in ${REPS:-1 2}; do
  for c in $cfgs; do
    for k in "$@"; do
      timeout 300 python bench.py --config $c --kernel $k --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-extra-configs ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('config $c rep $rep %-10s %9.1f ksamples/s  loop %.3f ms  %.4f us/step  roofline %.4f' % ('$k', d['value'], d['config']['loop_kernel_ms'], d['config']['us_per_step'], d['roofline']['frac']))
"
    done
  done
done
