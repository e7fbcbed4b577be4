#!/bin/bash
# Short GPU session for iterating on loop_batch.hip (~1.5 min): phase cycles at 8 and 4 rows per team, bench lines of configs 2
# and 4, and only the parity tests that run the batch kernel.  TAG names the output files (gpurun_out/qb_<TAG>_*).
#   /usr/local/graft/bin/gpurun --timeout 300 -- 'TAG=a bash tools/gpu_quick_batch.sh'
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${TAG:-x}
rm -f gpurun_out/qb_${T}_*
for b in 64 32; do
  timeout 120 python bench.py --config 2 --batch $b --frames 41 --steps 1 --warmup 1 --no-cpu-baseline --phase-profile > gpurun_out/qb_${T}_prof_b$b.json 2> gpurun_out/qb_${T}_prof_b$b.err
done
for c in 2 4; do
  timeout 200 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/qb_${T}_bench_c$c.json 2> gpurun_out/qb_${T}_bench_c$c.err
done
timeout 400 python -m pytest tests/test_gpu_baseline_sizes.py tests/test_gpu_parity.py -x -q -m gpu -k "batch or config2 or config4 or many" > gpurun_out/qb_${T}_pytest.log 2>&1
echo "rc pytest $?"; tail -2 gpurun_out/qb_${T}_pytest.log
python - <<PY
import json
for b in (64, 32):
    try:
        for l in open('gpurun_out/qb_${T}_prof_b%d.json' % b):
            if l.startswith('{'):
                d = json.loads(l)['phase_cycles_per_step']['wave0']; print('phase cycles B=%d wave0:' % b, d, 'total', sum(d))
    except Exception as e:
        print('B', b, 'no profile:', e)
PY
python - <<PY
import json
for c in (2, 4):
    try:
        for l in open('gpurun_out/qb_${T}_bench_c%d.json' % c):
            if l.startswith('{'):
                d = json.loads(l); print('config', c, d['value'], 'ksamples/s', d['config']['us_per_step'], 'us/step', d['roofline']['frac'])
    except Exception as e:
        print('config', c, 'no line:', e)
PY
