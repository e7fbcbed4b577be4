#!/bin/bash
# One GPU-box session: the whole `-m gpu` suite, the default bench line, the device-copy peak.  Outputs under gpurun_out/<TAG>_*.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'TAG=r3b bash tools/gpu_session.sh'
# WHAT selects the legs (default "tests bench copy"); PYTEST_ARGS narrows the test leg.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${TAG:-x}
WHAT=${WHAT:-tests bench copy}
for leg in $WHAT; do
  case $leg in
    tests)
      timeout ${TEST_TIMEOUT:-1200} python -m pytest tests -m gpu -q --durations=15 ${PYTEST_ARGS:--x} -s > gpurun_out/${T}_pytest.log 2>&1
      echo "rc pytest $?"; grep -E "^\[parity|^\[ragged|passed|failed|error" gpurun_out/${T}_pytest.log | tail -40 ;;
    bench)
      timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
      echo "rc bench $?"; python - <<PY
import json
for l in open('gpurun_out/${T}_bench.json'):
    if l.startswith('{'):
        d = json.loads(l)
        print('config1', d['value'], d['unit'], d['config']['us_per_step'], 'us/step', {k: v for k, v in d['roofline'].items() if k not in ('note', 'latency_model')})
        for k, e in d.get('extra_configs', {}).items():
            print('config', k, e.get('value'), e.get('ms_per_step'), (e.get('roofline') or {}).get('frac'), e.get('error'))
        print('cpu_baseline', d.get('cpu_baseline', {}).get('value'))
PY
      ;;
    copy)
      ( cd bench_micro && make -s devcopy 2>/dev/null; timeout 120 ./devcopy ) > gpurun_out/${T}_devcopy.txt 2>&1; cat gpurun_out/${T}_devcopy.txt ;;
  esac
done
