#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${TAG:-r4g}
timeout 1500 python -m pytest tests/test_gpu_baseline_sizes.py tests/test_gpu_ragged.py tests/test_gpu_config3.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/${T}_pytest.log 2>&1
echo "rc pytest $?"; grep -E "passed|failed|error|Error|ragged\]" gpurun_out/${T}_pytest.log | tail -10
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "rc bench $?"
python - <<PY
import json
for l in open('gpurun_out/${T}_bench.json'):
    if l.startswith('{'):
        d = json.loads(l)
        print('config1', d['value'], d['config']['us_per_step'], d['roofline']['frac'])
        for k, e in d.get('extra_configs', {}).items():
            print('config', k, e.get('value'), e.get('ms_per_step'), (e.get('roofline') or {}).get('frac'), (e.get('config') or {}).get('kernel'), e.get('error'))
        print('cpu_baseline', d.get('cpu_baseline'))
        print('cpu_port', d.get('cpu_port', {}).get('value'))
PY
