#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${TAG:-r4h}
nproc
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py -m gpu -q -x -k "team2 or config1" > gpurun_out/${T}_pytest.log 2>&1
echo "rc pytest $?"; grep -E "passed|failed|error|Error" gpurun_out/${T}_pytest.log | tail -5
REPS="1 2" bash tools/ab_configs.sh "1" product build_variants/libt2_r3.so build_variants/libt2_inline.so build_variants/libt2_single.so build_variants/libt2_stag2.so build_variants/libt2_stag5.so 2>&1 | tee gpurun_out/${T}_ab.txt
REPS="1" BENCH_ARGS="--kernel batch_cs" bash tools/ab_configs.sh "2 4" product build_variants/libcs_lateh1.so build_variants/libcs_lateh1_y.so 2>&1 | tee -a gpurun_out/${T}_ab.txt
timeout 420 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "rc bench $?"; grep "bench.py" gpurun_out/${T}_bench.err | tail -12
python - <<PY
import json
for l in open('gpurun_out/${T}_bench.json'):
    if l.startswith('{'):
        d = json.loads(l)
        print('config1', d['value'], d['config']['us_per_step'], d['roofline']['frac'])
        for k, e in d.get('extra_configs', {}).items():
            print('config', k, e.get('value'), e.get('ms_per_step'), (e.get('roofline') or {}).get('frac'), (e.get('config') or {}).get('kernel'), e.get('error'))
        print('cpu_baseline', {k: v for k, v in d.get('cpu_baseline', {}).items() if k != 'sample'})
        print('cpu_port', d.get('cpu_port', {}).get('value'))
PY
