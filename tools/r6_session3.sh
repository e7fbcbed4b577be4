#!/bin/bash
# round 6, GPU session 3: the conflict-free 4-row landing (CS_PUT2 = 2) -- parity of everything that touches the 4-row kernels, an interleaved
# A/B against CS_PUT2 = 1 (round 5) and 0 (ds_write_b64), and the LDS bank-conflict counters of the shipped build
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=$PWD
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py tests/test_gpu_fold_latency.py tests/test_gpu_ragged.py -m gpu -q -x -k "mol or MOL or config4 or fold or ragged or batch" --durations=8 > gpurun_out/r6s3_pytest.log 2>&1; echo "rc pytest $?"
grep -E "passed|failed|rror" gpurun_out/r6s3_pytest.log | tail -5
REPS="1 2 3 4 5 6" bash tools/ab_configs.sh "4" product build_variants/libput1.so build_variants/libput0.so > gpurun_out/r6s3_ab_put.txt 2>&1
cat gpurun_out/r6s3_ab_put.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES -d $R/gpurun_out/r6s3_sq_c4 -o sq -- python $R/bench.py --config 4 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs > $R/gpurun_out/r6s3_sq_c4_stdout.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/r6s3_sq_c4/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'loop_batch_cs' in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value'])
    print(f, dict(acc))
    if acc.get('SQ_LDS_IDX_ACTIVE'):
        print('LDS bank-conflict cycles / LDS-active cycles = %.4f' % (acc['SQ_LDS_BANK_CONFLICT'] / acc['SQ_LDS_IDX_ACTIVE']))
PY
