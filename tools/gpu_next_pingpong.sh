#!/bin/bash
# First GPU session for the opt-in ping-pong schedule of the 8-row batch kernel (WRNN_BATCH_PP=1, loop_batch.hip): bit-equality
# with the lock-step kernel, phase cycles of both schedules, bench lines of configs 2 / 3 with and without it.
#   /usr/local/graft/bin/gpurun --timeout 400 -- 'bash tools/gpu_next_pingpong.sh'
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/pp_*
WRNN_TEST_NEXT=1 timeout 300 python -m pytest tests/test_gpu_batch_pingpong.py -x -q > gpurun_out/pp_pytest.log 2>&1
echo "rc pytest_pp $?" >> gpurun_out/pp_summary.log
for pp in 0 1; do
  WRNN_BATCH_PP=$pp WRNN_TEAM_PROF=1 timeout 120 python bench.py --config 2 --frames 41 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pp_prof_c2_pp$pp.err
  WRNN_BATCH_PP=$pp timeout 200 python bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pp_bench_c2_pp$pp.json 2> gpurun_out/pp_bench_c2_pp$pp.err
done
( cd bench_micro && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma4_probe.hip -o mfma4_probe 2>/dev/null && timeout 60 ./mfma4_probe ) > gpurun_out/pp_mfma4_probe.txt 2>&1
grep co-issue gpurun_out/pp_mfma4_probe.txt
cat gpurun_out/pp_summary.log; tail -5 gpurun_out/pp_pytest.log
grep -h "wave 0" gpurun_out/pp_prof_c2_pp0.err gpurun_out/pp_prof_c2_pp1.err | head -4
python - <<'PY'
import json
for pp in (0, 1):
    try:
        for l in open(f'gpurun_out/pp_bench_c2_pp{pp}.json'):
            if l.startswith('{'):
                d = json.loads(l); print('pp', pp, d['value'], 'ksamples/s', d['config']['us_per_step'], 'us/step')
    except Exception as e:
        print('pp', pp, 'no line:', e)
PY
