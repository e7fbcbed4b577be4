#!/bin/bash
# round-2 GPU session I: final bench lines of the kernels as committed (configs 1, 2, 4, 3 at N = 1) + smoke
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/r2i_*
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2i_smoke.log 2>&1; echo "rc smoke $?" >> gpurun_out/r2i_summary.log
timeout 200 python bench.py --steps 3 --warmup 1 > gpurun_out/r2i_bench_c1.json 2> gpurun_out/r2i_bench_c1.err
for c in 2 4 3; do timeout 200 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2i_bench_c$c.json 2> gpurun_out/r2i_bench_c$c.err; done
WRNN_TEAM_PROF=1 timeout 120 python bench.py --config 2 --batch 64 --frames 41 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2i_prof_c2_b64.err
cat gpurun_out/r2i_summary.log
