#!/bin/bash
# round-2 GPU session L (last of the round): the MFMA loops software pipelined by hand (LDS operands of slab S + D requested
# before the MFMAs of slab S).  Phase cycles, bench lines of all four configs, kernel-trace + HBM counters of config 2 on the
# shipped kernels, then the whole GPU suite.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rm -rf gpurun_out/r2l_* gpurun_out/prof_r2l
for cfg in "2 64" "2 32"; do set -- $cfg
  WRNN_TEAM_PROF=1 timeout 120 python bench.py --config $1 --batch $2 --frames 41 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2l_prof_c$1_b$2.err
done
for c in 2 4 1 3; do
  timeout 200 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2l_bench_c$c.json 2> gpurun_out/r2l_bench_c$c.err
done
R=$PWD
mkdir -p gpurun_out/prof_r2l
( cd /tmp && export TMPDIR=/tmp
  timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2l/kt_c2 -o kt -- python $R/bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_r2l/kt_c2_stdout.log 2>&1
  timeout 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r2l/fetch_c2 -o fetch -- python $R/bench.py --config 2 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_r2l/fetch_c2_stdout.log 2>&1
  timeout 120 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r2l/write_c2 -o write -- python $R/bench.py --config 2 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_r2l/write_c2_stdout.log 2>&1
  timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES -d $R/gpurun_out/prof_r2l/sq_c2 -o sq -- python $R/bench.py --config 2 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_r2l/sq_c2_stdout.log 2>&1
)
python tools/pmc_summary.py gpurun_out/prof_r2l > gpurun_out/r2l_pmc_summary.txt 2>&1
find gpurun_out/prof_r2l -name "*.db" -size +8M -delete
timeout 600 python -m pytest tests -x -q -m gpu --durations=3 > gpurun_out/r2l_pytest_gpu.log 2>&1
echo "rc pytest_gpu $?" >> gpurun_out/r2l_summary.log
cat gpurun_out/r2l_summary.log; tail -3 gpurun_out/r2l_pytest_gpu.log
