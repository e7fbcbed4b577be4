#!/bin/bash
# GPU box: bench line + rocprofv3 kernel stats (+ one PMC pass for HBM traffic) -> gpurun_out/
set -x
mkdir -p gpurun_out/prof
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_line.json 2> gpurun_out/bench_err.log
cat gpurun_out/bench_line.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/kt_stdout.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_fetch -o fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_fetch_stdout.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_write -o write -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_write_stdout.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -type f | head -40
for f in $(find gpurun_out/prof/kt -name "*kernel_stats*"); do echo == $f; head -20 $f; done
