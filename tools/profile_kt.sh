#!/bin/bash
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'TAG=r03b bash tools/profile_kt.sh'
# GPU box: rocprofv3 --kernel-trace --stats of the bench command for configs 1, 2, 4 only (no counters): the per-kernel average durations the
# bench line's roofline objects must agree with.  Summary (tools/pmc_summary.py) -> gpurun_out/prof_<TAG>/kt_summary.txt
mkdir -p gpurun_out/prof_${TAG:-rXX}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in 1 2 4; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG:-rXX}/kt_c$c -o kt -- python $R/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > $R/gpurun_out/prof_${TAG:-rXX}/kt_c${c}_stdout.log 2>&1
  grep -h '^{' $R/gpurun_out/prof_${TAG:-rXX}/kt_c${c}_stdout.log | cut -c1-260
done
cd $R
python tools/pmc_summary.py gpurun_out/prof_${TAG:-rXX} > gpurun_out/prof_${TAG:-rXX}/kt_summary.txt 2>&1; grep -A4 "^== " gpurun_out/prof_${TAG:-rXX}/kt_summary.txt | cut -c1-200
