#!/bin/bash
#   /usr/local/graft/bin/gpurun --timeout 900 -- "COMMIT=$(git log -1 --format=%h) TAG=r05 bash tools/profile_round.sh"   (copy pmc_summary.txt -> profiles/<TAG>_rocprofv3_summary.txt, pmc.json -> profiles/<TAG>_pmc.json)
# GPU box: rocprofv3 kernel-trace stats of the bench command for configs 1, 2, 4 and PMC passes (HBM traffic; wave-cycle
# split, LDS conflicts) for the dominant loop kernels.  Counters are collected in their own runs (no tracing).
mkdir -p gpurun_out/prof_${TAG:-rXX}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in 1 2 4; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG:-rXX}/kt_c$c -o kt -- python $R/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > $R/gpurun_out/prof_${TAG:-rXX}/kt_c${c}_stdout.log 2>&1
done
for c in 1 2 4; do
  rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_${TAG:-rXX}/fetch_c$c -o fetch -- python $R/bench.py --config $c --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs > $R/gpurun_out/prof_${TAG:-rXX}/fetch_c${c}_stdout.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_${TAG:-rXX}/write_c$c -o write -- python $R/bench.py --config $c --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs > $R/gpurun_out/prof_${TAG:-rXX}/write_c${c}_stdout.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES -d $R/gpurun_out/prof_${TAG:-rXX}/sq_c$c -o sq -- python $R/bench.py --config $c --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs > $R/gpurun_out/prof_${TAG:-rXX}/sq_c${c}_stdout.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/prof_${TAG:-rXX}/inst_c$c -o inst -- python $R/bench.py --config $c --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs > $R/gpurun_out/prof_${TAG:-rXX}/inst_c${c}_stdout.log 2>&1
done
cd $R
for f in $(find gpurun_out/prof_${TAG:-rXX} -name "*kernel_stats*csv"); do echo "== $f"; head -8 $f; done
python tools/pmc_summary.py gpurun_out/prof_${TAG:-rXX} > gpurun_out/prof_${TAG:-rXX}/pmc_summary.txt 2>&1; cat gpurun_out/prof_${TAG:-rXX}/pmc_summary.txt
# the per-launch counter numbers bench.py attaches to its roofline objects, stamped with COMMIT (pass it: the GPU box has no .git) and the kernel-source hash
python tools/pmc_summary.py gpurun_out/prof_${TAG:-rXX} --json gpurun_out/prof_${TAG:-rXX}/pmc.json
