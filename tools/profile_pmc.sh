#!/bin/bash
# GPU box: PMC passes for the loop kernel (LDS conflicts, wave stall split, L2 hit rate).  Counters only (no tracing).
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d $R/gpurun_out/pmc/sq -o sq -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc/sq_stdout.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $R/gpurun_out/pmc/tcc -o tcc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc/tcc_stdout.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $R/gpurun_out/pmc/inst -o inst -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc/inst_stdout.log 2>&1
cd $R; find gpurun_out/pmc -name "*.db" | head
