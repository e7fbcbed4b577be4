#!/bin/bash
# round 6: the closing session -- whole -m gpu suite, the default bench line, kernel-trace + PMC passes of the shipped kernels (COMMIT=<hash> passed in)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
export PYTHONDONTWRITEBYTECODE=1
TAG=${TAG:-r06} WHAT="tests bench copy" TEST_TIMEOUT=1500 PYTEST_ARGS="" bash tools/gpu_session.sh
TAG=${TAG:-r06} COMMIT=$COMMIT bash tools/profile_round.sh > gpurun_out/${TAG:-r06}_profile_round.log 2>&1
tail -5 gpurun_out/${TAG:-r06}_profile_round.log
