#!/bin/bash
# Developer probe (GPU box): phase cycles of several workgroups of team 0 (variants built with -DCS_PROF_SPLIT=1 -DCS_PROF_WG=n), configs[4] and configs[2]
cd "$(dirname "$0")/.." || exit 1
for c in "4 32" "2 64"; do
  for lib in "$@"; do
    timeout 120 python tools/phase_profile.py $c 41 batch_cs $lib 2>&1 | grep -v "^$"
  done
done
