#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
: > gpurun_out/r03_entry_sweep.txt
for cfg in "4 4" "8 4" "8 8" "16 8" "4 8" "2 4"; do
  set -- $cfg
  echo "== bands $1 threads $2" >> gpurun_out/r03_entry_sweep.txt
  APSE_STAGE_BANDS=$1 APSE_STAGE_THREADS=$2 python3 tools/entry_probe.py 2>&1 | grep -A1 "^plain\|^upcoming" | grep -v "^--" >> gpurun_out/r03_entry_sweep.txt
done
cat gpurun_out/r03_entry_sweep.txt
