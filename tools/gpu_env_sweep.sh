#!/bin/bash
# usage (GPU box, via gpurun): tools/gpu_env_sweep.sh <tag> "<bench.py mode flags>" [ENV=VALUE | -] ...
# One bench line per setting ("-" = no variable), the whole list twice in the given order (boxes and runs differ by 1-3 %:
# compare settings within one call only).  Output: gpurun_out/<tag>_sweep.txt
set -o pipefail
tag=$1; args=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
: > $O/${tag}_sweep.txt
for rep in 1 2; do
  for setting in "$@"; do
    if [ "$setting" = "-" ]; then envs=(); else IFS=',' read -ra envs <<< "$setting"; fi
    env "${envs[@]}" timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-events --throughput-depth 0 --no-entrypoint --no-extra-modes $args > $O/sw_line.json 2> $O/sw_err.txt \
      || { echo "FAILED $setting" >> $O/${tag}_sweep.txt; tail -5 $O/sw_err.txt >> $O/${tag}_sweep.txt; exit 1; }
    python3 -c "import json,sys; d=json.load(open('$O/sw_line.json')); print(sys.argv[1].ljust(40), d['value'], 'frames/s, ms/step', d['ms_per_step'])" "$setting" >> $O/${tag}_sweep.txt
  done
done
cat $O/${tag}_sweep.txt
