#!/bin/bash
# usage (GPU box, via gpurun): tools/gpu_layer_probe.sh <tag> <prec> "<layer names>" [ENV=VALUE[,ENV=VALUE] | -] ...
# tools/layer_probe.py for every layer under every setting, twice.  Output: gpurun_out/<tag>_layers.txt
set -o pipefail
tag=$1; prec=$2; names=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
: > $O/${tag}_layers.txt
for rep in 1 2; do
  for name in $names; do
    for setting in "$@"; do
      if [ "$setting" = "-" ]; then envs=(); else IFS=',' read -ra envs <<< "$setting"; fi
      printf "%-44s " "$setting" >> $O/${tag}_layers.txt
      env "${envs[@]}" timeout -k 10 120 python3 tools/layer_probe.py $name $prec >> $O/${tag}_layers.txt 2> $O/lp_err.txt || { tail -5 $O/lp_err.txt >> $O/${tag}_layers.txt; exit 1; }
    done
  done
done
cat $O/${tag}_layers.txt
