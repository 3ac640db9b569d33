#!/bin/bash
# usage (GPU box, via gpurun): tools/gpu_lib_ab.sh <tag> <libB.so> "<mode flags>" ["<mode flags>" ...]
# Library A (tree) and library B (APSE_HIP_LIB) alternating A B A B per mode on ONE box.  Output: gpurun_out/<tag>_ab.txt
set -o pipefail
tag=$1; libb=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
: > $O/${tag}_ab.txt
run() {
  label=$1; shift
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-events --throughput-depth 0 --no-entrypoint --no-extra-modes "$@" > $O/ab_line.json 2> $O/ab_err.txt || { echo "FAILED $label $*" >> $O/${tag}_ab.txt; tail -5 $O/ab_err.txt >> $O/${tag}_ab.txt; return 1; }
  python3 -c "import json,sys; d=json.load(open('$O/ab_line.json')); print(sys.argv[1], (' '.join(sys.argv[2:]) or 'default').ljust(34), d['value'], 'frames/s, ms/step', d['ms_per_step'])" "$label" "$@" >> $O/${tag}_ab.txt
}
for args in "$@"; do
  run A $args && APSE_HIP_LIB=$R/$libb run B $args && run A $args && APSE_HIP_LIB=$R/$libb run B $args || exit 1
done
cat $O/${tag}_ab.txt
