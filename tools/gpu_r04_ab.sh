#!/bin/bash
# usage (GPU box, via gpurun): tools/gpu_r04_ab.sh <tag> <libB.so>
# Round-4 A/B of the 16-bit configurations on ONE box: library A (tree) / library B (APSE_HIP_LIB) / A without the 128x128 LDS-DMA
# tile (APSE_GLDS_NO128), each mode A B A B; then rocprofv3 kernel stats of configs[2] and configs[4] with library A.
set -o pipefail
tag=$1; libb=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
: > $O/${tag}_ab.txt
run() {
  label=$1; shift
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-events --throughput-depth 0 --no-entrypoint --no-extra-modes "$@" > $O/ab_line.json 2> $O/ab_err.txt || { echo "FAILED $label $*" >> $O/${tag}_ab.txt; tail -5 $O/ab_err.txt >> $O/${tag}_ab.txt; return 1; }
  python3 -c "import json,sys; d=json.load(open('$O/ab_line.json')); print('MODE', sys.argv[1], ' '.join(sys.argv[2:]), ':', d['value'], 'fps, ms/step', d['ms_per_step'])" "$label" "$@" >> $O/${tag}_ab.txt
}
for args in "--dtype bf16 --batch 4 --preproc" "--dtype f16 --batch 8" "--dtype bf16 --batch 4" "--dtype bf16 --batch 1"; do
  run A $args && APSE_HIP_LIB=$R/$libb run B $args && run A $args && APSE_HIP_LIB=$R/$libb run B $args && APSE_GLDS_NO128=1 run A-no128 $args || exit 1
done
cd /tmp && export TMPDIR=/tmp
for mode in "bf16b4pre:--dtype bf16 --batch 4 --preproc" "f16b8:--dtype f16 --batch 8"; do
  name=${mode%%:*}; args=${mode#*:}
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-events --throughput-depth 0 --no-entrypoint --no-extra-modes $args > /dev/null 2> $O/prof_$tag.err || exit 2
  cp $(find $O/prof_$tag -name "*kernel_stats.csv" | head -1) $O/${tag}_${name}_stats.csv
  cp $(find $O/prof_$tag -name "*kernel_trace.csv" | head -1) $O/${tag}_${name}_trace.csv 2>/dev/null
  rm -rf $O/prof_$tag
done
echo "ab done"
