#!/bin/bash
# usage (GPU box, via gpurun): tools/gpu_ab.sh <tag> [libB.so]
# Quick A/B after a kernel change: un-profiled bench lines of the main modes -> gpurun_out/<tag>_modes.txt.  With a second
# library (another build of the same sources, e.g. `make NT=0`; loaded through APSE_HIP_LIB) every mode is run A, B, A, B on the
# SAME box -- box-to-box spread is ~2 %, more than most single changes.  Then a kernel-stats pass of fp16 batch 8 (library A).
set -o pipefail
tag=$1; libb=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
: > $O/${tag}_modes.txt
run() {
  python3 bench.py --no-cpu-baseline --throughput-depth 0 --no-entrypoint --no-extra-modes "$@" > $O/ab_line.json 2> $O/ab_err.txt || { echo "FAILED $*" >> $O/${tag}_modes.txt; tail -5 $O/ab_err.txt >> $O/${tag}_modes.txt; return 1; }
  python3 -c "import json,sys,os; d=json.load(open('$O/ab_line.json')); print('MODE', os.environ.get('APSE_HIP_LIB','A').split('/')[-1], ' '.join(sys.argv[1:]) or 'default', ':', d['value'], 'fps, p50', d.get('p50_ms_per_frame'))" "$@" >> $O/${tag}_modes.txt
}
both() {
  if [ -n "$libb" ]; then
    run "$@" && APSE_HIP_LIB=$R/$libb run "$@" && run "$@" && APSE_HIP_LIB=$R/$libb run "$@"
  else
    run "$@"
  fi
}
both && both --batch 8 && both --dtype bf16 --batch 1 && both --dtype bf16 --batch 4 && both --dtype f16 --batch 8 || exit 1
[ -n "$libb" ] && { echo "ab done"; exit 0; }
run --dtype bf16 --batch 4 --preproc || exit 1
cd /tmp && export TMPDIR=/tmp
# kernel-stats passes (library A): GPU time per step is a steadier yardstick than frames/s
for mode in "f16b8:--dtype f16 --batch 8" "bf16b1:--dtype bf16 --batch 1" "f32b1:"; do
  name=${mode%%:*}; args=${mode#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --throughput-depth 0 --no-entrypoint --no-extra-modes $args > /dev/null 2> $O/prof_$tag.err || exit 2
  cp $(find $O/prof_$tag -name "*kernel_stats.csv" | head -1) $O/${tag}_${name}_stats.csv
  rm -rf $O/prof_$tag
  python3 -c "
import csv,sys
rows=list(csv.DictReader(open('$O/${tag}_${name}_stats.csv')))
print('KERNEL-TIME $name: %.1f us per step (14 steps)' % (sum(float(r['TotalDurationNs']) for r in rows)/1e3/14))" >> $O/${tag}_modes.txt
done
echo "ab done"
