#!/bin/bash
# usage (GPU box, via gpurun): tools/gpu_r04_kstats.sh <tag> <lib.so> [<lib.so> ...]
# rocprofv3 kernel stats of configs[2] (bf16 batch 4 + preproc) and configs[4] (fp16 batch 8) for each library on ONE box:
# gpurun_out/<tag>_<lib>_<mode>_stats.csv and a one-line-per-kernel summary of the convolution kernels in gpurun_out/<tag>_kstats.txt
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
: > $O/${tag}_kstats.txt
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  for mode in "bf16b4pre:--dtype bf16 --batch 4 --preproc" "f16b8:--dtype f16 --batch 8"; do
    m=${mode%%:*}; args=${mode#*:}
    APSE_HIP_LIB=$R/$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-events --throughput-depth 0 --no-entrypoint --no-extra-modes $args > $O/${tag}_${name}_${m}.json 2> $O/prof_$tag.err || { echo "FAILED $name $m" >> $O/${tag}_kstats.txt; tail -3 $O/prof_$tag.err >> $O/${tag}_kstats.txt; continue; }
    cp $(find $O/prof_$tag -name "*kernel_stats.csv" | head -1) $O/${tag}_${name}_${m}_stats.csv
    rm -rf $O/prof_$tag
    python3 - "$O/${tag}_${name}_${m}_stats.csv" "$name $m" >> $O/${tag}_kstats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 14e3
print("== %s: %.1f us of kernels per step" % (sys.argv[2], tot))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:14]:
    print("   %-70s calls %5s avg %8.1f us  total/step %8.1f us" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 14e3))
PY
  done
done
echo "kstats done"
