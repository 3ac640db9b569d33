#!/bin/bash
# usage (GPU box): tools/gpu_ab_kernels.sh <tag> <name regex> A B C ...
# rocprofv3 kernel stats (calls, average ns) of the kernels matching the regex, f32 batch 1 and fp16 batch 8, for library variants
# apse_uav_amd/libapse_hip_<X>.so (A = the tree's library) -> gpurun_out/<tag>_ab.txt
tag=$1; rx=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $O/${tag}_ab.txt
for v in "$@"; do
  if [ $v = A ]; then unset APSE_HIP_LIB; else export APSE_HIP_LIB=$R/apse_uav_amd/libapse_hip_$v.so; fi
  for mode in "f16b8:--dtype f16 --batch 8" "f32b1:"; do
    name=${mode%%:*}; args=${mode#*:}
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ab -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --throughput-depth 0 --no-entrypoint --no-extra-modes --no-events $args > $O/prof_ab.json 2> $O/prof_ab.err || exit 2
    f=$(find $O/prof_ab -name "*kernel_stats.csv" | head -1)
    echo "== lib $v $name" >> $O/${tag}_ab.txt
    python3 - "$f" "$rx" >> $O/${tag}_ab.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows:
    if re.search(sys.argv[2], r['Name']):
        print('  %-44s calls %5s  avg %9.1f ns' % (r['Name'].split('(')[0][:44], r['Calls'], float(r['AverageNs'])))
        tot += float(r['TotalDurationNs'])
print('  matched total per step: %.1f us;  all kernels per step: %.1f us' % (tot / 1e3 / 14, sum(float(r['TotalDurationNs']) for r in rows) / 1e3 / 14))
PY
    rm -rf $O/prof_ab
  done
done
cat $O/${tag}_ab.txt
