#!/bin/bash
# usage (GPU box): tools/gpu_ab_roi.sh A B C ...   -- rocprofv3 kernel stats of ROIAlign for library variants apse_uav_amd/libapse_hip_<X>.so (A = the tree's)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $O/r03_ab_roi.txt
for v in "$@"; do
  if [ $v = A ]; then unset APSE_HIP_LIB; else export APSE_HIP_LIB=$R/apse_uav_amd/libapse_hip_$v.so; fi
  for mode in "f16b8:--dtype f16 --batch 8" "f32b1:"; do
    name=${mode%%:*}; args=${mode#*:}
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ab -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --throughput-depth 0 --no-entrypoint --no-extra-modes --no-events $args > /dev/null 2> $O/prof_ab.err || exit 2
    f=$(find $O/prof_ab -name "*kernel_stats.csv" | head -1)
    echo "== lib $v $name" >> $O/r03_ab_roi.txt
    python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'roi_align' in r['Name']: print(r['Name'][:30], r['Calls'], r['AverageNs'])" >> $O/r03_ab_roi.txt
    rm -rf $O/prof_ab
  done
done
cat $O/r03_ab_roi.txt
