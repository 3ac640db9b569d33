#!/bin/bash
# usage (GPU box, via gpurun): tools/gpu_profile_round.sh <tag>      e.g. r01b
# Produces under gpurun_out/: <tag>_bench_n1.json (default bench), <tag>_kernel_stats.csv + <tag>_bench_profiled_run.json
# (rocprofv3 --kernel-trace --stats of the same command), <tag>_layer_report.txt, <tag>_pmc_traffic.txt (FETCH_SIZE and
# WRITE_SIZE in separate passes).  Copy what should be judged into profiles/.
set -o pipefail
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R && python bench.py > $O/${tag}_bench_n1.json 2> $O/${tag}_bench_n1.err || exit 1
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python3 $R/bench.py --no-cpu-baseline --throughput-depth 0 --no-entrypoint --no-extra-modes > $O/${tag}_bench_profiled_run.json 2> $O/prof_$tag.err || exit 2
echo "trace done"
cd $R
cp $(find $O/prof_$tag -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv
nd=$(python -c "import json;print(int(round(json.load(open('$O/${tag}_bench_profiled_run.json'))['config']['detections_per_frame'])))")
python tools/layer_report.py $(find $O/prof_$tag -name "*kernel_trace.csv" | head -1) $nd > $O/${tag}_layer_report.txt 2>&1
rm -rf $O/prof_$tag
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f_$tag -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-events --throughput-depth 0 --no-entrypoint --no-extra-modes > /dev/null 2> $O/pmc_f_$tag.err || exit 3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_$tag -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-events --throughput-depth 0 --no-entrypoint --no-extra-modes > /dev/null 2> $O/pmc_w_$tag.err || exit 4
cd $R
python tools/pmc_summary.py $O/pmc_f_$tag $O/pmc_w_$tag $O/${tag}_pmc_traffic.json > $O/${tag}_pmc_traffic.txt 2>&1
rm -rf $O/pmc_f_$tag $O/pmc_w_$tag
echo "pmc done"
