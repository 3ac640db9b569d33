#!/bin/bash
# usage (GPU box, via gpurun): tools/gpu_profile_mode.sh <tag> <bench args...>     e.g. r01d_f16_b8 --dtype f16 --batch 8
# Kernel-trace stats, FETCH_SIZE / WRITE_SIZE and MFMA-busy PMC passes of one bench mode (BASELINE config 5 asks for
# rocprof MFMA % and HBM GB/s of the fp16 batch-8 stream).  Files under gpurun_out/<tag>_*.
set -o pipefail
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --throughput-depth 0 --no-entrypoint --no-extra-modes "$@" > $O/${tag}_bench.json 2> $O/prof_$tag.err || exit 2
cp $(find $O/prof_$tag -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv
rm -rf $O/prof_$tag
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events --throughput-depth 0 --no-entrypoint --no-extra-modes "$@" > /dev/null 2> $O/pmc_${c}_$tag.err || exit 3
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events --throughput-depth 0 --no-entrypoint --no-extra-modes "$@" > /dev/null 2> $O/pmc_mfma_$tag.err || exit 4
cd $R
python tools/pmc_summary.py $O/pmc_FETCH_SIZE_$tag $O/pmc_WRITE_SIZE_$tag $O/${tag}_pmc_traffic.json > $O/${tag}_pmc_traffic.txt 2>&1
python tools/pmc_mfma.py $O/pmc_mfma_$tag > $O/${tag}_pmc_mfma_util.txt 2>&1
rm -rf $O/pmc_FETCH_SIZE_$tag $O/pmc_WRITE_SIZE_$tag $O/pmc_mfma_$tag
echo "mode profile done"
