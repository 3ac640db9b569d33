#!/bin/bash
# usage (GPU box): tools/gpu_pmc_tail.sh <tag> [bench args...]   -- SQ pass (MFMA busy, wave-time breakdown) of EVERY kernel of a bench run
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_tail_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events --throughput-depth 0 --no-entrypoint --no-extra-modes "$@" > /dev/null 2> $O/pmc_tail_$tag.err || exit 4
cd $R
python tools/pmc_mfma.py $O/pmc_tail_$tag --all > $O/${tag}_pmc_all_kernels.txt 2>&1
rm -rf $O/pmc_tail_$tag
echo "tail pmc done"
