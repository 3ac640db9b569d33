#!/bin/bash
# usage (GPU box): tools/gpu_pmc_conv.sh <layer> <cfg> <splitk> <prec> -> gpurun_out/pmc_<layer>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_$1 -- python3 $R/tools/conv_one.py $1 $2 $3 $4 6 > $R/gpurun_out/pmc_$1.log 2>&1
cd $R && python tools/pmc_mfma.py gpurun_out/pmc_$1 > gpurun_out/pmc_$1.txt 2>&1; rm -rf gpurun_out/pmc_$1
