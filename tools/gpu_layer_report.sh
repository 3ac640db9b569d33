#!/bin/bash
# usage (on the GPU box, via gpurun): tools/gpu_layer_report.sh <tag> [bench.py args...]
# kernel-trace of a short bench run -> per-layer table in gpurun_out/layer_<tag>.txt
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-events --throughput-depth 0 --no-entrypoint --no-extra-modes "$@" > $R/gpurun_out/prof_$tag.log 2>&1
cd $R
f=$(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1)
nd=$(python - <<PY
import json
for l in open("gpurun_out/prof_$tag.log"):
    if l.startswith("{"):
        c = json.loads(l)["config"]
        print(int(round(c["detections_per_frame"])), int(c["batch_per_gpu"]))
        break
PY
)
python tools/layer_report.py $f ${nd:-8 1} --timeline > gpurun_out/layer_$tag.txt 2>&1
rm -rf gpurun_out/prof_$tag
