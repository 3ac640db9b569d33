#!/bin/bash
# usage (GPU box, via gpurun): tools/gpu_modes.sh <tag>
# Un-profiled, un-instrumented (--no-events) bench lines of the modes DESIGN.md section 6 tabulates -> gpurun_out/<tag>_modes.txt
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for args in "" "--batch 2" "--batch 4" "--batch 8" "--dtype bf16 --batch 1" "--dtype bf16 --batch 4" "--dtype bf16 --batch 8" "--dtype f16 --batch 8" \
            "--dtype f16 --batch 8 --pipeline 3" "--dtype bf16 --batch 4 --preproc" "--from-host"; do
  python bench.py $args --no-cpu-baseline --no-events --throughput-depth 0 --no-entrypoint --no-extra-modes 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"MODE ${args:-default (no events)} :\", d[\"value\"], \"fps, p50 ms/frame\", d[\"p50_ms_per_frame\"])"
done > gpurun_out/${tag}_modes.txt
echo "modes done"
