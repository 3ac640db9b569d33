#!/bin/bash
# usage (GPU box, via gpurun): tools/gpu_glds_ablate.sh   -- where does conv_glds16's k-loop spend its time?
# Rebuilds conv_glds16.o with GL_ABLATE = 0 (full), 1 (no MFMA), 2 (no DMA after the first two stages), 3 (no fragment reads)
# and times out2 at batch 8 (tools/conv_sweep.py, cfg 11).  Diagnostic builds: results of 1..3 are wrong by construction.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/apse_uav_amd/csrc
for a in ${ABLATE_LIST:-0 1 2 3 4}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -ffp-contract=off -DAPSE_DP16=1 -DGL_ABLATE=$a -c conv_glds16.hip -o conv_glds16.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libapse_hip.so *.o
  echo "GL_ABLATE=$a"; (cd $R && python tools/conv_sweep.py --bf16 --st16 --batch=8 out2 res4.c2 2>/dev/null | tr ' ' '\n' | grep -A1 "cfg11/sk1" | tr '\n' ' '; echo)
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -ffp-contract=off -DAPSE_DP16=1 -c conv_glds16.hip -o conv_glds16.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libapse_hip.so *.o
