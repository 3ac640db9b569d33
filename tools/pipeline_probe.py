#!/usr/bin/env python3
"""PipelinedRcnnTracker at several depths: frames/s and the distribution of collect() intervals (diagnostic for the depth-6
collapse noted in engines/pipelined_tracker.py).  usage: python tools/pipeline_probe.py 3 4 6"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apse_uav_amd.config import setup_cfg
from apse_uav_amd.engines import pipelined_tracker as pt
from apse_uav_amd.synthetic import SyntheticSequence
from apse_uav_amd.utils.hostinfo import usable_cpus
from apse_uav_amd.weights import UAV4K_R101_CLS_BIAS, synthetic_association_state, synthetic_detector_state

torch.set_num_threads(min(torch.get_num_threads(), usable_cpus()))
H, W = 2160, 3840
sd = synthetic_detector_state(0, cls_bias=UAV4K_R101_CLS_BIAS)
asd = synthetic_association_state(1)
seq = SyntheticSequence("static", H, W)
frames = torch.stack([torch.from_numpy(seq.frame(i)) for i in range(4)]).cuda()
pt.MAX_DEPTH = 16
for depth in [int(a) for a in sys.argv[1:]] or [3, 4, 6]:
    drv = pt.PipelinedRcnnTracker(setup_cfg(), (H, W), asd, depth=depth, detector_state=sd)
    stamps = []
    n = 60
    for i in range(n):
        if len(drv._inflight) == depth:
            drv.collect()
            stamps.append(time.perf_counter())
        drv.submit(frames[i % 4])
    while drv._inflight:
        drv.collect()
        stamps.append(time.perf_counter())
    d = np.diff(np.array(stamps[10:])) * 1e3
    print("depth %d: %.1f frames/s; collect interval ms: median %.2f p90 %.2f max %.2f; queues env %s" % (
        depth, 1000.0 / d.mean(), np.median(d), np.percentile(d, 90), d.max(), os.environ.get("GPU_MAX_HW_QUEUES", "default")))
    del drv
    torch.cuda.synchronize()
