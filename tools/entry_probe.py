#!/usr/bin/env python3
"""Per-phase host timing of RcnnTracker.next_frame(np.ndarray) at 3840x2160 (diagnostic for bench.py's `entrypoint`)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apse_uav_amd.config import setup_cfg
from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
from apse_uav_amd.synthetic import SyntheticSequence
from apse_uav_amd.weights import UAV4K_R101_CLS_BIAS, synthetic_association_state, synthetic_detector_state

H, W = 2160, 3840
try:
    quota = open("/sys/fs/cgroup/cpu.max").read().strip()
except OSError:
    quota = "n/a"
print("cpus: os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "cgroup cpu.max", quota, "torch threads", torch.get_num_threads())
sd = synthetic_detector_state(0, cls_bias=UAV4K_R101_CLS_BIAS)
tr = RcnnTracker(setup_cfg(), (H, W), synthetic_association_state(1), detector_state=sd)
seq = SyntheticSequence("static", H, W)
frames = [seq.frame(i) for i in range(8)]
pr, model = tr.predictor, tr.predictor.model
resident = torch.from_numpy(frames[0]).cuda()[None]
for mode in ("plain", "upcoming", "nomask", "resident_masks", "resident_nomask"):
    tr.reset_tracker()
    rows = []
    for i in range(30):
        t = [time.perf_counter()]
        dev = resident if mode.startswith("resident") else pr._upload([frames[i % 8]])
        t.append(time.perf_counter())
        B = model.preprocess_frames(dev)
        pr._frames_consumed()
        model.run(B)
        t.append(time.perf_counter())
        if mode == "upcoming":
            pr.prefetch(frames[(i + 1) % 8])
        t.append(time.perf_counter())
        res = model.read(B)
        t.append(time.perf_counter())
        inst = model.instances_from(res, 0, not mode.endswith("nomask"))
        t.append(time.perf_counter())
        tr.frame_count += 1
        objs = tr._finish_frame(inst, None)
        tr.log_line(objs, 1, i)
        t.append(time.perf_counter())
        rows.append([1000 * (b - a) for a, b in zip(t[:-1], t[1:])])
    a = np.array(rows[3:])
    print(mode, "phases: upload, enqueue, prefetch, read(sync), instances_from, associate+line")
    print("  median ms", np.round(np.median(a, 0), 3), "total", round(float(np.median(a.sum(1))), 3))
    print("  max    ms", np.round(a.max(0), 3), "total", round(float(a.sum(1).max()), 3))
    print("  per-frame totals", np.round(a.sum(1), 2))
