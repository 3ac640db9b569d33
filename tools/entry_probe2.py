#!/usr/bin/env python3
"""Per-phase host timing INSIDE TrackPredictor._predict for RcnnTracker.next_frame(frame, upcoming=next) at 3840x2160."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apse_uav_amd.config import setup_cfg
from apse_uav_amd.engines import track_predictor as tp
from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
from apse_uav_amd.synthetic import SyntheticSequence
from apse_uav_amd.weights import UAV4K_R101_CLS_BIAS, synthetic_association_state, synthetic_detector_state

H, W = 2160, 3840
sd = synthetic_detector_state(0, cls_bias=UAV4K_R101_CLS_BIAS)
tr = RcnnTracker(setup_cfg(), (H, W), synthetic_association_state(1), detector_state=sd)
seq = SyntheticSequence("static", H, W)
frames = [seq.frame(i) for i in range(8)]
pr, model = tr.predictor, tr.predictor.model
marks = []


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        marks.append((label, 1e3 * (time.perf_counter() - t)))
        return r
    setattr(obj, name, w)


for name in ("_upload", "prefetch", "_prestage"):
    wrap(pr, name, name)
for name in ("preprocess_frames", "run", "read_begin", "read_end", "instances_from"):
    wrap(model, name, "model." + name)
for mode in ("plain", "upcoming"):
    tr.reset_tracker()
    rows = []
    for i in range(30):
        marks.clear()
        t0 = time.perf_counter()
        objs = tr.next_frame(frames[i % 8], upcoming=frames[(i + 1) % 8] if mode == "upcoming" else None)
        t1 = time.perf_counter()
        tr.log_line(objs, 1, i)
        t2 = time.perf_counter()
        rows.append((1e3 * (t1 - t0), 1e3 * (t2 - t1), list(marks)))
    print(mode, "median next_frame ms", round(float(np.median([r[0] for r in rows[5:]])), 3), "log_line", round(float(np.median([r[1] for r in rows[5:]])), 3))
    for r in rows[10:13]:
        print("   ", round(r[0], 3), [(k, round(v, 3)) for k, v in r[2]])
