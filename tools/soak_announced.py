#!/usr/bin/env python3
"""Soak of the announced loop: N frames of the dynamic 4K sequence through next_frame(f_t, upcoming=f_{t+1}) (every forward enqueued
one call ahead of its results, mask bit planes alternating) against the plain loop: ids, CSV line and packed record of every frame."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apse_uav_amd.config import setup_cfg
from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
from apse_uav_amd.sharding import pack_record
from apse_uav_amd.synthetic import SyntheticSequence
from apse_uav_amd.weights import UAV4K_R101_CLS_BIAS, synthetic_association_state, synthetic_detector_state

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
H, W = 2160, 3840
sd = synthetic_detector_state(0, cls_bias=UAV4K_R101_CLS_BIAS)
asd = synthetic_association_state(1)
seq = SyntheticSequence("dynamic", H, W)
a = RcnnTracker(setup_cfg(), (H, W), asd, detector_state=sd)
b = RcnnTracker(setup_cfg(), (H, W), asd, detector_state=sd)
cur = seq.frame(0)
bad = 0
for t in range(N):
    nxt = seq.frame(t + 1) if t + 1 < N else None
    oa = a.next_frame(cur)
    ra = (list(oa.ids) if len(oa) else [], a.log_line(oa, 1, t)[0], pack_record(a._last_record, 100, 128).tobytes())
    ob = b.next_frame(cur, upcoming=nxt)
    rb = (list(ob.ids) if len(ob) else [], b.log_line(ob, 1, t)[0], pack_record(b._last_record, 100, 128).tobytes())
    if ra != rb:
        bad += 1
        print("frame", t, "differs", ra[0], rb[0])
    cur = nxt
print("soak: %d frames, %d differing" % (N, bad))
sys.exit(1 if bad else 0)
