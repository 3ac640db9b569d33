#!/usr/bin/env python3
"""One 4K frame through the tracker: prints the detections' paste windows and masses (what the mask tail kernels work on)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apse_uav_amd.config import setup_cfg
from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
from apse_uav_amd.synthetic import SyntheticSequence
from apse_uav_amd.weights import UAV4K_R101_CLS_BIAS, synthetic_association_state, synthetic_detector_state

H, W = 2160, 3840
sd = synthetic_detector_state(0, cls_bias=UAV4K_R101_CLS_BIAS)
tr = RcnnTracker(setup_cfg(), (H, W), synthetic_association_state(1), detector_state=sd)
seq = SyntheticSequence(sys.argv[1] if len(sys.argv) > 1 else "static", H, W)
objs = tr.next_frame(seq.frame(0))
rec = tr._last_record
for k in sorted(rec.keys()):
    v = np.asarray(rec[k])
    if k in ("rects", "rect", "mask_rects", "boxes", "mass", "masses", "centroids"):
        print(k, v.tolist())
print("keys", sorted(rec.keys()))
