"""Dumps the box-head logits/proposals of one synthetic 4K frame (HIP path) for weight calibration."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apse_uav_amd.config import setup_cfg
from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
from apse_uav_amd.synthetic import SyntheticSequence
from apse_uav_amd.weights import synthetic_association_state, synthetic_detector_state
sd = synthetic_detector_state(0, bg_bias=0.0)
tr = RcnnTracker(setup_cfg(), (2160, 3840), synthetic_association_state(1), detector_state=sd)
seq = SyntheticSequence("static", 2160, 3840)
m = tr.predictor.model
out = {}
for t in range(2):
    tr.next_frame(seq.frame(t))
    out["pred%d" % t] = m.debug_tensor("box_pred").cpu().numpy().reshape(-1, 32)[:1000]
    out["props%d" % t] = m.debug_tensor("proposals").cpu().numpy().reshape(-1, 4)[:1000]
np.savez(os.path.join("gpurun_out", "logits.npz"), **out)
print("ok")
