"""Oracle for the reference's tracker + CSV log (TEST INFRASTRUCTURE).

Restates, on plain Python/numpy/torch-CPU data:
  RcnnTracker.get_features_rois          dcnn/engines/rcnn_tracker.py:156-189 (crop_features=False branch)
  AssociationHead.forward                dcnn/networks/association_head.py:16-27
  RcnnTracker.calculate_distance_matrix  dcnn/engines/rcnn_tracker.py:192-221
  associate_detections_to_objects        dcnn/engines/rcnn_tracker.py:122-147 ('embeddings' branch)
  ObjectInstances                        dcnn/structures/object_instances.py:48-176
  generate_log_oneline + CSV write       dcnn/scripts/tests/visualize_uav.py:117-141,223-233
The Hungarian step is scipy.optimize.linear_sum_assignment, as in the reference.
"""
import numpy as np
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

from . import ops
from . import mask_utils as mu

ASSOCIATION_ROI_SIZE = 10                 # rcnn_tracker.py:33
EMBEDDING_THRESHOLD = 0.6                 # rcnn_tracker.py:46
UNDETECTED_FRAMES_TH = 100                # rcnn_tracker.py:47


def features_rois(p2, boxes, image_width):
    """p2 [1, C, H, W]; boxes [N, 4] in original-frame pixels.  spatial_scale uses the
    *padded* feature width over the original width (rcnn_tracker.py:165 quirk)."""
    scale = p2.shape[3] / image_width
    rois = torch.cat([torch.zeros((boxes.shape[0], 1)), boxes.to(torch.float32)], dim=1)
    return ops.roi_pool(p2, rois, ASSOCIATION_ROI_SIZE, scale)


def association_head(rois, fc_weight, fc_bias):
    x = rois.reshape(rois.shape[0], -1)
    x = F.linear(x, fc_weight, fc_bias)
    return F.normalize(x, p=2, dim=1)


def distance_matrix(obj_emb, det_emb):
    """Literal restatement of rcnn_tracker.py:194-219 (tile, diff, bmm of diff with itself)."""
    O, N = len(obj_emb), len(det_emb)
    objs = torch.cat([torch.stack([obj_emb[i]] * N) for i in range(O)])
    dets = torch.cat([det_emb] * O)
    diffs = objs - dets
    d = torch.bmm(diffs.view(O * N, 1, -1), diffs.view(O * N, -1, 1))
    return d.view(O, N)


class TrackStore:
    """ObjectInstances semantics with plain lists (object_instances.py)."""

    def __init__(self):
        self.ids, self.frames_since, self.detected = [], [], []
        self.boxes, self.scores, self.classes, self.masks, self.emb = [], [], [], [], []
        self.assigned = []

    def __len__(self):
        return len(self.ids)

    def new_id(self):
        return 1 if not self.assigned else self.assigned[-1] + 1

    def add(self, det, i):
        nid = self.new_id()
        self.detected.append(True)
        self.ids.append(nid)
        self.frames_since.append(0)
        self.boxes.append(det["boxes"][i])
        self.scores.append(det["scores"][i])
        self.classes.append(det["classes"][i])
        self.masks.append(det["masks"][i])
        self.emb.append(det["emb"][i])
        self.assigned.append(nid)

    def associate(self, det, i, o):
        self.detected[o] = True
        self.frames_since[o] = 0
        self.boxes[o] = det["boxes"][i]
        self.classes[o] = det["classes"][i]
        self.masks[o] = det["masks"][i]
        self.emb[o] = det["emb"][i]           # score is *not* updated (object_instances.py:146-152)

    def delete_undetected(self, th):
        keep = [k for k in range(len(self)) if not (self.frames_since[k] > th)]
        for name in ("ids", "frames_since", "detected", "boxes", "scores", "classes", "masks", "emb"):
            setattr(self, name, [getattr(self, name)[k] for k in keep])

    def finish(self):
        self.frames_since = [0 if d else f + 1 for d, f in zip(self.detected, self.frames_since)]
        self.detected = [False] * len(self)

    def recent(self):
        idx = [k for k in range(len(self)) if self.detected[k]]
        return dict(ids=[self.ids[k] for k in idx], boxes=[self.boxes[k] for k in idx],
                    scores=[self.scores[k] for k in idx], classes=[self.classes[k] for k in idx],
                    masks=[self.masks[k] for k in idx], emb=[self.emb[k] for k in idx])


class TrackerOracle:
    """det dict per frame: boxes [N,4], scores [N], classes [N], masks list of (window, rect), emb [N,128]."""

    def __init__(self):
        self.store = TrackStore()
        self.frame_count = 0

    def next_frame(self, det):
        self.frame_count += 1
        n = len(det["boxes"])
        st = self.store
        if n > 0:
            if len(st) == 0:
                for i in range(n):
                    st.add(det, i)
            else:
                D = distance_matrix(st.emb, det["emb"])
                oi, di = linear_sum_assignment(D.numpy())
                matched = []
                for o, d in zip(oi, di):
                    if D[int(o), int(d)] < EMBEDDING_THRESHOLD:
                        st.associate(det, int(d), int(o))
                        matched.append(int(d))
                for i in range(n):
                    if i not in matched:
                        st.add(det, i)
        st.delete_undetected(UNDETECTED_FRAMES_TH)
        rec = st.recent()
        st.finish()
        return rec


def log_oneline(recent, host_id, frame_idx):
    """generate_log_oneline (visualize_uav.py:117-141) on oracle 'recent' dicts."""
    if len(recent["ids"]) == 0:
        return "", 0
    cents = [mu.window_centroid(w, r) for (w, r) in recent["masks"]]
    if host_id in recent["ids"]:
        hc = cents[recent["ids"].index(host_id)]
        clos = [mu.window_closest_point(w, r, hc) for (w, r) in recent["masks"]]
    else:
        clos = [("nan", "nan")] * len(recent["ids"])
    out = [str(frame_idx)]
    hi = max(recent["ids"])
    for oid in range(1, hi + 1):
        if oid in recent["ids"]:
            k = recent["ids"].index(oid)
            out += [str(cents[k][0]), str(cents[k][1]), str(clos[k][0]), str(clos[k][1])]
        else:
            out += [""] * 4
    return ",".join(out), hi


def raw_csv(lines, host_id, max_id):
    """The literal file visualize_uav.py:223-233 writes."""
    header = "frame"
    for i in range(1, max_id + 1):
        header += ",id_{} cent_x,id_{} cent_y,id_{} clos_x,id_{} clos_y".format(i, i, i, i)
    return "Ford id: {}\n".format(host_id) + header + "\n" + "\n".join(lines)
