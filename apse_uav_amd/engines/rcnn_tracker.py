"""RcnnTracker -- counterpart of /root/reference/dcnn/engines/rcnn_tracker.py:37-221.

Same constructor signature, attributes, constants and per-frame semantics
(``next_frame(frame) -> ObjectInstances`` of the objects seen this frame):
detector -> ROI-pooled p2 features -> 128-d embedding -> squared-L2 distance matrix ->
Hungarian assignment (scipy, as in the reference) -> ``dist < 0.6`` associates, other
detections become new objects in detection order -> objects unseen for > 100 frames are
dropped.  The metric is hard-wired to 'embeddings' like the reference's ``next_frame`` (:69);
its other two branches are dead code there (undefined names).

GPU work (detector, roi_pool, association FC + normalise, distance matrix, mask centroid /
closest points) runs in ``libapse_hip.so``; the sequential id bookkeeping stays on the host.
``next_record`` drives the same association from a per-frame record, which is what rank 0
does with the records gathered from the other GPUs in frame-sharded mode.
"""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from .. import _lib
from ..networks.association_head import AssociationHead
from ..structures.instances import Boxes, Instances
from ..structures.object_instances import ObjectInstances
from ..structures.window_mask import MaskList, WindowMask
from ..utils import csv_log
from ..weights import load_association_file
from .track_predictor import TrackPredictor

# module-level constants, as in the reference (rcnn_tracker.py:32-34 keeps them outside the config too)
ASSOCIATION_ROI_SIZE = 10


class RcnnTracker:

    def __init__(self, config, image_size, weights, association_metric='embeddings', DISPLAY_INFO=[], metadata=None,
                 detector_state=None):
        self.metadata = metadata
        self.DISPLAY_INFO = DISPLAY_INFO
        self.association_metric = association_metric
        self.MASKS_IOU_THRESHOLD = 0.7
        self.ASSOCIATION_EMBEDDING_THRESHOLD = 0.6
        self.OBJECT_UNDETECTED_FRAMES_TH = 100
        self.crop_features = False
        self.config = config
        self.image_size = image_size
        self.device = torch.device(config.MODEL.DEVICE)
        self.predictor = TrackPredictor(self.config, state_dict=detector_state)
        self.backbone_features_depth = self.predictor.model.backbone.output_shape()[
            config.MODEL.ROI_HEADS.IN_FEATURES[0]].channels

        self.association_head = AssociationHead(roi_size=ASSOCIATION_ROI_SIZE, input_depth=self.backbone_features_depth)
        state = load_association_file(weights) if isinstance(weights, str) else weights
        self.association_head.load_state_dict(state)
        self.association_head.to(self.device)
        self.predictor.model.attach_association_head(self.association_head)

        self.objects = ObjectInstances(image_size=image_size, display_info=self.DISPLAY_INFO, metadata=metadata)
        self.frame_count = 0
        self._last_record = None
        self._obj_det = {}

    # ------------------------------------------------------------------ per frame
    def next_frame(self, frame, upcoming=None):
        """``upcoming`` (build extension, optional): the frame the NEXT call will be given; its host copy + H2D then
        overlap this frame's GPU work (TrackPredictor.prefetch).  Results do not depend on it."""
        self.frame_count += 1
        if 'frame_count' in self.DISPLAY_INFO: print("\nFRAME: ", self.frame_count)
        # the record path below never reads the feature dict, so the announced frame's network may be enqueued ahead (run_ahead)
        detections, backbone_features = self.predictor(frame, upcoming=upcoming, run_ahead=upcoming is not None)
        detections = detections['instances']
        return self._finish_frame(detections, backbone_features)

    def next_record(self, record):
        """Same association driven by a per-frame record (FrameResults.record)."""
        self.frame_count += 1
        return self._finish_frame(instances_from_record(record, self.image_size, self.device), None, host_replay=True)

    def _finish_frame(self, detections, backbone_features, host_replay=True):
        """``host_replay`` is accepted for older callers; since round 4 the record path always associates on the host."""
        self._last_record = getattr(detections, "_record", None)
        self._obj_det = {}
        self.associate_detections_to_objects(detections, backbone_features=backbone_features, metric='embeddings')
        self.objects.delete_undetected_objects(self.OBJECT_UNDETECTED_FRAMES_TH)
        if 'objects' in self.DISPLAY_INFO: print(self.objects)
        recent_objects = self.objects.get_recent_objects()
        if 'recent_objects' in self.DISPLAY_INFO: print('RECENT OBJECTS:\n', recent_objects)
        self.objects.finish_association()
        return recent_objects

    def associate_detections_to_objects(self, detections, backbone_features=None, metric='embeddings'):
        if 'detections' in self.DISPLAY_INFO:
            print(len(detections), ' detections:')
        if metric != 'embeddings':
            raise NotImplementedError("only the 'embeddings' metric is live in the reference (rcnn_tracker.py:69)")
        if len(detections) > 0:
            self._associate(detections, backbone_features)

    def _associate(self, detections, backbone_features):
        rec = getattr(detections, "_record", None)
        if rec is not None:
            # embeddings already computed by the fused GPU stage and delivered with the results block; they stay on the host: the
            # sequential association (this method) is host work -- O x N x 128 flops -- in next_frame as on rank 0 of a sharded run
            # (SURVEY 8e).  Round 3 sent them back to the GPU for the distance matrix (H2D -> apse_sqdist -> .cpu(): ~0.15 ms of
            # round trips per frame behind the results copy); calculate_distance_matrix still runs the kernel for device tensors.
            detection_embeddings = torch.from_numpy(np.ascontiguousarray(rec["embeddings"]))
        else:
            rois = self.get_features_rois(detections, backbone_features, crop_features=self.crop_features)
            detection_embeddings = self.association_head(rois)
        if len(self.objects) == 0:
            for detection_id in range(len(detections)):
                self.objects.add_new_object(detection_id, detections, detection_embeddings)
                self._obj_det[self.objects.ids[-1]] = detection_id
        else:
            distances = self.calculate_distance_matrix(detection_embeddings)
            dist_np = distances.cpu().detach().numpy()
            match_obj_indexes, match_det_indexes = linear_sum_assignment(dist_np)
            matched_detections = []
            for obj_idx, det_idx in zip(match_obj_indexes, match_det_indexes):
                if 'hungarian_matches' in self.DISPLAY_INFO: print('obj {} to det {}'.format(obj_idx, det_idx))
                obj_idx = int(obj_idx)
                det_idx = int(det_idx)
                if dist_np[obj_idx, det_idx] < self.ASSOCIATION_EMBEDDING_THRESHOLD:
                    self.objects.associate_detection(det_idx, obj_idx, detections, detection_embeddings)
                    self._obj_det[self.objects.ids[obj_idx]] = det_idx
                    matched_detections.append(det_idx)
            for detection_id in range(len(detections)):
                if detection_id not in matched_detections:
                    self.objects.add_new_object(detection_id, detections, detection_embeddings)
                    self._obj_det[self.objects.ids[-1]] = detection_id

    def reset_tracker(self):
        self.objects = ObjectInstances(image_size=self.image_size, display_info=self.DISPLAY_INFO)
        self.frame_count = 0

    def get_features_rois(self, detections, backbone_features, crop_features=False):
        """roi_pool of p2 at the detections' boxes (rcnn_tracker.py:156-189, crop_features=False branch)."""
        if crop_features:
            raise NotImplementedError("crop_features=True is dead at inference in the reference (rcnn_tracker.py:48)")
        features = backbone_features[self.config.MODEL.ROI_HEADS.IN_FEATURES[0]]
        spatial_scale = features.size()[3] / self.image_size[1]
        n = len(detections)
        boxes = detections.pred_boxes.tensor.to(self.device, torch.float32).contiguous()
        feat = features[:1].permute(0, 2, 3, 1).contiguous()                     # NCHW -> NHWC (plumbing)
        out = torch.empty((n, ASSOCIATION_ROI_SIZE, ASSOCIATION_ROI_SIZE, feat.shape[3]), device=self.device)
        img = torch.zeros((n,), dtype=torch.int32, device=self.device)
        _lib.check(_lib.load().apse_roi_pool(_lib.ptr(feat), feat.shape[1], feat.shape[2], _lib.ptr(boxes), _lib.ptr(img), n,
                                             ASSOCIATION_ROI_SIZE, float(spatial_scale), _lib.ptr(out), _lib.stream_ptr()),
                   None, "apse_roi_pool")
        return out.permute(0, 3, 1, 2)

    def calculate_distance_matrix(self, detection_embeddings):
        """O x N squared L2 distances (rcnn_tracker.py:192-221).  Device embeddings -> HIP kernel; host
        embeddings (record path: the GPU already produced them, the replay is pure host) -> numpy f32."""
        if not detection_embeddings.is_cuda:
            obj = np.stack([np.asarray(e, np.float32) for e in self.objects.embeddings])
            det = detection_embeddings.numpy()
            diff = obj[:, None, :] - det[None, :, :]
            return torch.from_numpy((diff * diff).sum(axis=2, dtype=np.float32))
        obj = torch.stack([e.to(self.device) for e in self.objects.embeddings]).contiguous()
        det = detection_embeddings.to(self.device).contiguous()
        out = torch.empty((obj.shape[0], det.shape[0]), device=self.device, dtype=torch.float32)
        _lib.check(_lib.load().apse_sqdist(_lib.ptr(obj), _lib.ptr(det), obj.shape[0], det.shape[0], obj.shape[1],
                                           _lib.ptr(out), _lib.stream_ptr()), None, "apse_sqdist")
        return out

    # ------------------------------------------------------------------ CSV line of the current frame
    def log_line(self, recent_objects, host_id, frame_idx):
        """generate_log_oneline (visualize_uav.py:117-141) using the GPU's closest-point table."""
        rec = self._last_record
        lookup = None
        if rec is not None:
            ids = list(recent_objects.ids) if len(recent_objects) else []

            def lookup(k, hidx):
                c = rec["closest"][self._obj_det[ids[k]], self._obj_det[ids[hidx]]]
                return (float(c[0]), float(c[1]))
        return csv_log.generate_log_oneline(recent_objects, host_id, frame_idx, lookup)


def instances_from_record(rec, frame_hw, device):
    inst = Instances(frame_hw)
    inst.pred_boxes = Boxes(torch.from_numpy(np.asarray(rec["boxes"], np.float32)))
    inst.scores = torch.from_numpy(np.asarray(rec["scores"], np.float32))
    inst.pred_classes = torch.from_numpy(np.asarray(rec["classes"], np.int64))
    inst.pred_masks = MaskList(WindowMask(None, rec["rects"][k], frame_hw, rec["centroids"][k], rec["mass"][k])
                               for k in range(len(rec["scores"])))
    inst._record = rec
    return inst
