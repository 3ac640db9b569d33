"""RoiFeaturesGenerator -- counterpart of /root/reference/dcnn/engines/roi_features_generator.py:16-131.

Generates the RoI features of ground-truth objects that the association head is trained on: the tracker's
backbone (stem + res2..res5 + FPN) on one frame, then on ``p2`` (``in_features[0]``, :92) either
``torchvision.ops.roi_pool`` (:113) or, when instance masks are given, ``p2 * bilinear-resized mask`` followed by
``roi_align(aligned=False, sampling_ratio=4)`` (:93-111).  Same constructor and call contract:

    gen = RoiFeaturesGenerator(config, roi_size=8)
    ids, rois = gen.get_rois_features(original_image, objects, objects_masks=None)
    # objects rows: <frame>, <id>, <bb_left>, <bb_top>, <bb_width>, <bb_height>, <conf>;  rois: [N, C, roi_size, roi_size]

The frame goes through the same HIP path as the tracker (PIL-exact resize + normalise + pad, implicit-GEMM
backbone); the RoI stage is ``apse_roi_features`` (include/apse_hip.h).  ``objects_masks`` may be COCO RLE dicts
(what the reference passes, decoded here by utils/rle.py instead of pycocotools), or bool/u8 arrays [H, W].
Only backbone weights are needed: a checkpoint's non-backbone tensors may be absent (PartialCheckpointer,
dcnn/utils/partial_checkpointer.py:8-24, keeps the ``backbone.`` part only).
"""
import numpy as np
import torch

from .. import _lib
from ..networks.track_rcnn import TrackRCNN
from ..utils import rle
from ..weights import load_detector_file, synthetic_detector_state, blocks_from_state


class RoiFeaturesGenerator:
    def __init__(self, config, roi_size=8, state_dict=None):
        self.device = torch.device(config.MODEL.DEVICE)
        self.roi_size = roi_size
        self.in_features = config.MODEL.ROI_HEADS.IN_FEATURES
        self.cfg = config.clone()
        self.cfg.APSE.MAX_BATCH = 1
        self.model = TrackRCNN(self.cfg)
        self.model.to(self.device)
        if state_dict is None and config.MODEL.WEIGHTS:
            state_dict = load_detector_file(config.MODEL.WEIGHTS)
        if state_dict is not None:
            self.load_backbone(state_dict)
        self._staging = None

    def load_backbone(self, state_dict):
        """Keeps the ``backbone.*`` tensors of a detector checkpoint (keys with or without the prefix, as
        PartialCheckpointer strips it) and zero-fills the heads this generator never runs."""
        sd = {}
        for k, v in state_dict.items():
            if k.startswith("backbone."):
                sd[k] = v
            elif k.startswith("bottom_up.") or k.startswith("fpn_"):
                sd["backbone." + k] = v
        full = synthetic_detector_state(0, blocks_from_state(sd), num_classes=self.cfg.MODEL.ROI_HEADS.NUM_CLASSES)
        for k, v in full.items():
            if k not in sd:
                sd[k] = torch.zeros_like(v)
        self.model.load_state_dict(sd)

    def get_features_depth(self):
        return self.model.backbone.output_shape()[self.in_features[0]].channels

    def _masks_dense(self, objects_masks, height, width):
        out = np.zeros((len(objects_masks), height, width), dtype=np.uint8)
        for k, m in enumerate(objects_masks):
            if isinstance(m, dict):
                m = rle.decode(m)
            m = np.asarray(m.cpu() if hasattr(m, "cpu") else m)
            assert m.shape == (height, width), "mask %d has shape %s, frame is %s" % (k, m.shape, (height, width))
            out[k] = m != 0
        return out

    def get_rois_features(self, original_image, objects, objects_masks=None):
        """original_image: HxWx3 uint8 (BGR, as read by cv2); returns (ids tensor [N], rois tensor [N, C, S, S]) on the device."""
        objects = np.asarray(objects, dtype=np.float64).reshape(-1, 7) if len(objects) else np.zeros((0, 7))
        height, width = original_image.shape[:2]
        n = objects.shape[0]
        C = 256
        ids = torch.as_tensor(objects[:, 1], dtype=torch.float32).to(self.device)
        rois = torch.zeros((n, C, self.roi_size, self.roi_size), dtype=torch.float32, device=self.device)
        if n == 0:
            return ids, rois
        if self._staging is None or self._staging.shape != (1, height, width, 3):
            self._staging = torch.empty((1, height, width, 3), dtype=torch.uint8).pin_memory()
        self._staging[0].copy_(torch.from_numpy(np.ascontiguousarray(original_image)))
        frame = self._staging.to(self.device, non_blocking=True)
        m = self.model
        m.preprocess_frames(frame)
        s = _lib.stream_ptr()
        m._call("apse_backbone", 1, s)
        bb = objects[:, 2:6]
        boxes = np.stack([bb[:, 0], bb[:, 1], bb[:, 0] + bb[:, 2], bb[:, 1] + bb[:, 3]], axis=1).astype(np.float32)   # :91
        boxes_d = torch.from_numpy(boxes).to(self.device)
        masks_d = None
        if objects_masks is not None:
            assert len(objects_masks) == n, "one mask per object"
            masks_d = torch.from_numpy(self._masks_dense(objects_masks, height, width)).to(self.device)
        m._call("apse_roi_features", 0, _lib.ptr(boxes_d), _lib.ptr(masks_d) if masks_d is not None else None, n, self.roi_size,
                _lib.ptr(rois), s)
        torch.cuda.synchronize()
        return ids, rois
