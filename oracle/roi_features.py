"""TEST INFRASTRUCTURE (see oracle/__init__.py) -- CPU restatement of RoiFeaturesGenerator.get_rois_features
(/root/reference/dcnn/engines/roi_features_generator.py:68-117) on top of the detector oracle's backbone.

parity unpinned: the reference needs detectron2 / torchvision / pycocotools, none of which is importable here.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops


def get_rois_features(det_oracle, resized_chw, frame_hw, objects, masks=None, roi_size=8):
    """resized_chw: the frame after ResizeShortestEdge (:77-78; PIL resize, restated bit-exactly elsewhere) as a
    float CHW tensor; frame_hw: (height, width) of the original frame.
    objects rows: <frame>, <id>, <bb_left>, <bb_top>, <bb_width>, <bb_height>, <conf> (:76);
    masks: optional bool / u8 array [N, H, W] (the reference decodes COCO RLE dicts first, :93-97).
    Returns (ids float tensor [N], rois [N, C, roi_size, roi_size])."""
    objects = np.asarray(objects, dtype=np.float64)
    height, width = frame_hw
    x = det_oracle.preprocess(resized_chw)                              # :79-81 normalise + pad
    p2 = det_oracle.backbone(x)["p2"]                                   # in_features[0] (:92)
    bb = objects[:, 2:6]
    boxes = torch.tensor([[0, b[0], b[1], b[0] + b[2], b[1] + b[3]] for b in bb], dtype=torch.float32)   # :91
    spatial_scale = p2.shape[3] / width                                 # :105
    if masks is not None:
        m = torch.as_tensor(np.asarray(masks)).float()
        resized = F.interpolate(m.view(-1, 1, m.shape[1], m.shape[2]), size=(p2.shape[2], p2.shape[3]), mode="bilinear",
                                align_corners=False)                    # :99-101
        cropped = p2.expand(resized.shape[0], -1, -1, -1) * resized     # :102-104
        boxes[:, 0] = torch.arange(cropped.shape[0], dtype=torch.float32)   # :105-106
        rois = ops.roi_align_legacy(cropped, boxes, roi_size, spatial_scale, 4)   # :111
    else:
        rois = ops.roi_pool(p2, boxes, roi_size, spatial_scale)         # :113
    return torch.tensor(objects[:, 1], dtype=torch.float32), rois
