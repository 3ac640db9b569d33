"""Configuration node for the dcnn hot path.

The reference configures the path through detectron2's yacs ``CfgNode``
(/root/reference/dcnn/scripts/tests/visualize_uav.py:43-53 ``setup_cfg``;
YAML files dcnn/configs/Base-RCNN-FPN.yaml, dcnn/configs/mask_rcnn_R_101_FPN_3x.yaml).
yacs/detectron2 are not dependencies of this build, so this module provides a
small attribute-style node with the same surface the engines use
(``get_cfg``, ``merge_from_file`` with ``_BASE_``, ``clone``, ``freeze``) and only
the keys that define results on this path.
"""
import copy
import os

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        self.__dict__["_frozen"] = False
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if self.__dict__.get("_frozen"):
            raise AttributeError("attempt to modify frozen CfgNode key %r" % k)
        self[k] = v

    def freeze(self):
        self.__dict__["_frozen"] = True
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def defrost(self):
        self.__dict__["_frozen"] = False
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    def clone(self):
        c = CfgNode(copy.deepcopy(_plain(self)))
        return c

    def _merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), CfgNode):
                self[k]._merge(v)
            else:
                self[k] = CfgNode(v) if isinstance(v, dict) else (tuple(v) if isinstance(v, list) else v)

    def merge_from_file(self, path):
        with open(path) as f:
            data = yaml.safe_load(f) or {}
        base = data.pop("_BASE_", None)
        if base:
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(path), base)
            self.merge_from_file(base)
        self._merge(data)

    def merge_from_list(self, kv):
        for k, v in zip(kv[0::2], kv[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = v


def _plain(n):
    return {k: (_plain(v) if isinstance(v, CfgNode) else v) for k, v in n.items()}


def get_cfg():
    """detectron2 0.1.2 defaults for the keys this path reads, with the R-101-FPN mask
    settings of the reference YAMLs already merged (so no YAML file is needed)."""
    return CfgNode({
        "VERSION": 2,
        "MODEL": {
            "DEVICE": "cuda",
            "META_ARCHITECTURE": "GeneralizedRCNN",
            "WEIGHTS": "",
            "MASK_ON": True,
            "PIXEL_MEAN": (103.530, 116.280, 123.675),
            "PIXEL_STD": (1.0, 1.0, 1.0),
            "BACKBONE": {"NAME": "build_resnet_fpn_backbone"},
            "RESNETS": {"DEPTH": 101, "OUT_FEATURES": ("res2", "res3", "res4", "res5"),
                        "STRIDE_IN_1X1": True, "NORM": "FrozenBN", "RES2_OUT_CHANNELS": 256, "STEM_OUT_CHANNELS": 64},
            "FPN": {"IN_FEATURES": ("res2", "res3", "res4", "res5"), "OUT_CHANNELS": 256, "FUSE_TYPE": "sum", "NORM": ""},
            "ANCHOR_GENERATOR": {"SIZES": ((32,), (64,), (128,), (256,), (512,)),
                                 "ASPECT_RATIOS": ((0.5, 1.0, 2.0),), "OFFSET": 0.0},
            "RPN": {"IN_FEATURES": ("p2", "p3", "p4", "p5", "p6"), "PRE_NMS_TOPK_TEST": 1000,
                    "POST_NMS_TOPK_TEST": 1000, "NMS_THRESH": 0.7, "BBOX_REG_WEIGHTS": (1.0, 1.0, 1.0, 1.0),
                    "MIN_SIZE": 0},
            "ROI_HEADS": {"NAME": "StandardROIHeads", "IN_FEATURES": ("p2", "p3", "p4", "p5"), "NUM_CLASSES": 80,
                          "SCORE_THRESH_TEST": 0.05, "NMS_THRESH_TEST": 0.5},
            "ROI_BOX_HEAD": {"NAME": "FastRCNNConvFCHead", "NUM_FC": 2, "FC_DIM": 1024, "POOLER_RESOLUTION": 7,
                             "POOLER_SAMPLING_RATIO": 0, "POOLER_TYPE": "ROIAlignV2",
                             "BBOX_REG_WEIGHTS": (10.0, 10.0, 5.0, 5.0)},
            "ROI_MASK_HEAD": {"NAME": "MaskRCNNConvUpsampleHead", "NUM_CONV": 4, "CONV_DIM": 256,
                              "POOLER_RESOLUTION": 14, "POOLER_SAMPLING_RATIO": 0, "POOLER_TYPE": "ROIAlignV2"},
        },
        "INPUT": {"MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333, "FORMAT": "BGR"},
        "TEST": {"DETECTIONS_PER_IMAGE": 100},
        "DATASETS": {"TRAIN": ("coco_2017_train",), "TEST": ("coco_2017_val",)},
        # build-specific knobs (not in the reference): storage dtype and batch of the HIP path
        "APSE": {"DTYPE": "f32", "MAX_BATCH": 1, "FUSED_PREPROC": True, "STORAGE16": True},
    })


CLASSES_NAMES = ["car", "truck", "bus", "person"]          # visualize_uav.py:31


def setup_cfg(weights="", score_thresh=0.5, num_classes=4, device="cuda"):
    """Counterpart of visualize_uav.py:43-53."""
    cfg = get_cfg()
    cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = score_thresh
    cfg.MODEL.ROI_HEADS.NUM_CLASSES = num_classes
    cfg.MODEL.WEIGHTS = weights
    cfg.MODEL.MASK_ON = True
    cfg.MODEL.DEVICE = device
    return cfg
