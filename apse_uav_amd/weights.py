"""Weights for the dcnn hot path: key names, synthetic generator, file loaders.

Key names follow detectron2 0.1.2's GeneralizedRCNN state_dict as evidenced in the
reference (e.g. /root/reference/dcnn/utils/partial_checkpointer.py:15
``backbone.``, dcnn/scripts/add_mask_head_to_frcnn.py:57-73 ``roi_heads.mask_head.*``) and
``fc.weight`` / ``fc.bias`` for the association head
(/root/reference/dcnn/networks/association_head.py:13).

No weight file ships with the reference, so tests and the benchmark use seeded
synthetic weights of exactly these shapes (``synthetic_detector_state`` /
``synthetic_association_state``).  The distributions are chosen so activations stay
O(1) through 33 residual blocks in f32, bf16 and f16 (small gamma on each block's
last norm) and so that a handful of proposals pass the 0.5 score threshold.
"""
import math

import torch

R101_BLOCKS = (3, 4, 23, 3)


def _bn(sd, name, c, g, lo=0.5, hi=1.5):
    sd[name + ".weight"] = torch.empty(c).uniform_(lo, hi, generator=g)
    sd[name + ".bias"] = torch.randn(c, generator=g) * 0.1
    sd[name + ".running_mean"] = torch.randn(c, generator=g) * 0.1
    sd[name + ".running_var"] = torch.empty(c).uniform_(0.5, 1.5, generator=g)


def _conv(sd, name, cout, cin, k, g, gain=2.0, bias=False, bias_std=0.01):
    std = math.sqrt(gain / (cin * k * k))
    sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * std
    if bias:
        sd[name + ".bias"] = torch.randn(cout, generator=g) * bias_std


def _fc(sd, name, cout, cin, g, gain=2.0, bias_std=0.01):
    sd[name + ".weight"] = torch.randn(cout, cin, generator=g) * math.sqrt(gain / cin)
    sd[name + ".bias"] = torch.randn(cout, generator=g) * bias_std


# Calibrated class-score biases for the benchmark preset (seed 0, R-101, 4K synthetic "static"
# sequence): the random box head has large per-class logit means; these biases centre them and give
# the background a +2.5 margin so that a realistic handful (~8) of detections pass the 0.5 threshold
# (measured on the HIP path, tools/dump_logits.py).  Weights stay a pure function of (seed, preset).
UAV4K_R101_CLS_BIAS = (2.894, -0.240, -2.594, -0.394, -1.122 + 2.5)


def synthetic_detector_state(seed=0, blocks=R101_BLOCKS, num_classes=4, bg_bias=0.0, cls_gain=1.0, cls_bias=None):
    """Seeded random weights with the exact R-FPN Mask R-CNN shapes (f32, CPU)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    _conv(sd, "backbone.bottom_up.stem.conv1", 64, 3, 7, g)
    # inputs are mean-subtracted pixels (|x| ~ 60): bring the stem output to O(1)
    sd["backbone.bottom_up.stem.conv1.weight"] *= 1.0 / 60.0
    _bn(sd, "backbone.bottom_up.stem.conv1.norm", 64, g)
    cin = 64
    for si, nblk in enumerate(blocks):
        mid = 64 * (2 ** si)
        cout = 4 * mid
        for bi in range(nblk):
            p = "backbone.bottom_up.res%d.%d" % (si + 2, bi)
            if bi == 0:
                _conv(sd, p + ".shortcut", cout, cin, 1, g, gain=1.0)
                _bn(sd, p + ".shortcut.norm", cout, g, 0.7, 1.0)
            _conv(sd, p + ".conv1", mid, cin, 1, g)
            _bn(sd, p + ".conv1.norm", mid, g)
            _conv(sd, p + ".conv2", mid, mid, 3, g)
            _bn(sd, p + ".conv2.norm", mid, g)
            _conv(sd, p + ".conv3", cout, mid, 1, g)
            _bn(sd, p + ".conv3.norm", cout, g, 0.1, 0.3)
            cin = cout
    for lvl, c in zip((2, 3, 4, 5), (256, 512, 1024, 2048)):
        _conv(sd, "backbone.fpn_lateral%d" % lvl, 256, c, 1, g, gain=1.0, bias=True)
        _conv(sd, "backbone.fpn_output%d" % lvl, 256, 256, 3, g, gain=1.0, bias=True)
    _conv(sd, "proposal_generator.rpn_head.conv", 256, 256, 3, g, bias=True)
    _conv(sd, "proposal_generator.rpn_head.objectness_logits", 3, 256, 1, g, gain=1.0, bias=True)
    _conv(sd, "proposal_generator.rpn_head.anchor_deltas", 12, 256, 1, g, gain=0.05, bias=True)
    _fc(sd, "roi_heads.box_head.fc1", 1024, 256 * 7 * 7, g)
    _fc(sd, "roi_heads.box_head.fc2", 1024, 1024, g)
    _fc(sd, "roi_heads.box_predictor.cls_score", num_classes + 1, 1024, g, gain=cls_gain)
    sd["roi_heads.box_predictor.cls_score.bias"][num_classes] += bg_bias
    if cls_bias is not None:
        sd["roi_heads.box_predictor.cls_score.bias"] += torch.tensor(cls_bias, dtype=torch.float32)
    _fc(sd, "roi_heads.box_predictor.bbox_pred", num_classes * 4, 1024, g, gain=0.5)
    for i in range(1, 5):
        _conv(sd, "roi_heads.mask_head.mask_fcn%d" % i, 256, 256, 3, g, bias=True)
    sd["roi_heads.mask_head.deconv.weight"] = torch.randn(256, 256, 2, 2, generator=g) * math.sqrt(2.0 / 256)
    sd["roi_heads.mask_head.deconv.bias"] = torch.randn(256, generator=g) * 0.01
    _conv(sd, "roi_heads.mask_head.predictor", num_classes, 256, 1, g, gain=4.0, bias=True)
    return sd


def synthetic_association_state(seed=1, roi_size=10, depth=256, dim=128):
    g = torch.Generator().manual_seed(seed)
    n = depth * roi_size * roi_size
    return {"fc.weight": torch.randn(dim, n, generator=g) * math.sqrt(1.0 / n),
            "fc.bias": torch.randn(dim, generator=g) * 0.01}


def blocks_from_state(sd):
    blocks = []
    for s in (2, 3, 4, 5):
        n = 0
        while ("backbone.bottom_up.res%d.%d.conv1.weight" % (s, n)) in sd:
            n += 1
        blocks.append(n)
    return tuple(blocks)


def load_detector_file(path):
    """Reads a detectron2-style checkpoint: ``.pth`` = dict with key "model"
    (reference: dcnn/scripts/train/finetune_uav.py:273-283) or a bare state_dict.
    Only the tensor-only safe loader is used."""
    obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, dict) and "model" in obj and isinstance(obj["model"], dict):
        obj = obj["model"]
    return {k: torch.as_tensor(v).to(torch.float32) for k, v in obj.items()
            if not k.startswith("proposal_generator.anchor_generator") and k not in ("pixel_mean", "pixel_std")}


def load_association_file(path):
    """Plain state_dict, as rcnn_tracker.py:56 loads it."""
    obj = torch.load(path, map_location="cpu", weights_only=True)
    return {k: torch.as_tensor(v).to(torch.float32) for k, v in obj.items()}
