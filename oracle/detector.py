"""Oracle detector: Mask R-CNN R-101-FPN inference on CPU (TEST INFRASTRUCTURE).

Restates what the reference reaches through ``TrackRCNN.inference``
(/root/reference/dcnn/networks/track_rcnn.py:16-58), i.e. detectron2 0.1.2's
GeneralizedRCNN with the settings of dcnn/configs/Base-RCNN-FPN.yaml,
dcnn/configs/mask_rcnn_R_101_FPN_3x.yaml and the overrides of
dcnn/scripts/tests/visualize_uav.py:43-53.  detectron2 is not installed and its
source is not under /root/reference: **parity unpinned** (SURVEY.md 8c).

Stage-addressable: every method returns plain tensors so a HIP stage can be
teacher-forced with the oracle's inputs and diffed on its outputs.  Weights come
in as a ``state_dict`` with detectron2's key names.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ops

DEFAULT_CFG = dict(
    depth_blocks=(3, 4, 23, 3),      # R-101 (mask_rcnn_R_101_FPN_3x.yaml:5-6)
    min_size=800, max_size=1333,     # detectron2 INPUT.MIN/MAX_SIZE_TEST defaults
    size_divisibility=32,
    pixel_mean=(103.530, 116.280, 123.675), pixel_std=(1.0, 1.0, 1.0),
    anchor_sizes=(32, 64, 128, 256, 512), anchor_ratios=(0.5, 1.0, 2.0),
    strides=(4, 8, 16, 32, 64),
    rpn_pre_topk=1000, rpn_post_topk=1000, rpn_nms=0.7,
    num_classes=4, score_thresh=0.5, box_nms=0.5, dets_per_image=100,
    box_pool=7, mask_pool=14, mask_thresh=0.5,
    bbox_weights=(10.0, 10.0, 5.0, 5.0),
    # bf16=True emulates the product's "bf16 matrix cores, f32 storage" mode: FrozenBN folded into the
    # filters in f32, filters and layer inputs rounded to bf16 (nearest-even), f32 accumulation and f32
    # outputs; the narrow decision heads (RPN logits/deltas, box predictor, mask logits) stay f32.
    bf16=False,
    # storage16=True (with bf16): every activation the bulk GEMMs produce is also ROUNDED to the 16-bit type
    # where the product stores it (after bias / residual / ReLU), like cfg.APSE.STORAGE16.
    storage16=False,
)

F32_LAYERS = ("proposal_generator.rpn_head.objectness_logits", "proposal_generator.rpn_head.anchor_deltas",
              "roi_heads.mask_head.predictor")


_R16_DTYPE = [torch.bfloat16]


def _r16(t):
    return t.to(_R16_DTYPE[0]).to(torch.float32)


def resize_shape(h, w, min_size=800, max_size=1333):
    """detectron2 ResizeShortestEdge.get_transform output shape (reference:
    dcnn/engines/track_predictor.py:23-25,48)."""
    scale = min_size * 1.0 / min(h, w)
    if h < w:
        newh, neww = min_size, scale * w
    else:
        newh, neww = scale * h, min_size
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh = newh * scale
        neww = neww * scale
    return int(newh + 0.5), int(neww + 0.5)


class DetectorOracle:
    def __init__(self, state_dict, cfg=None):
        self.cfg = dict(DEFAULT_CFG)
        if cfg:
            self.cfg.update(cfg)
        self.sd = {k: v.detach().to(torch.float32).cpu() for k, v in state_dict.items()}
        if self.cfg["bf16"] == "f16":            # same emulation with IEEE half operands (BASELINE config 5)
            _R16_DTYPE[0] = torch.float16
        elif self.cfg["bf16"]:
            _R16_DTYPE[0] = torch.bfloat16

    # ------------------------------------------------------------------ helpers
    def _st(self, t):
        """Storage rounding of a tensor the product keeps in HBM as bf16 / f16."""
        return _r16(t) if (self.cfg["bf16"] and self.cfg["storage16"]) else t

    def _conv(self, x, name, stride=1, padding=0, relu=False, store=True):
        w = self.sd[name + ".weight"]
        b = self.sd.get(name + ".bias")
        if self.cfg["bf16"] and name not in F32_LAYERS:
            if (name + ".norm.weight") in self.sd:
                g = self.sd[name + ".norm.weight"]
                scale = g * (1.0 / torch.sqrt(self.sd[name + ".norm.running_var"] + 1e-5))
                b = self.sd[name + ".norm.bias"] - self.sd[name + ".norm.running_mean"] * scale
                w = w * scale.view(-1, 1, 1, 1)
            y = F.conv2d(_r16(x), _r16(w), b, stride=stride, padding=padding)
            y = F.relu(y) if relu else y
            return self._st(y) if store else y
        y = F.conv2d(x, w, b, stride=stride, padding=padding)
        if (name + ".norm.weight") in self.sd:
            # FrozenBatchNorm2d: x * scale + bias with scale = w * rsqrt(var + eps)
            eps = 1e-5
            g = self.sd[name + ".norm.weight"]
            be = self.sd[name + ".norm.bias"]
            mu = self.sd[name + ".norm.running_mean"]
            var = self.sd[name + ".norm.running_var"]
            scale = g * (var + eps).rsqrt()
            bias = be - mu * scale
            y = y * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
        return F.relu(y) if relu else y

    # ------------------------------------------------------------------ stages
    def preprocess(self, image_chw):
        """GeneralizedRCNN.preprocess_image: normalise, zero-pad to /32 (track_rcnn.py:35)."""
        c = self.cfg
        mean = torch.tensor(c["pixel_mean"], dtype=torch.float32).view(-1, 1, 1)
        std = torch.tensor(c["pixel_std"], dtype=torch.float32).view(-1, 1, 1)
        x = (image_chw.to(torch.float32) - mean) / std
        h, w = x.shape[-2:]
        d = c["size_divisibility"]
        ph = int(math.ceil(h / d) * d)
        pw = int(math.ceil(w / d) * d)
        return F.pad(x, [0, pw - w, 0, ph - h], value=0.0).unsqueeze(0)

    def stem(self, x):
        x = self._conv(x, "backbone.bottom_up.stem.conv1", stride=2, padding=3, relu=True)
        return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)

    def bottleneck(self, x, prefix, stride):
        out = self._conv(x, prefix + ".conv1", stride=stride, relu=True)       # STRIDE_IN_1X1
        out = self._conv(out, prefix + ".conv2", stride=1, padding=1, relu=True)
        out = self._conv(out, prefix + ".conv3", store=False)     # residual add + ReLU happen before the store
        if (prefix + ".shortcut.weight") in self.sd:
            sc = self._conv(x, prefix + ".shortcut", stride=stride)
        else:
            sc = x
        return self._st(F.relu(out + sc))

    def bottom_up(self, x):
        feats = {}
        x = self.stem(x)
        feats["stem"] = x
        for si, nblk in enumerate(self.cfg["depth_blocks"]):
            stage = "res%d" % (si + 2)
            for bi in range(nblk):
                stride = 2 if (bi == 0 and si > 0) else 1
                x = self.bottleneck(x, "backbone.bottom_up.%s.%d" % (stage, bi), stride)
            feats[stage] = x
        return feats

    def fpn(self, feats):
        """detectron2 FPN.forward (fuse sum, nearest x2) + LastLevelMaxPool."""
        out = {}
        prev = self._conv(feats["res5"], "backbone.fpn_lateral5")
        out["p5"] = self._conv(prev, "backbone.fpn_output5", padding=1)
        for lvl in (4, 3, 2):
            top = F.interpolate(prev, scale_factor=2, mode="nearest")
            lat = self._conv(feats["res%d" % lvl], "backbone.fpn_lateral%d" % lvl, store=False)
            prev = self._st(lat + top)
            out["p%d" % lvl] = self._conv(prev, "backbone.fpn_output%d" % lvl, padding=1)
        out["p6"] = F.max_pool2d(out["p5"], kernel_size=1, stride=2, padding=0)
        return out

    def backbone(self, x):
        feats = self.bottom_up(x)
        feats.update(self.fpn(feats))
        return feats

    def rpn_head(self, feats):
        logits, deltas = [], []
        for k in ("p2", "p3", "p4", "p5", "p6"):
            t = self._conv(feats[k], "proposal_generator.rpn_head.conv", padding=1, relu=True)
            logits.append(self._conv(t, "proposal_generator.rpn_head.objectness_logits"))
            deltas.append(self._conv(t, "proposal_generator.rpn_head.anchor_deltas"))
        return logits, deltas

    def rpn_select(self, logits, deltas, image_size, levels=None):
        """detectron2 RPNOutputs.predict_* + find_top_rpn_proposals (batch 1).

        Returns dict(boxes [P,4], logits [P], plus per-stage intermediates)."""
        c = self.cfg
        h_img, w_img = image_size
        top_scores, top_boxes, top_idx, lvls = [], [], [], []
        for li, (lg, dl) in enumerate(zip(logits, deltas)):
            if levels is not None and li not in levels:
                continue                       # SelectiveRPN: only some levels reach find_top_rpn_proposals
            _, A, H, W = lg.shape
            lg_f = lg.permute(0, 2, 3, 1).reshape(-1)                       # (y, x, a)
            dl_f = dl.view(1, A, 4, H, W).permute(0, 3, 4, 1, 2).reshape(-1, 4)
            anchors = ops.grid_anchors(H, W, c["strides"][li], c["anchor_sizes"][li], c["anchor_ratios"])
            k = min(c["rpn_pre_topk"], lg_f.numel())
            # detectron2 0.1.2 sorts the whole level and takes the head; ties -> ascending index
            order = torch.sort(lg_f, descending=True, stable=True).indices[:k]
            props = ops.apply_deltas(dl_f[order], anchors[order], (1.0, 1.0, 1.0, 1.0))
            top_scores.append(lg_f[order])
            top_boxes.append(props)
            top_idx.append(order)
            lvls.append(torch.full((k,), len(lvls), dtype=torch.int64))      # level ids count the levels present
        scores = torch.cat(top_scores)
        boxes = torch.cat(top_boxes)
        lvl = torch.cat(lvls)
        boxes = ops.clip_boxes(boxes, h_img, w_img)
        keep = ops.nonempty(boxes, 0.0)
        boxes_k, scores_k, lvl_k = boxes[keep], scores[keep], lvl[keep]
        kept = ops.batched_nms(boxes_k, scores_k, lvl_k, c["rpn_nms"])
        kept = kept[: c["rpn_post_topk"]]
        kept_t = torch.from_numpy(kept)
        return dict(boxes=boxes_k[kept_t], logits=scores_k[kept_t],
                    topk_idx=top_idx, topk_scores=top_scores, decoded=boxes, valid=keep,
                    level=lvl_k[kept_t])

    def box_features(self, feats, proposals):
        pooled = ops.roi_pooler([feats[k][0] for k in ("p2", "p3", "p4", "p5")], proposals, self.cfg["box_pool"])
        pooled = self._st(pooled)
        x = pooled.flatten(1)
        q = _r16 if self.cfg["bf16"] else (lambda t: t)
        x = self._st(F.relu(F.linear(q(x), q(self.sd["roi_heads.box_head.fc1.weight"]), self.sd["roi_heads.box_head.fc1.bias"])))
        x = self._st(F.relu(F.linear(q(x), q(self.sd["roi_heads.box_head.fc2.weight"]), self.sd["roi_heads.box_head.fc2.bias"])))
        cls = F.linear(x, self.sd["roi_heads.box_predictor.cls_score.weight"], self.sd["roi_heads.box_predictor.cls_score.bias"])
        reg = F.linear(x, self.sd["roi_heads.box_predictor.bbox_pred.weight"], self.sd["roi_heads.box_predictor.bbox_pred.bias"])
        return dict(pooled=pooled, cls_logits=cls, deltas=reg)

    def box_inference(self, cls_logits, deltas, proposals, image_size):
        """detectron2 FastRCNNOutputs.inference -> fast_rcnn_inference_single_image."""
        c = self.cfg
        K = c["num_classes"]
        n = proposals.shape[0]
        if n == 0:
            return dict(boxes=torch.zeros((0, 4)), scores=torch.zeros((0,)), classes=torch.zeros((0,), dtype=torch.int64),
                        roi_index=torch.zeros((0,), dtype=torch.int64))
        probs = F.softmax(cls_logits, dim=-1)
        boxes = ops.apply_deltas(deltas.reshape(n * K, 4), proposals.unsqueeze(1).expand(n, K, 4).reshape(-1, 4), c["bbox_weights"])
        boxes = ops.clip_boxes(boxes.reshape(-1, 4), image_size[0], image_size[1]).view(n, K, 4)
        scores = probs[:, :-1]                                   # background is the last column
        mask = scores > c["score_thresh"]
        inds = mask.nonzero()                                    # (roi, class), row-major
        cand_boxes = boxes[mask]
        cand_scores = scores[mask]
        kept = ops.batched_nms(cand_boxes, cand_scores, inds[:, 1], c["box_nms"])
        kept = kept[: c["dets_per_image"]]
        kept_t = torch.from_numpy(kept)
        return dict(boxes=cand_boxes[kept_t], scores=cand_scores[kept_t], classes=inds[kept_t, 1],
                    roi_index=inds[kept_t, 0], probs=probs, all_boxes=boxes)

    def mask_head(self, feats, boxes, classes):
        """Mask branch: ROIAlign 14 -> 4x(conv3x3+ReLU) -> deconv2x2+ReLU -> 1x1 -> sigmoid -> class channel."""
        n = boxes.shape[0]
        m = self.cfg["mask_pool"]
        if n == 0:
            return dict(pooled=torch.zeros((0, 256, m, m)), logits=torch.zeros((0, self.cfg["num_classes"], 2 * m, 2 * m)),
                        probs=torch.zeros((0, 2 * m, 2 * m)))
        x = self._st(ops.roi_pooler([feats[k][0] for k in ("p2", "p3", "p4", "p5")], boxes, m))
        pooled = x
        for i in range(1, 5):
            x = self._conv(x, "roi_heads.mask_head.mask_fcn%d" % i, padding=1, relu=True)
        q = _r16 if self.cfg["bf16"] else (lambda t: t)
        x = self._st(F.relu(F.conv_transpose2d(q(x), q(self.sd["roi_heads.mask_head.deconv.weight"]),
                                               self.sd["roi_heads.mask_head.deconv.bias"], stride=2)))
        logits = self._conv(x, "roi_heads.mask_head.predictor")
        probs = logits.sigmoid()[torch.arange(n), classes]
        return dict(pooled=pooled, logits=logits, probs=probs)

    def postprocess(self, boxes, scores, classes, mask_probs, image_size, out_h, out_w):
        """detectron2 detector_postprocess: rescale, clip, drop empty, paste masks (box windows)."""
        sx = out_w / image_size[1]
        sy = out_h / image_size[0]
        b = boxes.clone()
        b[:, 0::2] *= sx
        b[:, 1::2] *= sy
        b = ops.clip_boxes(b, out_h, out_w)
        keep = ops.nonempty(b)
        b, scores, classes, mask_probs = b[keep], scores[keep], classes[keep], mask_probs[keep]
        windows, rects = [], []
        for i in range(b.shape[0]):
            w, r = ops.paste_mask(mask_probs[i], b[i], out_h, out_w, self.cfg["mask_thresh"])
            windows.append(w)
            rects.append(r)
        return dict(boxes=b, scores=scores, classes=classes, mask_windows=windows, mask_rects=rects, keep=keep)

    # ------------------------------------------------------------------ whole path
    def inference(self, image_chw, out_h, out_w, given_boxes=None, given_classes=None, rpn_levels=None):
        """TrackRCNN.inference on one image (track_rcnn.py:16-58).  image_chw is the
        *resized* f32 CHW BGR image; out_h/out_w the original frame size.  With
        ``given_boxes`` (resized-image coordinates) the box branch is skipped
        (roi_heads.forward_with_given_boxes, track_rcnn.py:52-54)."""
        image_size = tuple(image_chw.shape[-2:])
        x = self.preprocess(image_chw)
        feats = self.backbone(x)
        if given_boxes is None:
            lg, dl = self.rpn_head(feats)
            prop = self.rpn_select(lg, dl, image_size, rpn_levels)
            bf = self.box_features(feats, prop["boxes"])
            det = self.box_inference(bf["cls_logits"], bf["deltas"], prop["boxes"], image_size)
            boxes, scores, classes = det["boxes"], det["scores"], det["classes"]
            box_det = det
        else:
            box_det = None
            boxes = given_boxes.to(torch.float32)
            classes = given_classes.to(torch.int64)
            scores = torch.ones((boxes.shape[0],), dtype=torch.float32)
            prop = None
        mh = self.mask_head(feats, boxes, classes)
        post = self.postprocess(boxes, scores, classes, mh["probs"], image_size, out_h, out_w)
        post["features"] = feats
        post["proposals"] = prop
        post["box_det"] = box_det              # pre-NMS candidates (probs, all_boxes) + kept (roi, class): threshold-margin analysis
        post["image_size"] = image_size
        return post
