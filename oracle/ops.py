"""Oracle building blocks (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restatements of the third-party operators the reference's hot path reaches
(SURVEY.md section 8a).  detectron2 0.1.2 / torchvision 0.6.0 are not installed
and their source is not under /root/reference, so each function restates the
published algorithm and cites the reference call site that reaches it.
Parity for these is *unpinned* (no reference test or fixture covers them).

All arithmetic is float32 on CPU, in the operation order of the original CPU
kernels (no FMA contraction: numpy / torch elementwise ops round each step).
"""
import math
import sys

import numpy as np
import torch
import torch.nn.functional as F

SCALE_CLAMP = math.log(1000.0 / 16)  # detectron2 Box2BoxTransform default


# --------------------------------------------------------------------------- anchors
def cell_anchors(size, ratios=(0.5, 1.0, 2.0)):
    """detectron2 DefaultAnchorGenerator.generate_cell_anchors (one size per level,
    reference config dcnn/configs/Base-RCNN-FPN.yaml:9-11).  Python doubles -> f32."""
    out = []
    area = float(size) ** 2.0
    for ar in ratios:
        w = math.sqrt(area / ar)
        h = ar * w
        out.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(out, dtype=torch.float32)


def grid_anchors(h, w, stride, size, ratios=(0.5, 1.0, 2.0)):
    """Anchors of one level in (y, x, a) order, offset 0 (detectron2 0.1.2)."""
    base = cell_anchors(size, ratios)
    sx = torch.arange(0, w * stride, step=stride, dtype=torch.float32)
    sy = torch.arange(0, h * stride, step=stride, dtype=torch.float32)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    xx = xx.reshape(-1)
    yy = yy.reshape(-1)
    shifts = torch.stack((xx, yy, xx, yy), dim=1)
    return (shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4)


# --------------------------------------------------------------------------- box coding
def apply_deltas(deltas, boxes, weights):
    """detectron2 Box2BoxTransform.apply_deltas; deltas [n, k*4], boxes [n, 4]."""
    boxes = boxes.to(deltas.dtype)
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx = deltas[:, 0::4] / wx
    dy = deltas[:, 1::4] / wy
    dw = deltas[:, 2::4] / ww
    dh = deltas[:, 3::4] / wh
    dw = torch.clamp(dw, max=SCALE_CLAMP)
    dh = torch.clamp(dh, max=SCALE_CLAMP)
    pred_ctr_x = dx * widths[:, None] + ctr_x[:, None]
    pred_ctr_y = dy * heights[:, None] + ctr_y[:, None]
    pred_w = torch.exp(dw) * widths[:, None]
    pred_h = torch.exp(dh) * heights[:, None]
    out = torch.zeros_like(deltas)
    out[:, 0::4] = pred_ctr_x - 0.5 * pred_w
    out[:, 1::4] = pred_ctr_y - 0.5 * pred_h
    out[:, 2::4] = pred_ctr_x + 0.5 * pred_w
    out[:, 3::4] = pred_ctr_y + 0.5 * pred_h
    return out


def clip_boxes(boxes, h, w):
    """detectron2 Boxes.clip((h, w)) on an [n, 4] tensor (returns a new tensor)."""
    b = boxes.clone()
    b[:, 0].clamp_(min=0, max=w)
    b[:, 1].clamp_(min=0, max=h)
    b[:, 2].clamp_(min=0, max=w)
    b[:, 3].clamp_(min=0, max=h)
    return b


def nonempty(boxes, threshold=0.0):
    return ((boxes[:, 2] - boxes[:, 0]) > threshold) & ((boxes[:, 3] - boxes[:, 1]) > threshold)


# --------------------------------------------------------------------------- NMS
def nms(boxes, scores, thr):
    """torchvision 0.6 CPU nms kernel restated (no +1, `ovr > thr` suppresses).

    Order: score descending; ties broken by ascending input index (the original
    uses an unstable sort; the stable order is this build's documented choice).
    Returns kept indices in processing order (int64 numpy)."""
    b = np.ascontiguousarray(boxes.detach().cpu().numpy(), dtype=np.float32)
    s = np.ascontiguousarray(scores.detach().cpu().numpy(), dtype=np.float32)
    n = b.shape[0]
    if n == 0:
        return np.zeros((0,), np.int64)
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(-s, kind="stable")
    suppressed = np.zeros(n, bool)
    keep = []
    zero = np.float32(0)
    thr32 = np.float32(thr)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        if rest.size == 0:
            break
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(zero, xx2 - xx1)
        h = np.maximum(zero, yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > thr32]] = True
    return np.asarray(keep, np.int64)


def batched_nms(boxes, scores, idxs, thr):
    """torchvision 0.6 batched_nms: per-category NMS through the coordinate-offset trick
    (offset = idx * (max_coordinate + 1), added in f32 -- the rounding is part of the result)."""
    if boxes.numel() == 0:
        return np.zeros((0,), np.int64)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + 1)
    return nms(boxes + offsets[:, None], scores, thr)


# --------------------------------------------------------------------------- ROIAlign (aligned=True, sampling_ratio=0)
def roi_align_v2(feat, rois, spatial_scale, out_size):
    """detectron2 ROIAlign(aligned=True, sampling_ratio=0) forward, one feature map.

    feat [C, H, W] f32; rois [n, 4] (x1, y1, x2, y2) in input-image pixels.
    Returns [n, C, out, out].  Follows detectron2/layers/csrc/ROIAlign (CPU):
    adaptive grid = ceil(roi / out), samples at bin_start + (i + .5) * bin / grid,
    bilinear with the (-1, size) validity window and the low/high clamp."""
    C, H, W = feat.shape
    n = rois.shape[0]
    out = torch.zeros((n, C, out_size, out_size), dtype=torch.float32)
    r = rois.detach().cpu().numpy().astype(np.float32)
    sc = np.float32(spatial_scale)
    half = np.float32(0.5)
    for k in range(n):
        sw = r[k, 0] * sc - half
        sh = r[k, 1] * sc - half
        ew = r[k, 2] * sc - half
        eh = r[k, 3] * sc - half
        rw = np.float32(ew - sw)
        rh = np.float32(eh - sh)
        bw = np.float32(rw / np.float32(out_size))
        bh = np.float32(rh / np.float32(out_size))
        gh = int(math.ceil(float(rh) / out_size))
        gw = int(math.ceil(float(rw) / out_size))
        cnt = max(gh * gw, 1)
        if gh <= 0 or gw <= 0:
            continue  # empty sample grid: output stays 0 (sum of nothing / 1)
        ph = np.arange(out_size, dtype=np.float32)[:, None]
        iy = np.arange(gh, dtype=np.float32)[None, :]
        ix = np.arange(gw, dtype=np.float32)[None, :]
        ys = (sh + ph * bh + (iy + half) * bh / np.float32(gh)).astype(np.float32).reshape(-1)
        xs = (sw + ph * bw + (ix + half) * bw / np.float32(gw)).astype(np.float32).reshape(-1)

        def prep(v, size):
            valid = ~((v < -1.0) | (v > size))
            v = np.where(v <= 0, np.float32(0), v).astype(np.float32)
            lo = v.astype(np.int32)
            top = lo >= size - 1
            hi = np.where(top, size - 1, lo + 1)
            lo = np.where(top, size - 1, lo)
            v = np.where(top, lo.astype(np.float32), v)
            l = (v - lo.astype(np.float32)).astype(np.float32)
            h = (np.float32(1) - l).astype(np.float32)
            return valid, lo, hi, l, h

        vy, ylo, yhi, ly, hy = prep(ys, H)
        vx, xlo, xhi, lx, hx = prep(xs, W)
        ylo_t = torch.from_numpy(np.clip(ylo, 0, H - 1).astype(np.int64))
        yhi_t = torch.from_numpy(np.clip(yhi, 0, H - 1).astype(np.int64))
        xlo_t = torch.from_numpy(np.clip(xlo, 0, W - 1).astype(np.int64))
        xhi_t = torch.from_numpy(np.clip(xhi, 0, W - 1).astype(np.int64))
        f_lo = feat[:, ylo_t, :]
        f_hi = feat[:, yhi_t, :]
        v1 = f_lo[:, :, xlo_t]
        v2 = f_lo[:, :, xhi_t]
        v3 = f_hi[:, :, xlo_t]
        v4 = f_hi[:, :, xhi_t]
        hy_t = torch.from_numpy(hy)[None, :, None]
        ly_t = torch.from_numpy(ly)[None, :, None]
        hx_t = torch.from_numpy(hx)[None, None, :]
        lx_t = torch.from_numpy(lx)[None, None, :]
        val = (hy_t * hx_t) * v1 + (hy_t * lx_t) * v2 + (ly_t * hx_t) * v3 + (ly_t * lx_t) * v4
        valid = torch.from_numpy(vy)[None, :, None] & torch.from_numpy(vx)[None, None, :]
        val = torch.where(valid, val, torch.zeros((), dtype=torch.float32))
        val = val.view(C, out_size, gh, out_size, gw).sum(dim=(2, 4))
        out[k] = val / np.float32(cnt)
    return out


def assign_levels(boxes, min_level=2, max_level=5, canonical_size=224, canonical_level=4):
    """detectron2 assign_boxes_to_levels (returns level - min_level, int64)."""
    eps = sys.float_info.epsilon
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    sizes = torch.sqrt(area)
    lv = torch.floor(canonical_level + torch.log2(sizes / canonical_size + eps))
    lv = torch.clamp(lv, min=min_level, max=max_level)
    return lv.to(torch.int64) - min_level


def roi_pooler(feats, boxes, out_size, scales=(0.25, 0.125, 0.0625, 0.03125)):
    """detectron2 ROIPooler(ROIAlignV2) over p2..p5; feats: list of [C, H, W]."""
    n = boxes.shape[0]
    C = feats[0].shape[0]
    out = torch.zeros((n, C, out_size, out_size), dtype=torch.float32)
    if n == 0:
        return out
    lv = assign_levels(boxes)
    for l, (f, s) in enumerate(zip(feats, scales)):
        idx = torch.nonzero(lv == l).squeeze(1)
        if idx.numel():
            out[idx] = roi_align_v2(f, boxes[idx], s, out_size)
    return out


# --------------------------------------------------------------------------- torchvision roi_pool
def roi_pool(feat, rois5, out_size, spatial_scale):
    """torchvision 0.6 ops.roi_pool forward (reference call: dcnn/engines/rcnn_tracker.py:182).

    feat [B, C, H, W]; rois5 [n, 5] = (batch, x1, y1, x2, y2).  Quantised bins, max,
    empty bin -> 0."""
    B, C, H, W = feat.shape
    n = rois5.shape[0]
    out = torch.zeros((n, C, out_size, out_size), dtype=torch.float32)
    r = rois5.detach().cpu().numpy().astype(np.float32)
    sc = np.float32(spatial_scale)

    def rnd(v):  # C round(): half away from zero, on the f32 product
        v = float(np.float32(v))
        return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)

    for k in range(n):
        b = int(r[k, 0])
        sw = rnd(r[k, 1] * sc)
        sh = rnd(r[k, 2] * sc)
        ew = rnd(r[k, 3] * sc)
        eh = rnd(r[k, 4] * sc)
        rw = max(ew - sw + 1, 1)
        rh = max(eh - sh + 1, 1)
        bh = np.float32(rh) / np.float32(out_size)
        bw = np.float32(rw) / np.float32(out_size)
        for ph in range(out_size):
            hs = int(math.floor(float(np.float32(ph) * bh)))
            he = int(math.ceil(float(np.float32(ph + 1) * bh)))
            hs = min(max(hs + sh, 0), H)
            he = min(max(he + sh, 0), H)
            for pw in range(out_size):
                ws = int(math.floor(float(np.float32(pw) * bw)))
                we = int(math.ceil(float(np.float32(pw + 1) * bw)))
                ws = min(max(ws + sw, 0), W)
                we = min(max(we + sw, 0), W)
                if he <= hs or we <= ws:
                    continue
                out[k, :, ph, pw] = feat[b, :, hs:he, ws:we].amax(dim=(1, 2))
    return out


# --------------------------------------------------------------------------- mask paste
def paste_window(boxes_row, img_h, img_w):
    """Integer window of detectron2 _do_paste_mask(skip_empty=True) for one box."""
    x0 = int(max(math.floor(float(boxes_row[0])) - 1, 0))
    y0 = int(max(math.floor(float(boxes_row[1])) - 1, 0))
    x1 = int(min(math.ceil(float(boxes_row[2])) + 1, img_w))
    y1 = int(min(math.ceil(float(boxes_row[3])) + 1, img_h))
    return x0, y0, x1, y1


def paste_mask(prob28, box, img_h, img_w, threshold=0.5):
    """detectron2 paste_masks_in_image, CPU branch (one mask at a time, box window only).

    prob28 [M, M] f32 probabilities; box [4] f32 in output-image pixels.
    Returns (window_bool [y1-y0, x1-x0], (x0, y0, x1, y1)).  Pixels outside the window are
    False (grid_sample zero padding < threshold)."""
    x0i, y0i, x1i, y1i = paste_window(box, img_h, img_w)
    if x1i <= x0i or y1i <= y0i:
        return torch.zeros((max(y1i - y0i, 0), max(x1i - x0i, 0)), dtype=torch.bool), (x0i, y0i, x1i, y1i)
    bx0, by0, bx1, by1 = [box[i].reshape(1, 1) for i in range(4)]
    img_y = torch.arange(y0i, y1i, dtype=torch.float32) + 0.5
    img_x = torch.arange(x0i, x1i, dtype=torch.float32) + 0.5
    img_y = (img_y - by0) / (by1 - by0) * 2 - 1
    img_x = (img_x - bx0) / (bx1 - bx0) * 2 - 1
    gx = img_x[:, None, :].expand(1, img_y.size(1), img_x.size(1))
    gy = img_y[:, :, None].expand(1, img_y.size(1), img_x.size(1))
    grid = torch.stack([gx, gy], dim=3)
    img = F.grid_sample(prob28[None, None].to(torch.float32), grid, align_corners=False)
    return (img[0, 0] >= threshold), (x0i, y0i, x1i, y1i)


def dense_mask(window, rect, img_h, img_w):
    x0, y0, x1, y1 = rect
    m = torch.zeros((img_h, img_w), dtype=torch.bool)
    if x1 > x0 and y1 > y0:
        m[y0:y1, x0:x1] = window
    return m


# --------------------------------------------------------------------------- legacy roi_align
def roi_align_legacy(feat, rois5, out_size, spatial_scale, sampling_ratio):
    """torchvision 0.6 ops.roi_align forward (aligned=False), the call of
    dcnn/engines/roi_features_generator.py:111 and dcnn/engines/rcnn_tracker.py:180 (sampling_ratio=4).

    feat [B, C, H, W]; rois5 [n, 5] = (batch, x1, y1, x2, y2) in input pixels.  torchvision/csrc/cpu/ROIAlign_cpu.cpp:
    no half-pixel shift, roi size clamped to >= 1, a fixed sampling_ratio x sampling_ratio grid per bin, bilinear
    samples with the (-1, size) validity window and the low/high clamp, mean over the grid.  parity unpinned:
    torchvision is not installed here, this restates its published algorithm."""
    B, C, H, W = feat.shape
    n = rois5.shape[0]
    out = torch.zeros((n, C, out_size, out_size), dtype=torch.float32)
    r = rois5.detach().cpu().numpy().astype(np.float32)
    sc = np.float32(spatial_scale)
    half = np.float32(0.5)
    g = int(sampling_ratio)
    for k in range(n):
        b = int(r[k, 0])
        sw, sh = r[k, 1] * sc, r[k, 2] * sc
        ew, eh = r[k, 3] * sc, r[k, 4] * sc
        rw = max(np.float32(ew - sw), np.float32(1.0))
        rh = max(np.float32(eh - sh), np.float32(1.0))
        bw = np.float32(rw / np.float32(out_size))
        bh = np.float32(rh / np.float32(out_size))
        ph = np.arange(out_size, dtype=np.float32)[:, None]
        ii = np.arange(g, dtype=np.float32)[None, :]
        ys = (sh + ph * bh + (ii + half) * bh / np.float32(g)).astype(np.float32).reshape(-1)
        xs = (sw + ph * bw + (ii + half) * bw / np.float32(g)).astype(np.float32).reshape(-1)

        def prep(v, size):
            valid = ~((v < -1.0) | (v > size))
            v = np.where(v <= 0, np.float32(0), v).astype(np.float32)
            lo = v.astype(np.int32)
            top = lo >= size - 1
            hi = np.where(top, size - 1, lo + 1)
            lo = np.where(top, size - 1, lo)
            v = np.where(top, lo.astype(np.float32), v)
            l = (v - lo.astype(np.float32)).astype(np.float32)
            h = (np.float32(1) - l).astype(np.float32)
            return valid, np.clip(lo, 0, size - 1), np.clip(hi, 0, size - 1), l, h

        vy, ylo, yhi, ly, hy = prep(ys, H)
        vx, xlo, xhi, lx, hx = prep(xs, W)
        f = feat[b]
        f_lo = f[:, torch.from_numpy(ylo.astype(np.int64)), :]
        f_hi = f[:, torch.from_numpy(yhi.astype(np.int64)), :]
        xl = torch.from_numpy(xlo.astype(np.int64))
        xh = torch.from_numpy(xhi.astype(np.int64))
        hy_t = torch.from_numpy(hy)[None, :, None]
        ly_t = torch.from_numpy(ly)[None, :, None]
        hx_t = torch.from_numpy(hx)[None, None, :]
        lx_t = torch.from_numpy(lx)[None, None, :]
        val = (hy_t * hx_t) * f_lo[:, :, xl] + (hy_t * lx_t) * f_lo[:, :, xh] + (ly_t * hx_t) * f_hi[:, :, xl] + (ly_t * lx_t) * f_hi[:, :, xh]
        valid = torch.from_numpy(vy)[None, :, None] & torch.from_numpy(vx)[None, None, :]
        val = torch.where(valid, val, torch.zeros((), dtype=torch.float32))
        out[k] = val.view(C, out_size, g, out_size, g).sum(dim=(2, 4)) / np.float32(g * g)
    return out
