"""MOTS result writers -- counterpart of /root/reference/dcnn/utils/mots_evaluation.py:25-117.

Same functions and semantics on the build's ``ObjectInstances`` whose ``pred_masks`` are box-local
:class:`WindowMask` objects instead of dense 2160x3840 tensors: every operation below touches only the paste
windows (and, for the id map, the one u16 output image), never N dense frames.

 * ``file_lines_from_instances``  (:25-55)  MOTS txt lines ``frame id class h w rle`` (COCO RLE string, utils/rle.py)
 * ``result_image_from_objects``  (:58-77)  u16 id map, ``class*1000 + id``, later objects overwrite earlier ones
 * ``parse_mots_seqmap``          (:80-94)
 * ``crop_overlapping_masks``     (:97-117) the lower-score mask of every overlapping pair loses the overlap
Class relabelling as in the reference: 0 (person) -> 2, 2 (car) -> 1, every other class is skipped (:31-37).
"""
import numpy as np
import torch

from ..structures.window_mask import WindowMask
from . import rle


def _mots_class(ob_class):
    if ob_class == 0:
        return 2
    if ob_class == 2:
        return 1
    return None


def _window_np(m):
    """bool ndarray of the paste window of a WindowMask (or of a dense mask: whole frame)."""
    if isinstance(m, WindowMask):
        return m.window().cpu().numpy(), m.rect
    d = torch.as_tensor(m).cpu().numpy().astype(bool)
    return d, (0, 0, d.shape[1], d.shape[0])


def _dense_np(m, image_size):
    win, (x0, y0, x1, y1) = _window_np(m)
    out = np.zeros(tuple(image_size), dtype=bool)
    if win.size:
        out[y0:y1, x0:x1] = win
    return out


def file_lines_from_instances(object_instances, frame_num, image_size):
    out_string = ""
    for obj_idx in range(len(object_instances)):
        ob_class = _mots_class(int(object_instances.pred_classes[obj_idx]))
        if ob_class is None:
            continue
        ob_id = object_instances.ids[obj_idx]
        ob_rle = rle.encode(_dense_np(object_instances.pred_masks[obj_idx], image_size))
        object_id = ob_class * 1000 + ob_id
        out_string += "%d %d %d %d %d %s\n" % (frame_num, object_id, ob_class, image_size[0], image_size[1],
                                               ob_rle["counts"].decode("ascii"))
    return out_string


def result_image_from_objects(object_instances, image_size):
    img = np.zeros(tuple(image_size), dtype=np.uint16)
    for obj_idx in range(len(object_instances)):
        ob_class = _mots_class(int(object_instances.pred_classes[obj_idx]))
        if ob_class is None:
            continue
        object_id = ob_class * 1000 + object_instances.ids[obj_idx]
        win, (x0, y0, x1, y1) = _window_np(object_instances.pred_masks[obj_idx])
        if win.size:
            img[y0:y1, x0:x1][win] = object_id
    return img.astype(np.uint16)


def parse_mots_seqmap(path):
    names, lengths = [], []
    with open(path, "r") as f:
        for line in f.readlines():
            parts = line.split(" ")
            names.append(parts[0].strip())
            lengths.append(int(parts[3].strip()) + 1)        # seqmaps give the index of the last frame (0-based)
    return names, lengths


def _repack(win, rect, frame_size, device):
    """bool window -> WindowMask (64 px per int64 word from column (x0 >> 6) << 6) with mass and 1-based floor centroid
    (mask_utils.get_mask_centroid rules) recomputed from the window."""
    x0, y0, x1, y1 = rect
    rows = y1 - y0
    wx0 = (x0 >> 6) << 6
    words = ((x1 + 63) >> 6) - (x0 >> 6)
    px = np.zeros((rows, words * 64), dtype=bool)
    px[:, x0 - wx0:x0 - wx0 + (x1 - x0)] = win
    weights = (np.uint64(1) << np.arange(64, dtype=np.uint64))
    bits = (px.reshape(rows, words, 64).astype(np.uint64) * weights).sum(axis=2, dtype=np.uint64).view(np.int64)
    mass = int(win.sum())
    if mass:
        ys, xs = np.nonzero(win)
        cen = (int((xs.astype(np.int64) + x0 + 1).sum() // mass), int((ys.astype(np.int64) + y0 + 1).sum() // mass))
    else:
        cen = (-1, -1)
    return WindowMask(torch.from_numpy(bits.copy()).to(device), rect, frame_size, cen, mass)


def crop_overlapping_masks(object_instances):
    """In place, in the reference's pair order (i < j, later pairs see earlier crops)."""
    n = len(object_instances)
    if n == 0:
        return
    masks = object_instances.pred_masks
    scores = object_instances.scores
    wins = [None] * n

    def get(k):
        if wins[k] is None:
            w, r = _window_np(masks[k])
            wins[k] = [w.copy(), tuple(r), False]
        return wins[k]

    for i in range(n):
        for j in range(i + 1, n):
            wi, wj = get(i), get(j)
            (ax0, ay0, ax1, ay1), (bx0, by0, bx1, by1) = wi[1], wj[1]
            x0, y0, x1, y1 = max(ax0, bx0), max(ay0, by0), min(ax1, bx1), min(ay1, by1)
            if x1 <= x0 or y1 <= y0:
                continue
            si = wi[0][y0 - ay0:y1 - ay0, x0 - ax0:x1 - ax0]
            sj = wj[0][y0 - by0:y1 - by0, x0 - bx0:x1 - bx0]
            inter = si & sj
            if not inter.any():
                continue
            if float(scores[i]) > float(scores[j]):
                sj &= ~inter
                wj[2] = True
            else:
                si &= ~inter
                wi[2] = True
    for k in range(n):
        if wins[k] is not None and wins[k][2]:
            m = masks[k]
            if isinstance(m, WindowMask):
                masks[k] = _repack(wins[k][0], wins[k][1], m.frame_size, m.device)
            else:
                masks[k] = torch.as_tensor(wins[k][0])
