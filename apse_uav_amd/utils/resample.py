"""Pillow's separable resampling tables, computed on the host.

The reference resizes each frame with ``PIL.Image.resize((1333, 750), BILINEAR)`` through
detectron2's ``ResizeShortestEdge`` (/root/reference/dcnn/engines/track_predictor.py:23-25,48).
Pillow's 8-bit path (src/libImaging/Resample.c: ``precompute_coeffs`` in double,
``normalize_coeffs_8bpc`` to 22-bit fixed point, horizontal pass then vertical pass, each
rounded to u8) is deterministic integer arithmetic, so the HIP kernels reproduce it bit for
bit from these tables (tests/test_resample.py checks the tables against Pillow itself).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resize_shortest_edge(h, w, min_size=800, max_size=1333):
    """detectron2 ResizeShortestEdge.get_transform output size (newh, neww)."""
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


def _bilinear(x):
    if x < 0.0:
        x = -x
    if x < 1.0:
        return 1.0 - x
    return 0.0


def precompute_coeffs(in_size, out_size):
    """Returns (bounds int32 [out,2] = (first, count), coef int32 [out, ksize], ksize)."""
    in0, in1 = 0.0, float(in_size)
    scale = (in1 - in0) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coef = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        k = []
        for x in range(xmax):
            w = _bilinear((x + xmin - center + 0.5) * ss)
            k.append(w)
            ww += w
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            coef[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx, 0] = xmin
        bounds[xx, 1] = xmax
    return bounds, coef, ksize


def resize_reference_numpy(img, out_h, out_w):
    """Host restatement of the two integer passes (used by the CPU tests to validate the tables)."""
    h, w, c = img.shape
    hb, hc, hk = precompute_coeffs(w, out_w)
    vb, vc, vk = precompute_coeffs(h, out_h)
    src = img.astype(np.int64)
    tmp = np.empty((h, out_w, c), np.uint8)
    half = 1 << (PRECISION_BITS - 1)
    for ox in range(out_w):
        x0, n = hb[ox]
        acc = (src[:, x0:x0 + n, :] * hc[ox, :n].astype(np.int64)[None, :, None]).sum(axis=1) + half
        tmp[:, ox, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    t64 = tmp.astype(np.int64)
    out = np.empty((out_h, out_w, c), np.uint8)
    for oy in range(out_h):
        y0, n = vb[oy]
        acc = (t64[y0:y0 + n] * vc[oy, :n].astype(np.int64)[:, None, None]).sum(axis=0) + half
        out[oy] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out
