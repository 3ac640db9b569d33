"""Mask geometry helpers -- counterpart of /root/reference/dcnn/utils/mask_utils.py:6-38.

``get_mask_centroid(mask)`` -> (x, y) floats, 1-based, floor of the masked coordinate mean;
``compute_closest_point(mask, the_point)`` -> (x, y) of the first row-major mask pixel with the
smallest f32 squared distance.  Both run on the GPU (HIP kernels behind the C ABI); ``mask`` is
a :class:`WindowMask` (centroid already computed with the paste) or a dense bool CUDA tensor.
The reference's IoU helpers (:41-77) are dead code there (undefined ``self``) and are not provided.
"""
import ctypes as C

import torch

from .. import _lib
from ..structures.window_mask import WindowMask


def _dense_u8(mask):
    if isinstance(mask, WindowMask):
        mask = mask.dense()
    if not mask.is_cuda:
        raise _lib.ApseError("mask_utils needs the mask on the GPU (no CPU fallback)")
    return mask.to(torch.uint8).contiguous()


def get_mask_centroid(mask):
    if isinstance(mask, WindowMask):
        return mask.centroid
    m = _dense_u8(mask)
    out = (C.c_int * 3)()
    _lib.check(_lib.load().apse_mask_centroid_dense(_lib.ptr(m), m.shape[0], m.shape[1], C.byref(out), _lib.stream_ptr()),
               None, "apse_mask_centroid_dense")
    if out[2] == 0:
        return (float("nan"), float("nan"))
    return (float(out[0]), float(out[1]))


def compute_closest_point(mask, the_point):
    m = _dense_u8(mask)
    out = (C.c_int * 2)()
    _lib.check(_lib.load().apse_mask_closest_dense(_lib.ptr(m), m.shape[0], m.shape[1], float(the_point[0]),
                                                   float(the_point[1]), C.byref(out), _lib.stream_ptr()),
               None, "apse_mask_closest_dense")
    if out[0] < 0:
        raise RuntimeError("compute_closest_point: empty mask")
    return (float(out[0]), float(out[1]))
