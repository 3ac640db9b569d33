"""Frame pre-processing -- counterpart of ``preprocess_img`` in
/root/reference/dcnn/scripts/tests/visualize_uav.py:56-71 (undistort with the camera parameters of
``data/cam_params.json`` + gamma 2 on the Lab L channel).  The reference driver has the call commented
out (:191) because its sequences were pre-processed offline; here it is an optional GPU stage in front of
the resize (``cfg.APSE.FUSED_PREPROC``), one HIP kernel per batch of frames.
"""
import ctypes as C
import json

import numpy as np
import torch

from .. import _lib


def gamma_lut(gamma=2.0):
    lut = np.empty(256, np.uint8)
    for i in range(256):
        lut[i] = np.uint8(np.clip(pow(i / 255.0, gamma) * 255.0, 0, 255))       # visualize_uav.py:65-67
    return lut


class FramePreprocessor:
    def __init__(self, cam_params, gamma=2.0, undistort=True, gamma_correct=True):
        if isinstance(cam_params, str):
            with open(cam_params) as f:
                cam_params = json.load(f)
        self.mtx = np.ascontiguousarray(np.asarray(cam_params["mtx"], np.float64).reshape(9))
        self.dist = np.ascontiguousarray(np.asarray(cam_params["dist"], np.float64).reshape(-1))
        self.lut_host = gamma_lut(gamma)
        self.undistort, self.gamma_correct = bool(undistort), bool(gamma_correct)

    def __call__(self, frames):
        """frames: uint8 CUDA tensor [B, H, W, 3] (BGR) -> new tensor of the same shape."""
        if not frames.is_cuda:
            raise _lib.ApseError("FramePreprocessor needs CUDA frames (no CPU fallback)")
        frames = frames.contiguous()
        out = torch.empty_like(frames)
        B, H, W, _ = frames.shape
        rc = _lib.load().apse_undistort_gamma(
            _lib.ptr(frames), _lib.ptr(out), B, H, W, self.mtx.ctypes.data_as(C.POINTER(C.c_double)),
            self.dist.ctypes.data_as(C.POINTER(C.c_double)), int(self.dist.size), _lib.ptr(self.lut_host), int(self.undistort),
            int(self.gamma_correct), _lib.stream_ptr())
        _lib.check(rc, None, "apse_undistort_gamma")
        return out

    def preprocess_img(self, frame):
        """numpy HxWx3 uint8 -> numpy, like the reference function."""
        d = torch.from_numpy(np.ascontiguousarray(frame)).cuda()[None]
        return self(d)[0].cpu().numpy()
