"""ctypes binding of libapse_hip.so (include/apse_hip.h).

PyTorch-ROCm is used only for device memory and streams: every call passes raw
``data_ptr()`` values and ``torch.cuda.current_stream().cuda_stream``.  There is
no CPU fallback: if the library is missing or no GPU is visible the product path
raises (the oracle under ``oracle/`` is test infrastructure and is never imported
from here).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# APSE_HIP_LIB: another build of the same sources (A/B measurements of a compile-time switch, tools/gpu_ab.sh); the
# default is the in-tree library
LIB_PATH = os.environ.get("APSE_HIP_LIB") or os.path.join(_HERE, "libapse_hip.so")

APSE_OK = 0


class ApseError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int), ("device", C.c_int), ("max_batch", C.c_int),
        ("frame_h", C.c_int), ("frame_w", C.c_int), ("image_h", C.c_int), ("image_w", C.c_int),
        ("blocks", C.c_int * 4), ("num_classes", C.c_int),
        ("score_thresh", C.c_float), ("box_nms", C.c_float), ("rpn_nms", C.c_float), ("mask_thresh", C.c_float),
        ("rpn_pre_topk", C.c_int), ("rpn_post_topk", C.c_int), ("dets_per_image", C.c_int),
        ("pixel_mean", C.c_float * 3), ("assoc_roi", C.c_int), ("embed_dim", C.c_int), ("assoc_scale", C.c_float),
        ("compute_dtype", C.c_int), ("storage16", C.c_int),
    ]


class ResultsLayout(C.Structure):
    _fields_ = [("bytes", C.c_size_t), ("n_max", C.c_int), ("dets_per_image", C.c_int), ("embed_dim", C.c_int),
                ("max_batch", C.c_int)] + [(k, C.c_size_t) for k in (
                    "total", "offset", "prop_count", "img", "cls", "roi", "score", "box_resized", "box", "valid", "rect",
                    "mass", "centroid", "closest", "embedding")]


class ConvDesc(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("B", "H", "W", "Cin", "Cout", "KH", "KW", "stride", "pad", "relu", "res_mode",
                                       "cfg", "splitk", "prec", "fuse_reduce", "x_st", "res_st", "y_st")]


_lib = None


def load():
    """Loads the shared library (raises ApseError when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ApseError("libapse_hip.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "or `make -C apse_uav_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and the library links the system one.  Whichever is
    # loaded first serves both, and with the system copy first the library found no device on the GPU box once torch had come up
    # on its own copy afterwards (build() followed by smoke() in one process).  torch first, always.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    sig = {
        "apse_create": ([C.POINTER(Config), C.POINTER(vp)], i),
        "apse_destroy": ([vp], None),
        "apse_last_error": ([vp], C.c_char_p),
        "apse_version": ([], C.c_char_p),
        "apse_set_weight": ([vp, C.c_char_p, vp, C.POINTER(C.c_int64), i], i),
        "apse_finalize_weights": ([vp], i),
        "apse_set_resize_tables": ([vp, vp, vp, i, vp, vp, i], i),
        "apse_set_camera": ([vp, C.POINTER(C.c_double), C.POINTER(C.c_double), i, vp, i, i], i),
        "apse_preprocess_frames": ([vp, vp, i, vp], i),
        "apse_preprocess_images": ([vp, vp, i, vp], i),
        "apse_backbone": ([vp, i, vp], i),
        "apse_rpn": ([vp, i, vp], i),
        "apse_rpn_levels": ([vp, i, i, vp], i),
        "apse_box_head": ([vp, i, vp], i),
        "apse_set_detections": ([vp, vp, vp, vp, vp, i, vp], i),
        "apse_mask_tail": ([vp, i, vp], i),
        "apse_embed": ([vp, i, vp], i),
        "apse_forward": ([vp, i, vp], i),
        "apse_results_describe": ([vp, C.POINTER(ResultsLayout)], i),
        "apse_read_results": ([vp, vp, sz, vp], i),
        "apse_read_results_begin": ([vp, vp, sz, vp], i),
        "apse_read_results_end": ([vp, vp], i),
        "apse_copy_mask_window": ([vp, i, i, i, i, i, vp, vp], i),
        "apse_copy_mask_windows": ([vp, i, vp, vp, vp, vp, vp], i),
        "apse_host_copy": ([vp, vp, sz, i], i),
        "apse_feature_shape": ([vp, C.c_char_p, C.POINTER(C.c_int * 3)], i),
        "apse_export_feature": ([vp, C.c_char_p, vp, i, vp], i),
        "apse_debug_tensor": ([vp, C.c_char_p, vp, sz, C.POINTER(sz), vp], i),
        "apse_flops": ([vp, i, C.c_double, C.c_double], C.c_double),
        "apse_profile": ([vp, i], i),
        "apse_profile_read": ([vp, C.POINTER(C.c_double * 42), i], i),
        "apse_conv_packed_elems": ([C.POINTER(ConvDesc)], sz),
        "apse_conv_pack_weight": ([C.POINTER(ConvDesc), vp, i, vp, vp], i),
        "apse_conv2d": ([C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, sz, vp], i),
        "apse_maxpool3x3s2": ([vp, vp, i, i, i, i, vp], i),
        "apse_maxpool3x3s2_typed": ([vp, vp, i, i, i, i, i, vp], i),
        "apse_roi_align": ([C.POINTER(vp), C.POINTER(i), C.POINTER(i), vp, i, i, i, vp, vp], i),
        "apse_roi_align_typed": ([C.POINTER(vp), C.POINTER(i), C.POINTER(i), vp, i, i, i, i, vp, vp], i),
        "apse_roi_pool": ([vp, i, i, vp, vp, i, i, f, vp, vp], i),
        "apse_roi_features": ([vp, i, vp, vp, i, i, vp, vp], i),
        "apse_nms_rank": ([vp, vp, vp, i, i, i, i, f, i, vp, vp, vp, vp, vp], i),
        "apse_mask_centroid_dense": ([vp, i, i, C.POINTER(C.c_int * 3), vp], i),
        "apse_mask_closest_dense": ([vp, i, i, f, f, C.POINTER(C.c_int * 2), vp], i),
        "apse_l2_normalize": ([vp, vp, i, i, vp], i),
        "apse_sqdist": ([vp, vp, i, i, i, vp, vp], i),
        "apse_undistort_gamma": ([vp, vp, i, i, i, C.POINTER(C.c_double), C.POINTER(C.c_double), i, vp, i, i, vp], i),
        "apse_lab_tables_host": ([vp, vp, C.c_size_t], C.c_size_t),
        "apse_replay_create": ([i, i, f, i], vp),
        "apse_replay_destroy": ([vp], None),
        "apse_replay_step": ([vp, i, i, vp, vp, vp, C.c_char_p, i, vp], i),
        "apse_replay_packed": ([vp, vp, C.c_longlong, i, i, i, vp, C.c_longlong], C.c_longlong),
        "apse_replay_max_id": ([vp], i),
        "apse_replay_next_id": ([vp], i),
        "apse_resize_normalize": ([vp, vp, vp, vp, vp, vp, i, vp, vp, i, i, i, i, i, i, i, i, C.POINTER(f * 3), vp], i),
    }
    for name, (args, ret) in sig.items():
        fn = getattr(lib, name)            # AttributeError here = ABI drift between header and library
        fn.argtypes = args
        fn.restype = ret
    _lib = lib
    return lib


EXPORTS = ["apse_create", "apse_destroy", "apse_last_error", "apse_version", "apse_set_weight", "apse_finalize_weights",
           "apse_set_resize_tables", "apse_set_camera", "apse_preprocess_frames", "apse_preprocess_images", "apse_backbone", "apse_rpn", "apse_rpn_levels",
           "apse_box_head", "apse_set_detections", "apse_mask_tail", "apse_embed", "apse_forward", "apse_results_describe",
           "apse_read_results", "apse_read_results_begin", "apse_read_results_end", "apse_copy_mask_window", "apse_copy_mask_windows", "apse_host_copy", "apse_feature_shape", "apse_export_feature", "apse_debug_tensor",
           "apse_flops", "apse_profile", "apse_profile_read", "apse_conv_packed_elems", "apse_conv_pack_weight", "apse_conv2d", "apse_maxpool3x3s2",
           "apse_maxpool3x3s2_typed", "apse_roi_align", "apse_roi_align_typed", "apse_roi_pool", "apse_roi_features", "apse_nms_rank", "apse_mask_centroid_dense", "apse_mask_closest_dense",
           "apse_l2_normalize", "apse_sqdist", "apse_undistort_gamma", "apse_lab_tables_host", "apse_resize_normalize", "apse_replay_create", "apse_replay_destroy",
           "apse_replay_step", "apse_replay_packed", "apse_replay_max_id", "apse_replay_next_id"]


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, ctx=None, what=""):
    if rc != APSE_OK:
        msg = load().apse_last_error(ctx).decode() if ctx is not None else ""
        raise ApseError("%s failed (code %d) %s" % (what, rc, msg))


def ptr(t):
    """Device (or host) pointer of a contiguous torch tensor / numpy array, or NULL."""
    if t is None:
        return C.c_void_p(0)
    if isinstance(t, np.ndarray):
        assert t.flags["C_CONTIGUOUS"]
        return C.c_void_p(t.ctypes.data)
    assert t.is_contiguous()
    return C.c_void_p(t.data_ptr())
