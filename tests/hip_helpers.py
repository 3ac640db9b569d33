"""Helpers for the -m gpu parity tests: thin wrappers over the C ABI (apse_uav_amd._lib)."""
import ctypes as C

import numpy as np
import torch

from apse_uav_amd import _lib


def to_nhwc(x, cpad=None):
    x = x.permute(0, 2, 3, 1).contiguous()
    if cpad and cpad > x.shape[3]:
        x = torch.nn.functional.pad(x, (0, cpad - x.shape[3]))
    return x.contiguous()


def hip_conv2d(x_nchw, w_oihw, bias=None, stride=1, pad=0, relu=False, residual=None, res_mode=0, cfg=-1, splitk=0,
               scale=None, prec=0, fuse=0, x_st=0, res_st=0, y_st=0):
    """x: CPU NCHW f32; returns CPU NCHW f32 computed by apse_conv2d on cuda:0."""
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B, Cin, H, W = x_nchw.shape
    Cout, _, KH, KW = w_oihw.shape
    cin_p = 4
    while cin_p < Cin:
        cin_p *= 2
    d = _lib.ConvDesc()
    d.B, d.H, d.W, d.Cin = B, H, W, cin_p
    d.Cout, d.KH, d.KW, d.stride, d.pad = Cout, KH, KW, stride, pad
    d.relu, d.res_mode, d.cfg, d.splitk, d.prec, d.fuse_reduce = int(relu), res_mode, cfg, splitk, prec, fuse
    d.x_st, d.res_st, d.y_st = x_st, res_st, y_st
    tdt = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}
    packed = np.zeros(lib.apse_conv_packed_elems(C.byref(d)), np.float32)
    w = np.ascontiguousarray(w_oihw.numpy(), np.float32)
    sc = None if scale is None else np.ascontiguousarray(scale.numpy(), np.float32)
    _lib.check(lib.apse_conv_pack_weight(C.byref(d), _lib.ptr(w), Cin, _lib.ptr(sc) if sc is not None else None,
                                         _lib.ptr(packed)), None, "pack")
    bias_p = torch.zeros(((Cout + 127) // 128) * 128)
    if bias is not None:
        bias_p[:Cout] = bias
    xd = to_nhwc(x_nchw, cin_p).to(dev).to(tdt[x_st]).contiguous()
    wd = torch.from_numpy(packed).to(dev)
    bd = bias_p.to(dev)
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    y = torch.full((B, OH, OW, Cout), float("nan"), device=dev, dtype=tdt[y_st])
    rd = None
    if residual is not None:
        rd = to_nhwc(residual).to(dev).to(tdt[res_st]).contiguous()
    ws = torch.empty((64 * B * OH * OW * Cout + 16,), device=dev)
    rc = lib.apse_conv2d(C.byref(d), _lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(rd), _lib.ptr(y), _lib.ptr(ws),
                         ws.numel() * 4, _lib.stream_ptr())
    assert rc == 0, "apse_conv2d rc=%d" % rc
    torch.cuda.synchronize()
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def err_stats(a, b):
    a = a.double()
    b = b.double()
    diff = (a - b).abs()
    scale = b.abs().max().clamp_min(1e-30)
    return dict(max_abs=float(diff.max()), rel_to_max=float(diff.max() / scale), nan=int(torch.isnan(a).sum()))


def _live_bytes(model, res):
    """Every live byte a frame produces: the record of image 0 (boxes, scores, classes, centroids, mass, rects,
    closest-point table, embeddings), the raw embeddings, the mask logits and the bit planes of its mask windows."""
    rec = res.record(0)
    parts = [np.ascontiguousarray(rec[k]).tobytes() for k in
             ("boxes", "scores", "classes", "centroids", "mass", "rects", "closest", "embeddings", "packed_index")]
    inst = model.instances_from(res, 0, want_masks=True)
    for m in inst.pred_masks:
        if m.bits is not None:
            parts.append(m.bits.cpu().numpy().tobytes())
    n = len(rec["scores"])
    logits = model.debug_tensor("mask_logits").cpu().numpy().reshape(-1)[: n * 28 * 28 * 4]
    parts.append(logits.tobytes())
    parts.append(model.debug_tensor("embedding_raw").cpu().numpy().reshape(-1)[: n * 128].tobytes())
    return n, b"".join(parts)


def history_independence(tracker, frame_a, image_hw, rng_seed=0):
    """Frame A after forwards that left 0, 8 and 100 detections in the packed list must give the same BYTES: the
    tile shape / K split of the count-limited GEMMs (mask head, deconv, mask logits, association FC) are plan
    constants, the previous count only sizes their grid (csrc/detector.hip add_conv / run_plan)."""
    pr = tracker.predictor
    model = pr.model
    dev = pr._upload([frame_a])
    rng = np.random.RandomState(rng_seed)
    ih, iw = image_hw
    outs = []
    for prev in (0, 8, 100):
        x0 = rng.uniform(0, iw - 60, prev)
        y0 = rng.uniform(0, ih - 60, prev)
        boxes = np.stack([x0, y0, x0 + rng.uniform(8, 56, prev), y0 + rng.uniform(8, 56, prev)], 1).astype(np.float32)
        given = (boxes.reshape(-1, 4), np.zeros(prev, np.int32), np.array([prev], np.int32))
        model.preprocess_frames(dev)
        model.run(1, given)
        assert model.read(1).total == prev                 # the count the next forward sees as its hint
        model.preprocess_frames(dev)
        model.run(1)
        outs.append(_live_bytes(model, model.read(1)))
    return outs

