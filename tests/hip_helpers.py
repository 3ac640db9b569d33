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
