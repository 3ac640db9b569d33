#!/usr/bin/env python3
"""Runs ONE convolution shape / tile config a few times (for rocprofv3 --pmc passes)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apse_uav_amd import _lib

SH = {"out2": (1, 192, 336, 256, 256, 3, 1, 1), "res4.c2": (1, 48, 84, 256, 256, 3, 1, 1), "res4.c1": (1, 48, 84, 1024, 256, 1, 1, 0),
      "res4.c3": (1, 48, 84, 256, 1024, 1, 1, 0), "res2.c3": (1, 192, 336, 64, 256, 1, 1, 0)}
name, cfg, sk, prec, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
B, H, W, Cin, Cout, K, st, pad = SH[name]
lib = _lib.load()
d = _lib.ConvDesc()
d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad = B, H, W, Cin, Cout, K, K, st, pad
d.relu, d.res_mode, d.cfg, d.splitk, d.prec, d.fuse_reduce = 1, 0, cfg, sk, prec, 0
OH, OW = (H + 2 * pad - K) // st + 1, (W + 2 * pad - K) // st + 1
x = torch.randn(B, H, W, Cin, device="cuda")
w = torch.randn(lib.apse_conv_packed_elems(C.byref(d)), device="cuda") * 0.01
b = torch.zeros(((Cout + 127) // 128) * 128, device="cuda")
y = torch.empty(B, OH, OW, Cout, device="cuda")
ws = torch.empty(max(sk, 1) * B * OH * OW * Cout + 16, device="cuda")
for _ in range(iters):
    assert lib.apse_conv2d(C.byref(d), _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, _lib.ptr(y), _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()) == 0
torch.cuda.synchronize()
print("done", name, float(y.abs().mean()))
