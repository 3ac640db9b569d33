#!/usr/bin/env python3
"""Times apse_conv2d for representative layer shapes over tile configs and split-K factors (GPU)."""
import ctypes as C
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apse_uav_amd import _lib

lib = _lib.load()
SHAPES = [  # name, B, H, W, Cin, Cout, K, stride, pad
    ("res4.c1", 1, 48, 84, 1024, 256, 1, 1, 0), ("res4.c2", 1, 48, 84, 256, 256, 3, 1, 1), ("res4.c3", 1, 48, 84, 256, 1024, 1, 1, 0),
    ("res5.c2", 1, 24, 42, 512, 512, 3, 1, 1), ("res5.c1", 1, 24, 42, 2048, 512, 1, 1, 0), ("res5.c3", 1, 24, 42, 512, 2048, 1, 1, 0),
    ("res3.c2", 1, 96, 168, 128, 128, 3, 1, 1), ("res3.c3", 1, 96, 168, 128, 512, 1, 1, 0), ("res3.c1", 1, 96, 168, 512, 128, 1, 1, 0),
    ("res2.c3", 1, 192, 336, 64, 256, 1, 1, 0), ("res2.c2", 1, 192, 336, 64, 64, 3, 1, 1), ("lat2", 1, 192, 336, 256, 256, 1, 1, 0),
    ("res2.sc", 1, 192, 336, 64, 256, 1, 1, 0), ("res3.sc", 1, 192, 336, 256, 512, 1, 2, 0), ("lat3", 1, 96, 168, 512, 256, 1, 1, 0), ("res3.c1b", 1, 96, 168, 512, 128, 1, 1, 0), ("res4.sc", 1, 96, 168, 512, 1024, 1, 2, 0), ("lat4", 1, 48, 84, 1024, 256, 1, 1, 0), ("res2.c1", 1, 192, 336, 256, 64, 1, 1, 0), ("out3", 1, 96, 168, 256, 256, 3, 1, 1), ("out2", 1, 192, 336, 256, 256, 3, 1, 1), ("fc1", 1000, 7, 7, 256, 1024, 7, 1, 0), ("mask8", 8, 14, 14, 256, 256, 3, 1, 1), ("rpn_head", 1, 1, 85932, 256, 15, 1, 1, 0), ("mlogit", 8, 28, 28, 256, 4, 1, 1, 0),
]
PREC = 1 if "--bf16" in sys.argv else 0
ST16 = 1 if "--st16" in sys.argv else 0          # 16-bit activation storage for x / y (and the residual with --res)
RES = 1 if "--res" in sys.argv else 0
BMUL = 1
for a in sys.argv[1:]:
    if a.startswith("--batch="):
        BMUL = int(a.split("=")[1])
names = [a for a in sys.argv[1:] if not a.startswith("--")]
if names:
    SHAPES = [s for s in SHAPES if s[0] in names]
dev = "cuda"
for (name, B, H, W, Cin, Cout, K, st, pad) in SHAPES:
    B = B * BMUL
    d = _lib.ConvDesc()
    d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad = B, H, W, Cin, Cout, K, K, st, pad
    d.relu, d.res_mode, d.prec = 1, RES, PREC
    if PREC and ST16:
        d.x_st = d.y_st = PREC
        d.res_st = PREC if RES else 0
    OH = (H + 2 * pad - K) // st + 1
    OW = (W + 2 * pad - K) // st + 1
    M = B * OH * OW
    flops = 2.0 * M * Cout * K * K * Cin
    tdt = torch.bfloat16 if (PREC and ST16) else torch.float32
    x = torch.randn(B, H, W, Cin, device=dev).to(tdt)
    wpk = torch.randn(lib.apse_conv_packed_elems(C.byref(d)), device=dev) * 0.01
    bias = torch.zeros(((Cout + 127) // 128) * 128, device=dev)
    y = torch.empty(B, OH, OW, Cout, device=dev, dtype=tdt)
    rsd = torch.randn(B, OH, OW, Cout, device=dev).to(tdt) if RES else None
    ws = torch.empty(16 * M * Cout + 16, device=dev)
    res = []
    for cfg in ((0, 1, 3, 6, 8, 9, 10, 11) if (PREC and ST16) else (0, 1, 3, 6) if PREC else (0, 1, 2, 3, 4, 5, 6, 7, 9, 13)):
        if cfg in (0, 8) and Cout < 128:
            continue
        for sk in (1, 2, 3, 4, 6, 8, 16):
            if cfg in (9, 10, 11, 13) and sk > 1:
                continue
            d.cfg, d.splitk = cfg, sk
            steps = K * ((K * Cin + 31) // 32)
            if sk > max(1, steps // 2):
                continue
            ok = True
            for it in range(3):
                rc = lib.apse_conv2d(C.byref(d), _lib.ptr(x), _lib.ptr(wpk), _lib.ptr(bias), _lib.ptr(rsd), _lib.ptr(y), _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr())
                ok = ok and rc == 0
            if not ok:
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for it in range(n):
                lib.apse_conv2d(C.byref(d), _lib.ptr(x), _lib.ptr(wpk), _lib.ptr(bias), _lib.ptr(rsd), _lib.ptr(y), _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr())
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / n
            res.append((us, cfg, sk))
    res.sort()
    byts = (x.numel() * x.element_size() + y.numel() * y.element_size() * (2 if RES else 1))
    print("%-8s M=%6d N=%5d K=%5d  %6.1f MB  best: %s" % (name, M, Cout, K * K * Cin, byts / 1e6, "  ".join("cfg%d/sk%d %.1fus %.0fTF %.2fTB/s" % (c, s, u, flops / u / 1e6, byts / u / 1e6) for u, c, s in res[:7])))
