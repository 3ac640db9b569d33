#!/usr/bin/env python3
"""Times ONE 16-bit-storage convolution layer in isolation (HIP events around `iters` back-to-back launches on the launch
stream) and prints its checksum, so that two libraries (APSE_HIP_LIB) or two settings of an A/B switch can be compared on the
same box:  python tools/layer_probe.py <name> <prec 1|2> [cfg] [iters]
Shapes are the fp16 batch-8 / bf16 batch-4 layers of the 4K frame (M = batch x rows x cols of the level)."""
import ctypes as C
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apse_uav_amd import _lib

# name: (B, H, W, Cin, Cout, K, stride, pad, res_mode)
SH = {
    "res4.c3.b8": (8, 50, 84, 256, 1024, 1, 1, 0, 1), "res4.c1.b8": (8, 50, 84, 1024, 256, 1, 1, 0, 0),
    "res4.c2.b8": (8, 50, 84, 256, 256, 3, 1, 1, 0), "res3.c3.b8": (8, 100, 168, 128, 512, 1, 1, 0, 1),
    "res2.c3.b8": (8, 200, 336, 64, 256, 1, 1, 0, 1), "res4.c3.b4": (4, 50, 84, 256, 1024, 1, 1, 0, 1),
    "res5.c3.b8": (8, 25, 42, 512, 2048, 1, 1, 0, 1),
    # res4 conv3 with exactly 256 / 512 row blocks of 128
    "c3.even8": (8, 64, 64, 256, 1024, 1, 1, 0, 1), "c3.even16": (16, 64, 64, 256, 1024, 1, 1, 0, 1), "lat2.b8": (8, 200, 336, 256, 256, 1, 1, 0, 2),
}


def main():
    name, prec = sys.argv[1], int(sys.argv[2])
    cfg = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 50
    B, H, W, Cin, Cout, K, st, pad, res_mode = SH[name]
    if os.environ.get("PROBE_NO_RESIDUAL"):
        res_mode = 0
    lib = _lib.load()
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if prec == 1 else torch.float16
    d = _lib.ConvDesc()
    d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad = B, H, W, Cin, Cout, K, K, st, pad
    d.relu, d.res_mode, d.cfg, d.splitk, d.prec, d.fuse_reduce = 1, res_mode, cfg, 0, prec, 0
    d.x_st, d.res_st, d.y_st = prec, prec if res_mode else 0, prec
    g = torch.Generator().manual_seed(7)
    w = (torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5).numpy()
    packed = np.zeros(lib.apse_conv_packed_elems(C.byref(d)), np.float32)
    _lib.check(lib.apse_conv_pack_weight(C.byref(d), _lib.ptr(np.ascontiguousarray(w)), Cin, None, _lib.ptr(packed)), None, "pack")
    OH, OW = (H + 2 * pad - K) // st + 1, (W + 2 * pad - K) // st + 1
    x = torch.randn(B, H, W, Cin, generator=g).to(dev).to(dt)
    wd = torch.from_numpy(packed).to(dev)
    bd = torch.randn(((Cout + 127) // 128) * 128, generator=g).to(dev)
    rd = None
    if res_mode == 1:
        rd = torch.randn(B, OH, OW, Cout, generator=g).to(dev).to(dt)
    elif res_mode == 2:
        rd = torch.randn(B, OH // 2, OW // 2, Cout, generator=g).to(dev).to(dt)
    # a ring of outputs larger than the Infinity Cache, as in the network (every layer writes a fresh map)
    ring = [torch.empty(B, OH, OW, Cout, device=dev, dtype=dt) for _ in range(8)]
    ws = torch.empty((16,), device=dev)

    def launch(y):
        rc = lib.apse_conv2d(C.byref(d), _lib.ptr(x), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(rd), _lib.ptr(y), _lib.ptr(ws), ws.numel() * 4,
                             _lib.stream_ptr())
        assert rc == 0, rc

    for i in range(5):
        launch(ring[i % 8])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        launch(ring[i % 8])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    digest = hashlib.sha256(ring[0].view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
    M = B * OH * OW
    gb = (M * Cin * (K * K if K == 1 else 1) + M * Cout * (2 if res_mode == 1 else 1)) * 2 / 1e9
    print("%-12s prec %d cfg %3d  %8.2f us  %7.1f TFLOP/s  %5.2f TB/s (activations + residual + output)  sha %s  %s"
          % (name, prec, cfg, us, 2.0 * M * Cout * Cin * K * K / us / 1e6, gb / us * 1e3, digest, os.environ.get("APSE_HIP_LIB", "tree")))


main()
