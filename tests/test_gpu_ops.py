"""-m gpu: stage-level parity of the HIP operators against the CPU oracle / torch CPU f32.

Every case feeds identical seeded inputs to both sides.  Float outputs: tolerance written
per test (f32 accumulation order differs; the MFMA path is an exact-f32 fma chain).
Integer / index outputs: exact.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _log(logdir, name, obj):
    with open(os.path.join(logdir, "ops_parity.log"), "a") as f:
        f.write(name + " " + json.dumps(obj) + "\n")


CONV_CASES = [
    # name, B, Cin, H, W, Cout, K, stride, pad, relu, res_mode, cfg, splitk
    ("c1x1_128", 1, 64, 24, 40, 256, 1, 1, 0, True, 0, 0, 0),
    ("c1x1_64t", 1, 256, 20, 28, 64, 1, 1, 0, True, 0, 1, 0),
    ("c3x3_auto", 1, 64, 30, 44, 64, 3, 1, 1, True, 0, -1, 0),
    ("c3x3_128", 2, 128, 17, 23, 128, 3, 1, 1, False, 1, 0, 0),
    ("c1x1_s2", 1, 256, 24, 36, 128, 1, 2, 0, True, 0, -1, 0),
    ("c1x1_n32", 1, 256, 19, 21, 15, 1, 1, 0, False, 0, 2, 0),
    ("c3x3_n64", 1, 64, 26, 38, 64, 3, 1, 1, True, 0, 3, 0),
    ("stem7x7", 1, 3, 64, 96, 64, 7, 2, 3, True, 0, -1, 0),
    ("splitk", 1, 512, 12, 14, 128, 3, 1, 1, True, 1, 1, 5),
    ("upres", 1, 128, 16, 24, 256, 1, 1, 0, False, 2, -1, 0),
    ("fc7x7", 37, 256, 7, 7, 1024, 7, 1, 0, True, 0, -1, 0),
    ("fc_small_m", 3, 256, 10, 10, 128, 10, 1, 0, False, 0, -1, 0),
    ("big_k", 1, 2048, 6, 10, 512, 1, 1, 0, True, 0, -1, 0),
    # the remaining tile shapes: 32-deep 64x64 / 128x32, and the 8-wave two-k-group 64x64 shapes (64- and 128-deep steps),
    # with ragged M, odd step counts, a residual and a split on top
    ("t64_k32", 1, 96, 21, 27, 192, 3, 1, 1, True, 1, 4, 0),
    ("t128x32_k32", 2, 64, 13, 19, 40, 1, 1, 0, False, 0, 5, 0),
    ("w8_k64", 1, 160, 23, 29, 132, 3, 1, 1, True, 1, 6, 0),
    ("w8_k128", 1, 288, 18, 22, 256, 1, 1, 0, True, 2, 7, 0),
    ("w8_k128_3x3", 2, 64, 15, 17, 64, 3, 2, 1, False, 0, 7, 0),
    ("w8_split", 1, 512, 9, 11, 128, 3, 1, 1, True, 1, 6, 3),
    # 256x128 tile (8 waves as 4x2; 16-bit operands only -- the f32 path maps it to 128x128): ragged M, two n tiles, residual
    ("t256x128", 1, 128, 40, 52, 256, 3, 1, 1, True, 1, 8, 0),
    ("t256x128_1x1", 2, 256, 33, 31, 384, 1, 1, 0, False, 0, 8, 0),
    # conv_glds16 (cfg 11: the LDS-DMA 256x128 kernel; 16-bit stored operands only): 3x3 with padding taps on every border and a
    # ragged last tile, one / two / many 64-deep stages, stride 2, both residual forms, Cout not a multiple of the tile
    ("g_3x3_c64", 1, 64, 30, 44, 64, 3, 1, 1, True, 1, 11, 0),
    ("g_3x3_c256", 2, 256, 21, 19, 256, 3, 1, 1, True, 0, 11, 0),
    ("g_1x1_k64", 1, 64, 17, 23, 128, 1, 1, 0, False, 0, 11, 0),
    ("g_1x1_k128", 1, 128, 16, 24, 200, 1, 1, 0, True, 2, 11, 0),
    ("g_3x3_s2", 1, 128, 27, 33, 136, 3, 2, 1, False, 0, 11, 0),
    ("g_fc7x7", 37, 256, 7, 7, 1024, 7, 1, 0, True, 0, 11, 0),
    ("g_wide_256x256", 1, 64, 361, 359, 256, 3, 1, 1, True, 1, 11, 0),      # >= 500 tiles of 256x256: the two-stage wide-tile variant
    # round 4: the kernel picks its 128x128 tile below 200 tiles of 256x128 (every case above but the wide one) and 256x128 from there on
    ("g_256x128_many", 1, 64, 230, 231, 128, 3, 1, 1, True, 1, 11, 0),      # 208 tiles of 256x128, ragged last tile, residual
    ("g_256x128_1x1_k512", 1, 512, 228, 230, 136, 1, 1, 0, False, 0, 11, 0),  # 8 stages, Cout not a multiple of the tile (2 x 205 tiles)
    ("g_128x128_two_stages", 2, 128, 31, 29, 256, 1, 1, 0, True, 1, 11, 0),   # T = 2: prologue and drain of the ping-pong loop only
    ("g_128x128_one_stage", 1, 64, 40, 44, 128, 1, 1, 0, False, 0, 11, 0),    # T = 1
    # 128x256 tile (Cout = 256 and 200..640 tiles of 128 rows): every output column in one tile, three stages of 48 KB
    ("g_128x256_tall", 1, 64, 161, 160, 256, 3, 1, 1, True, 1, 11, 0),       # 202 tiles, ragged last one, residual
    ("g_128x256_1x1_k512", 1, 512, 163, 161, 256, 1, 1, 0, False, 0, 11, 0),  # 8 stages
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d_parity(case, logdir):
    from hip_helpers import hip_conv2d, err_stats
    name, B, Cin, H, W, Cout, K, stride, pad, relu, res_mode, cfg, splitk = case
    if cfg == 11:
        pytest.skip("conv_glds16 takes 16-bit stored operands only (test_conv2d_16bit_storage)")
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    res = None
    if res_mode == 1:
        res = torch.randn(ref.shape, generator=g)
        ref = ref + res
    elif res_mode == 2:
        res = torch.randn(B, Cout, ref.shape[2] // 2, ref.shape[3] // 2, generator=g)
        ref = ref + F.interpolate(res, scale_factor=2, mode="nearest")
    if relu:
        ref = F.relu(ref)
    out = hip_conv2d(x, w, b, stride, pad, relu, res, res_mode, cfg, splitk)
    st = err_stats(out, ref)
    _log(logdir, "conv/" + name, st)
    assert st["nan"] == 0
    assert st["rel_to_max"] < 2e-5, st          # f32 tolerance: accumulation-order noise only


@pytest.mark.parametrize("prec", [0])
@pytest.mark.parametrize("shape", [(1, 512, 12, 14, 128, 3, 5, 1), (1, 256, 48, 84, 256, 3, 8, 0), (3, 256, 10, 10, 128, 10, 64, 1),
                                   (1, 1024, 24, 42, 512, 1, 4, 1)])
def test_conv2d_fused_splitk(shape, prec, logdir):
    """Split-K with the in-launch reduction (last-arriving K-slice sums the slabs): must equal the separate
    reduce-kernel path bit for bit (same fixed summation order), run repeatedly on a busy GPU."""
    from hip_helpers import hip_conv2d
    B, Cin, H, W, Cout, K, sk, cfg = shape
    g = torch.Generator().manual_seed(Cin + Cout + sk)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    pad = 1 if K == 3 else 0
    ref = hip_conv2d(x, w, b, 1, pad, True, None, 0, cfg, sk, prec=prec, fuse=0)
    for rep in range(6):
        out = hip_conv2d(x, w, b, 1, pad, True, None, 0, cfg, sk, prec=prec, fuse=1)
        assert torch.equal(out, ref), (rep, float((out - ref).abs().max()))


@pytest.mark.parametrize("prec", [1, 2], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[2] >= 8], ids=[c[0] for c in CONV_CASES if c[2] >= 8])
def test_conv2d_16bit_storage(case, prec, logdir):
    """16-bit STORAGE mode: activations, residual and output live in HBM as bf16 / f16; operands go to the
    matrix cores without conversion, accumulation is f32, the output is rounded once at the store."""
    from hip_helpers import hip_conv2d, err_stats
    import zlib
    name, B, Cin, H, W, Cout, K, stride, pad, relu, res_mode, cfg, splitk = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)
    dt = torch.bfloat16 if prec == 1 else torch.float16
    r16 = lambda t: t.to(dt).to(torch.float32)
    x = r16(torch.randn(B, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, r16(w), b, stride=stride, padding=pad)
    res = None
    if res_mode == 1:
        res = r16(torch.randn(ref.shape, generator=g))
        ref = ref + res
    elif res_mode == 2:
        res = r16(torch.randn(B, Cout, ref.shape[2] // 2, ref.shape[3] // 2, generator=g))
        ref = ref + F.interpolate(res, scale_factor=2, mode="nearest")
    if relu:
        ref = F.relu(ref)
    out = hip_conv2d(x, w, b, stride, pad, relu, res, res_mode, cfg, splitk, prec=prec, x_st=prec, res_st=prec if res is not None else 0,
                     y_st=prec)
    # the kernel's f32 result differs from the reference's by summation order only; after the 16-bit store the two
    # agree except where that tiny difference straddles a rounding boundary (then by one 16-bit ulp)
    ulp = 2.0 ** (-7 if prec == 1 else -10)        # one unit in the last place, relative to the binade's lower edge
    diff = (out - r16(ref)).abs()
    # + 4e-6: f32 summation-order noise of a unit-variance sum that cancels to |ref| < 1e-3 [round 4: 7 of 7.1 M outputs of the K = 512
    # case sit at 2.3e-6 from the reference with |ref| = 6e-4; the LDS-DMA kernel and the tiled kernel agree on them bit for bit]
    tol = ulp * ref.abs().clamp_min(1e-3) * 1.01 + 4e-6
    frac = float((diff > 1e-6 * ref.abs().clamp_min(1.0)).float().mean())
    worst = int(torch.argmax(diff / tol))
    info = dict(max=float(diff.max()), frac_differs=frac, violations=int((diff > tol).sum()), worst_ratio=float((diff / tol).flatten()[worst]),
                worst_ref=float(ref.flatten()[worst]), worst_out=float(out.flatten()[worst]))
    if cfg == 11:
        # the LDS-DMA kernel (any of its tiles, ping-pong schedule) runs the MFMA chain of the register-staged tiles: same bits
        tiled = hip_conv2d(x, w, b, stride, pad, relu, res, res_mode, 0, splitk, prec=prec, x_st=prec, res_st=prec if res is not None else 0,
                           y_st=prec)
        info["equal_to_tiled"] = bool(torch.equal(out, tiled))
        info["unequal_elems"] = int((out != tiled).sum())
    _log(logdir, "conv_16bit_storage/%d/" % prec + name, info)
    if cfg == 11:
        assert info["equal_to_tiled"], info
    assert bool((diff <= tol).all()) and frac < 0.02, info
    # f32-compute kernel reading 16-bit activations (decision layers in 16-bit storage mode)
    if cfg in (-1, 2) and res_mode == 0:
        out32 = hip_conv2d(x, w, b, stride, pad, relu, None, 0, cfg, splitk, prec=0, x_st=prec)
        ref32 = F.conv2d(x, w, b, stride=stride, padding=pad)
        ref32 = F.relu(ref32) if relu else ref32
        assert err_stats(out32, ref32)["rel_to_max"] < 2e-5


@pytest.mark.parametrize("prec", [1, 2], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", [(3, 256, 28, 28, 256, 3, 1), (5, 256, 14, 14, 256, 3, 1), (2, 256, 37, 31, 1024, 1, 0), (1, 512, 20, 26, 128, 3, 1)])
def test_conv2d_16bit_tiles_are_bit_identical(shape, prec):
    """16-bit operands, unsplit: the single-k-group 4-wave tiles -- 128x128 (cfg 0), 64x64 (cfg 1), 128x64 (cfg 3) -- add every
    accumulator's products in the same ascending-k order, so they must give the SAME BITS (ragged M, residual, ReLU included).  The
    library relies on it: for the GEMMs over the packed detection list the tile follows the hinted row count (detector.hip run_plan),
    and a frame's results must not depend on that hint."""
    from hip_helpers import hip_conv2d
    B, Cin, H, W, Cout, K, pad = shape
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + K)
    dt = torch.bfloat16 if prec == 1 else torch.float16
    r16 = lambda t: t.to(dt).to(torch.float32)
    x = r16(torch.randn(B, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    res = r16(torch.randn(B, Cout, H, W, generator=g))
    outs = [hip_conv2d(x, w, b, 1, pad, True, res, 1, cfg, 0, prec=prec, x_st=prec, res_st=prec, y_st=prec) for cfg in (0, 1, 3)]
    assert int(torch.isnan(outs[0]).sum()) == 0 and float(outs[0].abs().max()) > 0
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


STREAM_CASES = [
    # name, B, Cin (= K), H, W, Cout, stride, relu, res_mode      -- the memory-streaming 1x1 kernel (csrc/conv1x1_stream.hip, cfg 9)
    ("s_k64_n256_res", 1, 64, 24, 40, 256, 1, True, 1),          # res2 conv3: residual + ReLU
    ("s_k64_n256_ragged", 2, 64, 13, 19, 256, 1, False, 0),      # M = 494: ragged last block, rows past M
    ("s_k128_n512_res", 1, 128, 18, 26, 512, 1, True, 1),        # res3 conv3
    ("s_k256_n1024_res", 1, 256, 12, 21, 1024, 1, True, 1),      # res4 conv3 (N chunks split over blockIdx.y)
    ("s_k256_n256_up", 1, 256, 16, 24, 256, 1, False, 2),        # FPN lateral: nearest-2x upsampled top-down add
    ("s_k256_n512_s2", 1, 256, 24, 36, 512, 2, False, 0),        # stride-2 shortcut (res3.0)
    ("s_k64_n128", 3, 64, 9, 11, 128, 1, True, 0),               # a single chunk
    ("s_k256_n256_bigmap", 1, 256, 184, 180, 256, 1, False, 2),
    ("s_k256_n64_bigmap", 1, 256, 182, 181, 64, 1, True, 0),     # res2 conv1 shape class (f32 only: N = 64 needs the 64-column chunks)
    # round 4: the fp16 batch-8 res4 conv3 itself (263 row blocks, one per CU) and ragged / split-N neighbours of it
    ("s_k256_n1024_res4_b8", 8, 256, 50, 84, 1024, 1, True, 1),
    ("s_k256_n1024_pairs", 1, 256, 64, 128, 1024, 1, True, 1),
    ("s_k256_n1024_ragged", 1, 256, 97, 127, 1024, 1, False, 1),
    ("s_k256_n512_small", 3, 256, 23, 29, 512, 1, True, 1),
    ("s_k128_n192", 2, 128, 15, 13, 192, 1, True, 1),            # f32 only: N a multiple of 64, not of 128  # FPN lateral 2 shape class: >= 32768 rows per image (f32 takes K = 256 there)
]

@pytest.mark.parametrize("prec", [0, 1, 2], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", STREAM_CASES, ids=[c[0] for c in STREAM_CASES])
def test_conv1x1_stream(case, prec, logdir):
    """conv1x1_stream (cfg 9): f32 against torch CPU f32 AND bit-for-bit against the tiled kernel (same MFMA chain, same k
    order); 16-bit storage modes within one ulp of the storage type of the rounded f32 reference."""
    from hip_helpers import hip_conv2d, err_stats
    import zlib
    name, B, Cin, H, W, Cout, stride, relu, res_mode = case
    cfg = 9
    if prec != 0 and Cout % 128:
        pytest.skip("16-bit operands: 128-column chunks only")
    if prec == 0 and Cin == 256 and (stride != 1 or (H // stride) * (W // stride) < 32768):
        pytest.skip("f32 operands, K = 256: the streaming kernel is taken for maps of >= 32768 rows only (equal to the tiled kernel below)")
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)
    dt = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}[prec]
    r16 = lambda t: t.to(dt).to(torch.float32)
    x = r16(torch.randn(B, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, r16(w), b, stride=stride)
    res = None
    mag = ref.abs()                                   # |terms| of the final sum: bounds the f32 summation-order noise where they cancel
    if res_mode == 1:
        res = r16(torch.randn(ref.shape, generator=g))
        mag = mag + res.abs()
        ref = ref + res
    elif res_mode == 2:
        res = r16(torch.randn(B, Cout, ref.shape[2] // 2, ref.shape[3] // 2, generator=g))
        up = F.interpolate(res, scale_factor=2, mode="nearest")
        mag = mag + up.abs()
        ref = ref + up
    if relu:
        ref = F.relu(ref)
    kw = dict(prec=prec, x_st=prec, res_st=prec if res is not None else 0, y_st=prec)
    out = hip_conv2d(x, w, b, stride, 0, relu, res, res_mode, cfg, 0, **kw)
    tiled = hip_conv2d(x, w, b, stride, 0, relu, res, res_mode, 0, 0, **kw)
    assert int(torch.isnan(out).sum()) == 0
    if prec == 0:
        st = err_stats(out, ref)
        _log(logdir, "conv_stream/f32/" + name, dict(st, equal_to_tiled=bool(torch.equal(out, tiled))))
        assert st["rel_to_max"] < 2e-5, st
        assert torch.equal(out, tiled)
    else:
        ulp = 2.0 ** (-7 if prec == 1 else -10)
        diff = (out - r16(ref)).abs()
        tol = ulp * ref.abs().clamp_min(1e-3) * 1.01 + 1e-6 + 2e-6 * mag       # last term: conv + residual that cancel (8.5 M outputs in the big-map case)
        frac = float((diff > 1e-6 * ref.abs().clamp_min(1.0)).float().mean())
        same = float((out == tiled).float().mean())
        _log(logdir, "conv_stream/%d/" % prec + name, dict(max=float(diff.max()), frac_differs=frac, equal_to_tiled_frac=same))
        assert bool((diff <= tol).all()) and frac < 0.02
        assert same == 1.0                      # [1.0 in every logged case, rounds 2-3] same MFMA chain, bias / residual added at the same points: bit-identical


SKINNY_CASES = [
    # name, B, H, W, Cout, relu      -- conv_skinny16 (csrc/conv_skinny.hip, cfg 13): 1x1 over 256 channels, <= 16 outputs
    ("sk16_rpn_head", 1, 37, 53, 15, False),          # 3 objectness + 12 deltas; M = 1961: ragged last group of tiles
    ("sk16_mask_logits", 3, 28, 28, 4, False),        # 4 class logits per mask pixel
    ("sk16_full16_relu", 2, 16, 24, 16, True),
    ("sk16_tiny", 1, 3, 5, 1, False),                 # fewer rows than one tile
]


@pytest.mark.parametrize("xst", [0, 1, 2], ids=["x_f32", "x_bf16", "x_f16"])
@pytest.mark.parametrize("case", SKINNY_CASES, ids=[c[0] for c in SKINNY_CASES])
def test_conv_skinny16(case, xst, logdir):
    """f32 filters, exact f32 MFMA in every mode; x stored as f32 / bf16 / f16.  Against torch CPU f32 on the same (rounded) x."""
    from hip_helpers import hip_conv2d, err_stats
    import zlib
    name, B, H, W, Cout, relu = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)
    dt = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}[xst]
    x = torch.randn(B, 256, H, W, generator=g).to(dt).to(torch.float32)
    w = torch.randn(Cout, 256, 1, 1, generator=g) / 16.0
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, b)
    if relu:
        ref = F.relu(ref)
    out = hip_conv2d(x, w, b, 1, 0, relu, None, 0, 13, 0, prec=0, x_st=xst, res_st=0, y_st=0)
    tiled = hip_conv2d(x, w, b, 1, 0, relu, None, 0, 2, 0, prec=0, x_st=xst, res_st=0, y_st=0)
    st = err_stats(out, ref)
    _log(logdir, "conv_skinny/%d/" % xst + name, dict(st, vs_tiled=err_stats(out, tiled)["rel_to_max"]))
    assert st["nan"] == 0 and st["rel_to_max"] < 2e-5, st
    assert torch.equal(out, hip_conv2d(x, w, b, 1, 0, relu, None, 0, 13, 0, prec=0, x_st=xst, res_st=0, y_st=0))


def test_conv_skinny16_refuses_ineligible_layers():
    from apse_uav_amd import _lib
    lib = _lib.load()
    for (cin, cout, k) in ((256, 32, 1), (128, 8, 1), (256, 8, 3)):
        d = _lib.ConvDesc()
        d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad = 1, 8, 8, cin, cout, k, k, 1, k // 2
        d.cfg = 13
        x = torch.zeros(1, 8, 8, cin, device="cuda")
        wp = torch.zeros(lib.apse_conv_packed_elems(C.byref(d)), device="cuda")
        y = torch.zeros(1, 8, 8, cout, device="cuda")
        ws = torch.zeros(16, device="cuda")
        assert lib.apse_conv2d(C.byref(d), _lib.ptr(x), _lib.ptr(wp), None, None, _lib.ptr(y), _lib.ptr(ws), 64, _lib.stream_ptr()) != 0


def test_conv1x1_stream_refuses_ineligible_layers():
    """cfg 9 on a layer the streaming kernel does not take (3x3, K = 512, narrow N) is an error, never a silent fallback."""
    from apse_uav_amd import _lib
    lib = _lib.load()
    for (cin, cout, k) in ((64, 256, 3), (512, 256, 1), (64, 64, 1)):
        d = _lib.ConvDesc()
        d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad = 1, 8, 8, cin, cout, k, k, 1, k // 2
        d.cfg = 9
        x = torch.zeros(1, 8, 8, cin, device="cuda")
        wp = torch.zeros(lib.apse_conv_packed_elems(C.byref(d)), device="cuda")
        y = torch.zeros(1, 8, 8, cout, device="cuda")
        ws = torch.zeros(16, device="cuda")
        assert lib.apse_conv2d(C.byref(d), _lib.ptr(x), _lib.ptr(wp), None, None, _lib.ptr(y), _lib.ptr(ws), 64, _lib.stream_ptr()) != 0


def test_maxpool(logdir):
    from apse_uav_amd import _lib
    from hip_helpers import to_nhwc
    x = torch.randn(2, 64, 37, 50)
    ref = F.max_pool2d(x, 3, 2, 1)
    xd = to_nhwc(x).cuda()
    y = torch.empty((2, ref.shape[2], ref.shape[3], 64), device="cuda")
    assert _lib.load().apse_maxpool3x3s2(_lib.ptr(xd), _lib.ptr(y), 2, 37, 50, 64, _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(y.cpu().permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("st,dtype", [(1, torch.bfloat16), (2, torch.float16)])
def test_maxpool_16bit(st, dtype):
    """The 8-channels-per-thread form of the 16-bit modes: max is exact, so bit-equal to torch on the same 16-bit values."""
    from apse_uav_amd import _lib
    from hip_helpers import to_nhwc
    x = torch.randn(2, 64, 37, 51).to(dtype)
    ref = F.max_pool2d(x.float(), 3, 2, 1)
    xd = to_nhwc(x.float()).to(dtype).cuda().contiguous()
    y = torch.empty((2, ref.shape[2], ref.shape[3], 64), device="cuda", dtype=dtype)
    assert _lib.load().apse_maxpool3x3s2_typed(_lib.ptr(xd), _lib.ptr(y), 2, 37, 51, 64, st, _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(y.cpu().float().permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("hw", [(2160, 3840), (540, 960), (333, 516)])
def test_pil_resize_bit_exact(hw, logdir):
    """HIP two-pass integer resize + normalise vs Pillow itself (the reference's resampler)."""
    from PIL import Image
    from apse_uav_amd import _lib
    from apse_uav_amd.utils import resample
    H, W = hw
    rng = np.random.default_rng(H)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    oh, ow = resample.resize_shortest_edge(H, W)
    if H == 333:
        oh, ow = 121, 203
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    hb, hc, hk = resample.precompute_coeffs(W, ow)
    vb, vc, vk = resample.precompute_coeffs(H, oh)
    dev = "cuda"
    ph, pw = (oh + 31) // 32 * 32, (ow + 31) // 32 * 32
    src = torch.from_numpy(img).to(dev)
    tmp = torch.empty((H, ow, 3), dtype=torch.uint8, device=dev)
    out = torch.zeros((1, ph, pw, 4), device=dev)
    rs = torch.empty((oh, ow, 3), dtype=torch.uint8, device=dev)
    t = [torch.from_numpy(a).to(dev) for a in (hb, hc, vb, vc)]
    mean = (C.c_float * 3)(103.530, 116.280, 123.675)
    rc = _lib.load().apse_resize_normalize(_lib.ptr(src), _lib.ptr(tmp), _lib.ptr(out), _lib.ptr(rs), _lib.ptr(t[0]),
                                           _lib.ptr(t[1]), hk, _lib.ptr(t[2]), _lib.ptr(t[3]), vk, 1, H, W, oh, ow, ph, pw,
                                           C.byref(mean), _lib.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    got = rs.cpu().numpy()
    nd = int((got != ref).sum())
    _log(logdir, "resize/%dx%d" % hw, dict(mismatch=nd))
    assert nd == 0
    exp = torch.from_numpy(ref.astype(np.float32)) - torch.tensor([103.530, 116.280, 123.675])
    o = out.cpu()[0]
    assert torch.equal(o[:oh, :ow, :3], exp)
    assert float(o[oh:].abs().sum()) == 0.0 and float(o[:, ow:].abs().sum()) == 0.0 and float(o[..., 3].abs().sum()) == 0.0


def test_roi_align_parity(logdir):
    from oracle import ops
    from apse_uav_amd import _lib
    from hip_helpers import to_nhwc, err_stats
    g = torch.Generator().manual_seed(3)
    sizes = [(48, 84), (24, 42), (12, 21), (6, 11)]
    feats = [torch.randn(1, 256, h, w, generator=g) for h, w in sizes]
    n = 300
    cx = torch.rand(n, generator=g) * 330
    cy = torch.rand(n, generator=g) * 190
    bw = torch.rand(n, generator=g) ** 2 * 300 + 1
    bh = torch.rand(n, generator=g) ** 2 * 180 + 1
    boxes = torch.stack([(cx - bw / 2).clamp(0, 336), (cy - bh / 2).clamp(0, 192), (cx + bw / 2).clamp(0, 336),
                         (cy + bh / 2).clamp(0, 192)], dim=1)
    boxes[0] = torch.tensor([0., 0., 336., 192.])
    boxes[1] = torch.tensor([10., 10., 10.5, 10.2])
    boxes[2] = torch.tensor([0., 80., 336., 108.])        # wide and flat on the finest level: 13 samples per bin across, a
    #                                                       tap window wider than RA_CAP cells -> the per-sample loop
    boxes[3] = torch.tensor([-20., -30., 40., 25.])       # samples left of / above the map
    boxes[4] = torch.tensor([300., 170., 400., 260.])     # samples right of / below the map
    for R in (7, 14):
        ref = ops.roi_pooler([f[0] for f in feats], boxes, R)
        fd = [to_nhwc(f).cuda() for f in feats]
        ptrs = (C.c_void_p * 4)(*[f.data_ptr() for f in fd])
        hs = (C.c_int * 4)(*[s[0] for s in sizes])
        ws = (C.c_int * 4)(*[s[1] for s in sizes])
        out = torch.full((n, R, R, 256), float("nan"), device="cuda")
        bd = boxes.cuda().contiguous()
        assert _lib.load().apse_roi_align(ptrs, hs, ws, _lib.ptr(bd), n, n, R, _lib.ptr(out), _lib.stream_ptr()) == 0
        torch.cuda.synchronize()
        st = err_stats(out.cpu().permute(0, 3, 1, 2), ref)
        _log(logdir, "roi_align/%d" % R, st)
        assert st["nan"] == 0 and st["max_abs"] < 1e-4, st       # f32, bilinear weights: order-of-sum noise


@pytest.mark.parametrize("st,dtype", [(1, torch.bfloat16), (2, torch.float16)])
def test_roi_align_16bit_maps(logdir, st, dtype):
    """The form the 16-bit modes run (maps and output stored as bf16 / f16, two map cells per load, f32 arithmetic):
    equal to the oracle on the SAME rounded maps up to the rounding of the 16-bit output."""
    from oracle import ops
    from apse_uav_amd import _lib
    from hip_helpers import to_nhwc
    g = torch.Generator().manual_seed(5)
    sizes = [(48, 84), (24, 42), (12, 21), (6, 11)]
    feats = [torch.randn(2, 256, h, w, generator=g).to(dtype) for h, w in sizes]
    per = 150
    n = 2 * per
    cx = torch.rand(n, generator=g) * 330
    cy = torch.rand(n, generator=g) * 190
    bw = torch.rand(n, generator=g) ** 2 * 300 + 1
    bh = torch.rand(n, generator=g) ** 2 * 180 + 1
    boxes = torch.stack([(cx - bw / 2).clamp(0, 336), (cy - bh / 2).clamp(0, 192), (cx + bw / 2).clamp(0, 336),
                         (cy + bh / 2).clamp(0, 192)], dim=1)
    boxes[0] = torch.tensor([0., 0., 336., 192.])
    boxes[1] = torch.tensor([10., 10., 10.5, 10.2])         # one sample per bin, window of 1-2 cells (odd tail of a half-wave)
    boxes[2] = torch.tensor([0., 80., 336., 108.])          # per-sample loop
    boxes[per] = torch.tensor([-20., -30., 40., 25.])
    boxes[per + 1] = torch.tensor([300., 170., 400., 260.])
    fd = [to_nhwc(f.float()).to(dtype).cuda().contiguous() for f in feats]
    ptrs = (C.c_void_p * 4)(*[f.data_ptr() for f in fd])
    hs = (C.c_int * 4)(*[s_[0] for s_ in sizes])
    ws = (C.c_int * 4)(*[s_[1] for s_ in sizes])
    bd = boxes.cuda().contiguous()
    for R in (7, 14):
        ref = torch.cat([ops.roi_pooler([f[b].float() for f in feats], boxes[b * per:(b + 1) * per], R) for b in range(2)])
        out = torch.full((n, R, R, 256), float("nan"), device="cuda", dtype=dtype)
        assert _lib.load().apse_roi_align_typed(ptrs, hs, ws, _lib.ptr(bd), n, per, R, st, _lib.ptr(out), _lib.stream_ptr()) == 0
        torch.cuda.synchronize()
        got = out.cpu().float().permute(0, 3, 1, 2)
        assert not torch.isnan(got).any()
        ulp = 2.0 ** (-7 if st == 1 else -10)               # bf16: 8 significand bits, f16: 11
        err = ((got - ref).abs() / ref.abs().clamp_min(0.25)).max().item()
        _log(logdir, "roi_align16/%d/%d" % (st, R), dict(max_rel=err))
        assert err <= 0.51 * ulp * 1.05 + 1e-5, err           # half an ulp of the stored type (+ order-of-sum noise)


def test_roi_pool_parity(logdir):
    from oracle import ops
    from apse_uav_amd import _lib
    from hip_helpers import to_nhwc
    g = torch.Generator().manual_seed(5)
    feat = torch.randn(1, 256, 48, 84, generator=g)
    n = 40
    x1 = torch.rand(n, generator=g) * 800
    y1 = torch.rand(n, generator=g) * 500
    boxes = torch.stack([x1, y1, x1 + torch.rand(n, generator=g) * 300, y1 + torch.rand(n, generator=g) * 150], dim=1)
    boxes[0] = torch.tensor([0., 0., 960., 540.])
    boxes[1] = torch.tensor([955., 530., 960., 540.])
    scale = 84 / 960.0
    rois = torch.cat([torch.zeros(n, 1), boxes], dim=1)
    ref = ops.roi_pool(feat, rois, 10, scale)
    out = torch.full((n, 10, 10, 256), float("nan"), device="cuda")
    fd = to_nhwc(feat).cuda()
    img = torch.zeros(n, dtype=torch.int32, device="cuda")
    bd = boxes.cuda().contiguous()
    assert _lib.load().apse_roi_pool(_lib.ptr(fd), 48, 84, _lib.ptr(bd), _lib.ptr(img), n, 10, float(scale), _lib.ptr(out),
                                     _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(out.cpu().permute(0, 3, 1, 2), ref)            # max of the same cells: bit-exact


@pytest.mark.parametrize("seed,n,ncat,thr,topk", [(0, 5000, 5, 0.7, 1000), (1, 4000, 4, 0.5, 100), (2, 37, 4, 0.5, 100),
                                                   (3, 1, 1, 0.5, 10)])
def test_nms_rank_exact(seed, n, ncat, thr, topk, logdir):
    """Per-category NMS + ranking: kept indices must equal torchvision-style batched_nms exactly."""
    from oracle import ops
    from apse_uav_amd import _lib
    g = torch.Generator().manual_seed(seed)
    cx = torch.rand(n, generator=g) * 1300
    cy = torch.rand(n, generator=g) * 700
    w = torch.rand(n, generator=g) * 200 + 2
    h = torch.rand(n, generator=g) * 200 + 2
    boxes = torch.stack([(cx - w / 2).clamp(0, 1333), (cy - h / 2).clamp(0, 750), (cx + w / 2).clamp(0, 1333),
                         (cy + h / 2).clamp(0, 750)], dim=1)
    scores = torch.rand(n, generator=g)
    scores[::7] = scores[min(3, n - 1)]          # exact ties: exercises the index tie-break
    valid = (torch.rand(n, generator=g) > 0.1)
    if ncat == 5:
        per = n // ncat
        cat = torch.arange(n) // per
        cat_div, cat_mod = per, 0
    else:
        cat = torch.arange(n) % ncat
        cat_div, cat_mod = 0, ncat
    vi = torch.nonzero(valid).squeeze(1)
    kept = ops.batched_nms(boxes[vi], scores[vi], cat[vi], thr)[:topk]
    ref_idx = vi[torch.from_numpy(kept)].numpy()
    dev = "cuda"
    ob = torch.empty((topk, 4), device=dev)
    os_ = torch.empty((topk,), device=dev)
    oi = torch.empty((topk,), dtype=torch.int32, device=dev)
    oc = torch.zeros((1,), dtype=torch.int32, device=dev)
    bd, sdv, vd = boxes.cuda().contiguous(), scores.cuda(), valid.to(torch.int32).cuda()   # keep alive across the call
    rc = _lib.load().apse_nms_rank(_lib.ptr(bd), _lib.ptr(sdv), _lib.ptr(vd), n, cat_div, cat_mod, ncat, thr, topk,
                                   _lib.ptr(ob), _lib.ptr(os_), _lib.ptr(oi), _lib.ptr(oc), _lib.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    cnt = int(oc.cpu()[0])
    got = oi.cpu().numpy()[:cnt]
    _log(logdir, "nms/%d" % seed, dict(ref=len(ref_idx), got=cnt, equal=bool(len(ref_idx) == cnt and (got == ref_idx).all())))
    assert cnt == len(ref_idx)
    assert (got == ref_idx).all()
    assert torch.equal(ob.cpu()[:cnt], boxes[torch.from_numpy(ref_idx.astype(np.int64))])


def test_mask_utils_golden(golden_dir, logdir):
    """get_mask_centroid / compute_closest_point on the GPU vs vectors from the reference's own functions."""
    import sys
    sys.path.insert(0, golden_dir)
    from make_golden import make_mask
    from apse_uav_amd.utils import mask_utils
    from oracle import mask_utils as omu
    with open(os.path.join(golden_dir, "mask_utils_golden.json")) as f:
        gold = json.load(f)
    H, W = gold["height"], gold["width"]
    for c in gold["cases"]:
        m = torch.from_numpy(make_mask(H, W, c["spec"])).cuda()
        cen = mask_utils.get_mask_centroid(m)
        clo = mask_utils.compute_closest_point(m, c["point"])
        _log(logdir, "mask/" + c["name"], dict(cen=cen, ref_cen=c["centroid"], clo=clo, ref_clo=c["closest"]))
        assert list(clo) == c["closest"], c["name"]
        mh = make_mask(H, W, c["spec"])
        exact = omu.get_mask_centroid(mh)
        assert list(cen) == list(exact), c["name"]            # bit-exact vs the integer restatement
        # vs the reference's own f32 result: equal, except where the exact mean is an integer and the
        # reference's order-dependent f32 sum lands just below it (DESIGN.md "Centroid")
        col = mh.sum(axis=0).astype(np.int64); row = mh.sum(axis=1).astype(np.int64); mass = int(mh.sum())
        sx = int((col * (np.arange(W) + 1)).sum()); sy = int((row * (np.arange(H) + 1)).sum())
        for axis, s_ in ((0, sx), (1, sy)):
            if s_ % mass == 0:
                assert cen[axis] - c["centroid"][axis] in (0.0, 1.0), c["name"]
            else:
                assert cen[axis] == c["centroid"][axis], c["name"]


def test_closest_point_word_search_equals_full_scan(logdir):
    """Round 4: for integer targets the closest-point kernels take TWO candidates per 64-pixel word -- the nearest set bit at or left of
    the target column and the nearest one right of it (csrc/mask_tail.hip mt_word_nearest carries the argument why the row's minimiser
    of (f32 distance, row-major index) is one of them in frames up to 4096 x 4096).  Random masks -- blobs, blobs with holes, 30 %
    speckle, word-aligned edges, masks touching the frame border -- and integer targets outside the mask, on interior pixels, on
    boundary pixels, inside holes, and > 4096 px away (f32 distances above 2^24: rounding ties) against the oracle's full row-major
    argmin (compute_closest_point restated, dcnn/utils/mask_utils.py:6-23); one non-integer target takes the every-pixel kernel.
    The context's table kernel (same helper) is held to the oracle by the detector / full-size tests."""
    from apse_uav_amd.utils import mask_utils
    from oracle import mask_utils as omu
    H, W = 2160, 3840
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:H, 0:W]
    checked = 0
    for case in range(6):
        m = np.zeros((H, W), bool)
        if case < 4:
            for _ in range(int(rng.integers(1, 5))):
                cx, cy = rng.integers(0, W), rng.integers(0, H)
                rx, ry = rng.integers(20, 400), rng.integers(20, 300)
                m |= ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0
            if case >= 2:                                        # holes
                for _ in range(6):
                    ys, xs = np.nonzero(m)
                    k = rng.integers(0, len(ys))
                    m[max(ys[k] - 9, 0):ys[k] + 9, max(xs[k] - 14, 0):xs[k] + 14] = False
        elif case == 4:                                          # speckle inside a word-aligned rectangle
            m[700:1100, 1280:1920] = rng.random((400, 640)) < 0.3
        else:                                                    # far corner blob: distances > 2^24 from targets near (1, 1)
            m |= ((xx - 3700) / 111) ** 2 + ((yy - 2040) / 90) ** 2 <= 1.0
            m[2100:2160, 3600:3840] = True
        ys, xs = np.nonzero(m)
        x0, y0, x1, y1 = int(xs.min()), int(ys.min()), int(xs.max()) + 1, int(ys.max()) + 1
        win, rect = m[y0:y1, x0:x1], (x0, y0, x1, y1)
        md = torch.from_numpy(m).cuda()
        targets = [(1.0, 1.0), (3840.0, 2160.0), (3.0, 5.0), (float(W // 2), float(H // 2))]
        for _ in range(10):
            k = rng.integers(0, len(ys))
            targets.append((float(xs[k] + 1), float(ys[k] + 1)))                            # on the mask (interior or boundary)
            targets.append((float(rng.integers(1, W + 1)), float(rng.integers(1, H + 1))))  # anywhere
        holes = np.argwhere(~win)
        for k in rng.integers(0, max(len(holes), 1), 4 if len(holes) else 0):
            targets.append((float(holes[k][1] + 1 + x0), float(holes[k][0] + 1 + y0)))      # non-mask pixel inside the window (holes)
        targets.append((1911.5, 966.25))                                                    # not a pixel position: full-scan kernel
        for t in targets:
            got = mask_utils.compute_closest_point(md, t)
            ref = omu.window_closest_point(win, rect, t)
            assert got == ref, (case, t, got, ref)
            checked += 1
    _log(logdir, "closest_wordwise_vs_full", dict(cases=6, targets=checked))


def test_association_head_golden(golden_dir, logdir):
    from apse_uav_amd.networks.association_head import AssociationHead
    import sys
    sys.path.insert(0, golden_dir)
    from make_golden import formula_tensor
    g = np.load(os.path.join(golden_dir, "association_head_golden.npz"))
    head = AssociationHead(10, 256)
    head.load_state_dict({"fc.weight": formula_tensor((128, 25600), 131, 71, 257, 8192.0),
                          "fc.bias": formula_tensor((128,), 17, 5, 61, 64.0)})
    x = formula_tensor((3, 256, 10, 10), 37, 11, 509, 97.0)
    x[1] = torch.relu(x[1])
    x[2] = 0.0
    y = head(x.cuda()).cpu().numpy()
    d = float(np.abs(y - g["full_out"]).max())
    small = AssociationHead(10, 8)
    small.load_state_dict({"fc.weight": torch.from_numpy(g["small_w"]), "fc.bias": torch.from_numpy(g["small_b"])})
    ys = small(torch.from_numpy(g["small_x"]).cuda()).cpu().numpy()
    d2 = float(np.abs(ys - g["small_out"]).max())
    _log(logdir, "assoc_head", dict(full=d, small=d2))
    assert d < 2e-5 and d2 < 2e-6            # unit vectors, f32: accumulation-order noise (K = 25600 / 800)


def test_sqdist_and_normalize(logdir):
    from oracle import tracker as otr
    from apse_uav_amd import _lib
    g = torch.Generator().manual_seed(9)
    a = torch.nn.functional.normalize(torch.randn(7, 128, generator=g), dim=1)
    b = torch.nn.functional.normalize(torch.randn(5, 128, generator=g), dim=1)
    ref = otr.distance_matrix([a[i] for i in range(7)], b)
    out = torch.empty((7, 5), device="cuda")
    ad, bd = a.cuda(), b.cuda()
    assert _lib.load().apse_sqdist(_lib.ptr(ad), _lib.ptr(bd), 7, 5, 128, _lib.ptr(out), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert float((out.cpu() - ref).abs().max()) < 1e-6


def test_undistort_gamma(golden_dir, logdir):
    """preprocess_img on the GPU vs oracle/preproc.py at 3840x2160: the fixed-point undistort AND the Lab gamma are integer
    arithmetic on tables (round 3), so every byte must be equal -- stand-alone operator and on saturated random colours too.
    Parity of either side with OpenCV itself: unpinned (cv2 absent; visualize_uav.py:62-69)."""
    from oracle import preproc as op
    from apse_uav_amd.utils.preprocess import FramePreprocessor
    with open(os.path.join(golden_dir, "cam_params.json")) as f:
        cam = json.load(f)
    from apse_uav_amd.synthetic import SyntheticSequence
    frame = SyntheticSequence("static", 2160, 3840).frame(0)
    rng = np.random.default_rng(0)
    frame[500:700, 900:1300] = rng.integers(0, 256, (200, 400, 3), dtype=np.uint8)      # high-frequency patch, every colour octant
    d = torch.from_numpy(frame).cuda()[None]
    und = FramePreprocessor(cam, undistort=True, gamma_correct=False)(d)[0].cpu().numpy()
    ref_u = op.undistort(frame, cam["mtx"], cam["dist"])
    nbad = int((und != ref_u).sum())
    full = FramePreprocessor(cam)(d)[0].cpu().numpy()
    ref_f = op.lab_gamma(ref_u, op.gamma_lut())
    gbad = int((full != ref_f).sum())
    only_gamma = FramePreprocessor(cam, undistort=False, gamma_correct=True)(d)[0].cpu().numpy()
    gbad2 = int((only_gamma != op.lab_gamma(frame, op.gamma_lut())).sum())
    # all 2^24 colours through the Lab step (a 4096 x 4096 image)
    v = np.arange(1 << 24, dtype=np.uint32)
    cube = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], -1).astype(np.uint8).reshape(4096, 4096, 3)
    got_cube = FramePreprocessor(cam, undistort=False, gamma_correct=True)(torch.from_numpy(cube).cuda()[None])[0].cpu().numpy()
    cbad = int((got_cube != op.lab_gamma(cube, op.gamma_lut())).sum())
    _log(logdir, "preproc", dict(undistort_mismatch=nbad, gamma_mismatch=gbad, gamma_only_mismatch=gbad2, all_colours_mismatch=cbad))
    assert nbad == 0
    assert gbad == 0 and gbad2 == 0 and cbad == 0
    assert (full.astype(np.int32).mean() < frame.astype(np.int32).mean())                  # gamma 2 darkens


def test_c_abi_rejects_bad_arguments():
    """Every export returns a negative APSE_E_* code on bad input (no exception crosses the ABI, nothing is launched)."""
    import ctypes as C
    from apse_uav_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc()
    d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad = 1, 8, 8, 24, 16, 1, 1, 1, 0      # Cin not a power of two
    d.cfg, d.splitk = -1, 0
    x = torch.zeros(1, 8, 8, 32, device="cuda")
    w = torch.zeros(128 * 32, device="cuda")
    y = torch.zeros(1, 8, 8, 16, device="cuda")
    s = _lib.stream_ptr()
    assert lib.apse_conv2d(C.byref(d), _lib.ptr(x), _lib.ptr(w), None, None, _lib.ptr(y), None, 0, s) < 0
    d.Cin = 32
    d.cfg = 11                                                                                   # no such tile shape
    assert lib.apse_conv2d(C.byref(d), _lib.ptr(x), _lib.ptr(w), None, None, _lib.ptr(y), None, 0, s) < 0
    d.cfg, d.splitk, d.Cin = 1, 4, 256                                                           # split-K without a workspace
    x2 = torch.zeros(1, 8, 8, 256, device="cuda")
    w2 = torch.zeros(128 * 256, device="cuda")
    assert lib.apse_conv2d(C.byref(d), _lib.ptr(x2), _lib.ptr(w2), None, None, _lib.ptr(y), None, 0, s) < 0
    d.Cin = 32
    d.cfg, d.splitk, d.res_mode, d.Cout = -1, 0, 1, 18                                           # residual rows need Cout % 4 == 0
    assert lib.apse_conv2d(C.byref(d), _lib.ptr(x), _lib.ptr(w), None, _lib.ptr(y), _lib.ptr(y), None, 0, s) < 0
    d.res_mode, d.Cout, d.prec, d.x_st = 0, 16, 1, 2                                             # f16 storage with bf16 operands
    assert lib.apse_conv2d(C.byref(d), _lib.ptr(x), _lib.ptr(w), None, None, _lib.ptr(y), None, 0, s) < 0
    d.x_st, d.Cin, d.Cout = 0, 64, 128                                                           # 16-bit operands from f32-STORED activations: the
    x3 = torch.zeros(1, 8, 8, 64, device="cuda")                                                 # converting kernel was removed in round 4 -> refused
    w3 = torch.zeros(128 * 64, device="cuda")
    y3 = torch.zeros(1, 8, 8, 128, device="cuda")
    assert lib.apse_conv2d(C.byref(d), _lib.ptr(x3), _lib.ptr(w3), None, None, _lib.ptr(y3), None, 0, s) < 0
    d.prec = 0
    assert lib.apse_conv2d(None, _lib.ptr(x), _lib.ptr(w), None, None, _lib.ptr(y), None, 0, s) < 0
    # context-level calls on a null / unfinished context
    assert lib.apse_backbone(None, 1, s) < 0
    assert lib.apse_roi_features(None, 0, None, None, 1, 8, None, s) < 0
    cfg = _lib.Config()
    cfg.struct_size = 4                                                                          # wrong struct size
    ctx = C.c_void_p()
    assert lib.apse_create(C.byref(cfg), C.byref(ctx)) < 0 and not ctx.value
    assert b"size" in lib.apse_last_error(None)
    cfg.struct_size = C.sizeof(_lib.Config)
    cfg.max_batch, cfg.frame_w, cfg.frame_h, cfg.num_classes, cfg.dets_per_image = 1, 480, 270, 4, 100
    cfg.compute_dtype, cfg.storage16 = 1, 0                                                      # 16-bit matrix cores on f32-stored tensors: removed
    assert lib.apse_create(C.byref(cfg), C.byref(ctx)) < 0 and not ctx.value
    assert b"storage16" in lib.apse_last_error(None)
    torch.cuda.synchronize()
