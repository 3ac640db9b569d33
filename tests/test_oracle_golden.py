"""CPU (-m "not gpu"): the oracle against the golden vectors produced by the reference's own Python
(tests/golden/make_golden.py), and against Pillow / scipy where the reference calls those directly."""
import json
import os
import sys

import numpy as np
import torch


def _make_mask():
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import make_mask, formula_tensor
    return make_mask, formula_tensor


def test_mask_utils_vs_reference_vectors(golden_dir):
    from oracle import mask_utils as omu
    make_mask, _ = _make_mask()
    with open(os.path.join(golden_dir, "mask_utils_golden.json")) as f:
        gold = json.load(f)
    H, W = gold["height"], gold["width"]
    n_order_dependent = 0
    for c in gold["cases"]:
        m = make_mask(H, W, c["spec"])
        assert int(m.sum()) == c["mass"]
        assert list(omu.compute_closest_point(m, c["point"])) == c["closest"], c["name"]
        cen = omu.get_mask_centroid(m)
        assert list(omu.compute_closest_point(m, cen)) == c["closest_to_own_centroid"] or cen != tuple(c["centroid"])
        col = m.sum(axis=0).astype(np.int64)
        row = m.sum(axis=1).astype(np.int64)
        sums = (int((col * (np.arange(W) + 1)).sum()), int((row * (np.arange(H) + 1)).sum()))
        for axis in (0, 1):
            if sums[axis] % c["mass"] == 0 and cen[axis] != c["centroid"][axis]:
                # exact mean is an integer: the reference's f32 full-frame sum can land one below it
                assert cen[axis] - c["centroid"][axis] == 1.0, c["name"]
                n_order_dependent += 1
            else:
                assert cen[axis] == c["centroid"][axis], c["name"]
    assert n_order_dependent <= 4


def test_window_helpers_match_dense():
    from oracle import mask_utils as omu
    rng = np.random.default_rng(0)
    m = np.zeros((300, 500), bool)
    m[40:120, 210:330] = rng.random((80, 120)) > 0.3
    rect = (200, 30, 340, 130)
    win = m[rect[1]:rect[3], rect[0]:rect[2]]
    assert omu.window_centroid(win, rect) == omu.get_mask_centroid(m)
    for pt in [(1.0, 1.0), (260.0, 80.0), (499.0, 299.0)]:
        assert omu.window_closest_point(win, rect, pt) == omu.compute_closest_point(m, pt)


def test_association_head_vs_reference_vectors(golden_dir):
    from oracle import tracker as otr
    _, formula_tensor = _make_mask()
    g = np.load(os.path.join(golden_dir, "association_head_golden.npz"))
    w = formula_tensor((128, 25600), 131, 71, 257, 8192.0)
    b = formula_tensor((128,), 17, 5, 61, 64.0)
    x = formula_tensor((3, 256, 10, 10), 37, 11, 509, 97.0)
    x[1] = torch.relu(x[1])
    x[2] = 0.0
    y = otr.association_head(x, w, b).numpy()
    assert np.abs(y - g["full_out"]).max() < 1e-6
    ys = otr.association_head(torch.from_numpy(g["small_x"]), torch.from_numpy(g["small_w"]), torch.from_numpy(g["small_b"]))
    assert np.abs(ys.numpy() - g["small_out"]).max() < 1e-6


def test_distance_matrix_matches_cdist():
    from oracle import tracker as otr
    g = torch.Generator().manual_seed(3)
    a = torch.nn.functional.normalize(torch.randn(6, 128, generator=g), dim=1)
    b = torch.nn.functional.normalize(torch.randn(4, 128, generator=g), dim=1)
    d = otr.distance_matrix([a[i] for i in range(6)], b)
    assert torch.allclose(d, torch.cdist(a, b) ** 2, atol=1e-5)


def test_resize_tables_bit_exact_with_pillow():
    """The reference resizes with PIL (track_predictor.py:48): our table builder + two integer passes
    must reproduce Pillow bit for bit (this is the host-side check of what the HIP kernels implement)."""
    from PIL import Image
    from apse_uav_amd.utils import resample
    rng = np.random.default_rng(1)
    for (h, w, oh, ow) in [(216, 384, 75, 133), (135, 240, 188, 333), (97, 160, 33, 57)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(resample.resize_reference_numpy(img, oh, ow), ref)
    assert resample.resize_shortest_edge(2160, 3840) == (750, 1333)
    hb, hc, hk = resample.precompute_coeffs(3840, 1333)
    assert hk == 7 and hb[:, 1].max() <= 7 and int(hc.sum(axis=1).min()) > (1 << 22) - 8


def test_oracle_nms_matches_bruteforce():
    from oracle import ops
    g = torch.Generator().manual_seed(0)
    n = 300
    xy = torch.rand(n, 2, generator=g) * 200
    wh = torch.rand(n, 2, generator=g) * 60 + 1
    boxes = torch.cat([xy, xy + wh], dim=1)
    scores = torch.rand(n, generator=g)
    keep = ops.nms(boxes, scores, 0.5)
    # brute force greedy
    order = torch.argsort(scores, descending=True, stable=True).tolist()
    alive, ref = set(order), []
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    for i in order:
        if i not in alive:
            continue
        ref.append(i)
        for j in list(alive):
            if j == i:
                continue
            lt = torch.max(boxes[i, :2], boxes[j, :2])
            rb = torch.min(boxes[i, 2:], boxes[j, 2:])
            inter = (rb - lt).clamp(min=0).prod()
            if inter / (area[i] + area[j] - inter) > 0.5 and scores[j] <= scores[i]:
                alive.discard(j)
    assert keep.tolist() == ref


def test_oracle_roi_align_constant_and_linear():
    """ROIAlignV2 of a constant map is the constant; of a linear ramp it is the ramp at the bin centre."""
    from oracle import ops
    feat = torch.full((3, 20, 30), 2.5)
    rois = torch.tensor([[4.0, 4.0, 60.0, 40.0], [10.0, 8.0, 13.0, 9.5]])
    out = ops.roi_align_v2(feat, rois, 0.25, 7)
    assert torch.allclose(out, torch.full_like(out, 2.5), atol=1e-6)
    ramp = torch.arange(30, dtype=torch.float32).view(1, 1, 30).expand(1, 20, 30).contiguous()
    out = ops.roi_align_v2(ramp, torch.tensor([[8.0, 8.0, 64.0, 40.0]]), 0.25, 7)
    x0, bw = 8.0 * 0.25 - 0.5, (64.0 - 8.0) * 0.25 / 7
    expect = torch.tensor([x0 + (k + 0.5) * bw for k in range(7)])
    assert torch.allclose(out[0, 0, 3], expect, atol=1e-5)


def test_oracle_roi_pool_simple():
    from oracle import ops
    feat = torch.arange(2 * 12 * 16, dtype=torch.float32).view(1, 2, 12, 16)
    out = ops.roi_pool(feat, torch.tensor([[0.0, 0.0, 0.0, 15.0, 11.0]]), 2, 1.0)
    assert out.shape == (1, 2, 2, 2)
    assert out[0, 0].tolist() == [[5 * 16 + 7, 5 * 16 + 15], [11 * 16 + 7, 11 * 16 + 15]]


def test_oracle_small_detector_runs_and_is_deterministic():
    from apse_uav_amd.weights import synthetic_detector_state
    from oracle.detector import DetectorOracle
    sd = synthetic_detector_state(0, (1, 1, 1, 1))
    o = DetectorOracle(sd, dict(depth_blocks=(1, 1, 1, 1)))
    g = torch.Generator().manual_seed(0)
    img = torch.rand(3, 96, 160, generator=g) * 255
    a = o.inference(img, 192, 320)
    b = o.inference(img, 192, 320)
    assert a["features"]["p2"].shape == (1, 256, 24, 40)
    assert torch.equal(a["boxes"], b["boxes"]) and len(a["mask_windows"]) == a["boxes"].shape[0]
    assert a["proposals"]["boxes"].shape[0] <= 1000
