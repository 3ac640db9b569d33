"""CPU tests (no GPU): COCO RLE codec, the windowed MOTS writers against the dense oracle, legacy roi_align oracle sanity."""
import numpy as np
import pytest
import torch

from apse_uav_amd.structures.instances import Instances
from apse_uav_amd.structures.window_mask import MaskList
from apse_uav_amd.utils import mots_evaluation as me
from apse_uav_amd.utils import rle
from oracle import mots as omots
from oracle import ops

# the example line the reference quotes for the MOTS txt format (dcnn/utils/mots_evaluation.py:11-22): a data vector
MOTS_EXAMPLE = ("52 1005 1 375 1242 WSV:2d;1O10000O10000O1O100O100O1O100O1000000000000000O100O102N5K00O1O1N2O110OO2O001O1NTga3")


def test_rle_reference_example_roundtrip():
    frame, oid, cls, h, w, s = MOTS_EXAMPLE.split(" ")
    h, w = int(h), int(w)
    counts = rle.string_to_counts(s)
    assert sum(counts) == h * w                       # the runs tile the 375 x 1242 image exactly
    assert rle.counts_to_string(counts) == s          # re-encoding reproduces the reference's string
    m = rle.decode({"size": [h, w], "counts": s})
    assert m.shape == (h, w) and m.sum() == sum(counts[1::2])
    assert rle.encode(m)["counts"].decode() == s
    assert omots.rle_string(m) == s                   # the oracle's scalar restatement agrees


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_rle_random_masks(seed):
    g = np.random.default_rng(seed)
    h, w = int(g.integers(1, 40)), int(g.integers(1, 50))
    m = g.random((h, w)) < (0.1, 0.5, 0.95)[seed]
    if seed == 2:
        m[0, 0] = True                                # first run of zeros has length 0
    enc = rle.encode(m)
    assert enc["counts"].decode() == omots.rle_string(m)
    assert np.array_equal(rle.decode(enc).astype(bool), m)
    assert np.array_equal(rle.decode({"size": [h, w], "counts": rle.counts_from_mask(m)}).astype(bool), m)


def _objects(seed, n, H=90, W=130):
    """n overlapping blobs: dense masks for the oracle + an ObjectInstances-like store of WindowMasks."""
    g = np.random.default_rng(seed)
    dense = np.zeros((n, H, W), dtype=bool)
    rects = []
    for k in range(n):
        x0, y0 = int(g.integers(0, W - 70)), int(g.integers(0, H - 40))
        x1, y1 = x0 + int(g.integers(5, 70)), y0 + int(g.integers(5, 40))
        yy, xx = np.mgrid[y0:y1, x0:x1]
        blob = ((xx - (x0 + x1) / 2) / max((x1 - x0) / 2, 1)) ** 2 + ((yy - (y0 + y1) / 2) / max((y1 - y0) / 2, 1)) ** 2 <= 1.0
        dense[k, y0:y1, x0:x1] = blob
        rects.append((x0, y0, x1, y1))
    classes = g.integers(0, 4, size=n)
    scores = g.random(n).astype(np.float32)
    scores[1 % n] = scores[0]                          # a tie: the reference crops mask i then
    ids = list(range(1, n + 1))
    inst = Instances((H, W))
    inst.pred_classes = torch.as_tensor(classes)
    inst.scores = list(torch.as_tensor(scores))
    inst.ids = ids
    inst.pred_masks = MaskList(me._repack(dense[k][r[1]:r[3], r[0]:r[2]], r, (H, W), "cpu") for k, r in enumerate(rects))
    return inst, {"classes": classes.tolist(), "ids": ids, "scores": scores.tolist(), "masks": dense.copy()}, (H, W)


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_mots_writers_match_dense_oracle(seed):
    inst, objs, size = _objects(seed, 7)
    for k in range(len(inst)):                          # the window repack is lossless
        assert np.array_equal(inst.pred_masks[k].dense().numpy(), objs["masks"][k])
    assert np.array_equal(me.result_image_from_objects(inst, size), omots.result_image(objs, size))
    assert me.file_lines_from_instances(inst, 12, size) == omots.file_lines(objs, 12, size)
    me.crop_overlapping_masks(inst)
    omots.crop_overlapping(objs)
    for k in range(len(inst)):
        got = inst.pred_masks[k]
        assert np.array_equal(got.dense().numpy(), objs["masks"][k]), k
        assert got.mass == int(objs["masks"][k].sum())
        if got.mass:
            ys, xs = np.nonzero(objs["masks"][k])
            assert got.centroid == (float((xs + 1).sum() // got.mass), float((ys + 1).sum() // got.mass))
    # after cropping no two masks overlap, and the writers still agree
    assert (np.sum(objs["masks"], axis=0) <= 1).all()
    assert np.array_equal(me.result_image_from_objects(inst, size), omots.result_image(objs, size))
    assert me.file_lines_from_instances(inst, 13, size) == omots.file_lines(objs, 13, size)


def test_parse_mots_seqmap(tmp_path):
    p = tmp_path / "val.seqmap"
    p.write_text("0002 empty 000000 000232\n0006 empty 000000 000269\n")
    assert me.parse_mots_seqmap(str(p)) == (["0002", "0006"], [233, 270])


def test_oracle_roi_align_legacy_properties():
    g = torch.Generator().manual_seed(0)
    # a constant map gives the constant; a bin whose samples all fall outside the map gives 0
    feat = torch.full((1, 3, 12, 16), 2.5)
    rois = torch.tensor([[0, 2.0, 3.0, 30.0, 20.0], [0, 400.0, 400.0, 500.0, 500.0]])
    out = ops.roi_align_legacy(feat, rois, 4, 0.25, 4)
    assert torch.allclose(out[0], torch.full((3, 4, 4), 2.5))
    assert float(out[1].abs().max()) == 0.0
    # a linear ramp is reproduced at the bin centres (bilinear sampling is exact on affine maps, no half-pixel shift)
    yy, xx = torch.meshgrid(torch.arange(12.0), torch.arange(16.0), indexing="ij")
    ramp = (3 * xx + 5 * yy).view(1, 1, 12, 16)
    roi = torch.tensor([[0, 8.0, 4.0, 40.0, 36.0]])
    out = ops.roi_align_legacy(ramp, roi, 4, 0.25, 4)[0, 0]
    cx = 2.0 + (torch.arange(4.0) + 0.5) * 2.0
    cy = 1.0 + (torch.arange(4.0) + 0.5) * 2.0
    assert torch.allclose(out, 3 * cx[None, :] + 5 * cy[:, None], atol=1e-4)
    # roi smaller than one pixel: size clamps to 1
    small = ops.roi_align_legacy(ramp, torch.tensor([[0, 8.0, 8.0, 8.4, 8.4]]), 2, 1.0, 2)[0, 0]
    assert torch.allclose(small, torch.tensor([[3 * 8.25 + 5 * 8.25, 3 * 8.75 + 5 * 8.25], [3 * 8.25 + 5 * 8.75, 3 * 8.75 + 5 * 8.75]]), atol=1e-4)
