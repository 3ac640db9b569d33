"""-m gpu: RoiFeaturesGenerator (HIP backbone + apse_roi_features) against the CPU oracle restatement of
dcnn/engines/roi_features_generator.py:68-117, both branches (roi_pool / masked roi_align).  parity unpinned:
the reference itself needs detectron2 + torchvision + pycocotools (absent); the oracle restates their algorithms."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BLOCKS = (1, 1, 1, 1)
FRAME = (270, 480)


def _cfg():
    from apse_uav_amd.config import setup_cfg
    cfg = setup_cfg()
    cfg.INPUT.MIN_SIZE_TEST = 256
    cfg.INPUT.MAX_SIZE_TEST = 448
    return cfg


@pytest.fixture(scope="module")
def setup():
    from PIL import Image
    from apse_uav_amd.weights import synthetic_detector_state
    from apse_uav_amd.synthetic import SyntheticSequence
    from apse_uav_amd.engines.roi_features_generator import RoiFeaturesGenerator
    from apse_uav_amd.utils import resample
    from oracle.detector import DetectorOracle
    sd = synthetic_detector_state(0, BLOCKS)
    # only backbone tensors, without the prefix: what PartialCheckpointer hands to the bare backbone
    part = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
    gen = RoiFeaturesGenerator(_cfg(), roi_size=8, state_dict=part)
    oracle = DetectorOracle(sd, dict(depth_blocks=BLOCKS, min_size=256, max_size=448))
    ih, iw = resample.resize_shortest_edge(FRAME[0], FRAME[1], 256, 448)
    frame = SyntheticSequence("dynamic", FRAME[0], FRAME[1]).frame(3)
    img = np.asarray(Image.fromarray(frame).resize((iw, ih), Image.BILINEAR))
    x = torch.as_tensor(img.astype("float32").transpose(2, 0, 1))
    # MOT rows: <frame>, <id>, <bb_left>, <bb_top>, <bb_width>, <bb_height>, <conf>; one box leaves the frame, one is tiny
    objects = np.array([[3, 11, 40.0, 30.0, 120.5, 60.25, 1], [3, 7, 200.0, 100.0, 90.0, 150.0, 1], [3, 2, 400.0, 200.0, 120.0, 100.0, 1],
                        [3, 5, 10.2, 250.0, 3.0, 2.0, 1], [3, 9, 0.0, 0.0, 480.0, 270.0, 1]])
    return dict(gen=gen, oracle=oracle, frame=frame, x=x, objects=objects)


def _masks(objects, H, W):
    out = np.zeros((len(objects), H, W), dtype=bool)
    yy, xx = np.mgrid[0:H, 0:W]
    for k, o in enumerate(objects):
        cx, cy, rx, ry = o[2] + o[4] / 2, o[3] + o[5] / 2, max(o[4] / 2, 1.0), max(o[5] / 2, 1.0)
        out[k] = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0
    return out


def _log(logdir, name, obj):
    with open(os.path.join(logdir, "detector_parity.log"), "a") as f:
        f.write(name + " " + json.dumps(obj) + "\n")


def test_roi_pool_branch(setup, logdir):
    from oracle import roi_features as orf
    gen = setup["gen"]
    ids, rois = gen.get_rois_features(setup["frame"], setup["objects"])
    rid, ref = orf.get_rois_features(setup["oracle"], setup["x"], FRAME, setup["objects"], None, 8)
    assert gen.get_features_depth() == 256
    assert tuple(rois.shape) == (5, 256, 8, 8) and rois.is_cuda
    assert torch.equal(ids.cpu(), rid)
    d = float((rois.cpu() - ref).abs().max() / ref.abs().max())
    _log(logdir, "roi_features/pool", {"rel_to_max": d})
    assert d < 2e-5            # f32 features (accumulation-order noise), max-pooling picks the same cells


def test_masked_roi_align_branch(setup, logdir):
    from oracle import roi_features as orf
    from apse_uav_amd.utils import rle
    gen = setup["gen"]
    masks = _masks(setup["objects"], *FRAME)
    # the reference's input format: COCO RLE dicts (one of them as a plain array)
    rles = [rle.encode(m) for m in masks]
    rles[1] = masks[1]
    ids, rois = gen.get_rois_features(setup["frame"], setup["objects"], rles)
    rid, ref = orf.get_rois_features(setup["oracle"], setup["x"], FRAME, setup["objects"], masks, 8)
    assert torch.equal(ids.cpu(), rid)
    d = float((rois.cpu() - ref).abs().max() / ref.abs().max())
    _log(logdir, "roi_features/masked_align", {"rel_to_max": d, "ref_absmax": float(ref.abs().max())})
    assert float(ref.abs().max()) > 0
    assert d < 2e-5            # f32: backbone accumulation order + the 4-tap / 16-sample sums


def test_no_objects(setup):
    ids, rois = setup["gen"].get_rois_features(setup["frame"], np.zeros((0, 7)))
    assert tuple(ids.shape) == (0,) and tuple(rois.shape) == (0, 256, 8, 8)
