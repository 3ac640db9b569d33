"""-m gpu: the reference-contract entry points INTEGRATION.md tells a maintainer to bind.

* ``TrackRCNN.inference(batched_inputs, detected_instances)`` (/root/reference/dcnn/networks/track_rcnn.py:16-58, called
  from dcnn/engines/track_predictor.py:50-51) with the reference's model input: a PIL-resized f32 CHW image;
* ``apse_preprocess_images`` (normalise + pad of that input, detectron2 ``preprocess_image``) against the oracle;
* ``apse_forward`` == the five stage calls; ``apse_rpn`` == ``apse_rpn_levels(31)``;
* ``SelectiveMaskRCNN.scan`` (dcnn/networks/selective_rcnn.py:27-84).
Small configuration (one bottleneck per stage), all through the C ABI.
"""
import ctypes as C
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


def _log(logdir, name, obj):
    with open(os.path.join(logdir, "boundary_parity.log"), "a") as f:
        f.write(name + " " + json.dumps(obj) + "\n")


@pytest.fixture(scope="module")
def ctx():
    from PIL import Image
    from apse_uav_amd.engines.track_predictor import TrackPredictor
    from apse_uav_amd.networks.association_head import AssociationHead
    from apse_uav_amd.synthetic import SyntheticSequence
    from apse_uav_amd.utils import resample
    from apse_uav_amd.weights import synthetic_association_state, synthetic_detector_state
    from oracle.detector import DetectorOracle
    sd = synthetic_detector_state(0, BLOCKS)
    asd = synthetic_association_state(1)
    seq = SyntheticSequence("dynamic", *FRAME)
    pr = TrackPredictor(_cfg(), state_dict=sd)
    head = AssociationHead(roi_size=10, input_depth=256)
    head.load_state_dict(asd)
    pr.model.attach_association_head(head)
    ih, iw = resample.resize_shortest_edge(FRAME[0], FRAME[1], 256, 448)
    frame = seq.frame(0)
    img = np.asarray(Image.fromarray(frame).resize((iw, ih), Image.BILINEAR))
    x = torch.as_tensor(img.astype("float32").transpose(2, 0, 1))          # what track_predictor.py:48-50 builds
    oracle = DetectorOracle(sd, dict(depth_blocks=BLOCKS, min_size=256, max_size=448))
    return dict(sd=sd, asd=asd, pr=pr, frame=frame, x=x, oracle=oracle, ih=ih, iw=iw, head=head)


def test_preprocess_images_equals_oracle_preprocess(ctx):
    """apse_preprocess_images / chw_to_nhwc4_norm: (x - PIXEL_MEAN) / 1, zero pad to a multiple of 32, BGR0 NHWC."""
    model = ctx["pr"].model
    model.preprocess_images(ctx["x"].unsqueeze(0).cuda(), FRAME)
    got = model.debug_tensor("input").cpu()
    ref = ctx["oracle"].preprocess(ctx["x"])                              # [1, 3, PH, PW]
    PH, PW = ref.shape[-2:]
    got = got.view(-1, PH, PW, 4)[0]
    assert torch.equal(got[:, :, :3].permute(2, 0, 1), ref[0])
    assert float(got[:, :, 3].abs().max()) == 0.0
    assert ctx["ih"] < PH                                                    # 252 -> 256: the bottom rows are padding
    assert float(got[ctx["ih"]:].abs().max()) == 0.0 and float(got[:, ctx["iw"]:].abs().sum()) == 0.0
    # and it is the tensor the fused u8 path builds (PIL-exact resize + the same normalisation)
    model.preprocess_frames(ctx["pr"]._upload([ctx["frame"]]))
    assert torch.equal(model.debug_tensor("input").cpu().view(-1, PH, PW, 4)[0], got)


def test_inference_contract_vs_frames_path_and_oracle(ctx, logdir):
    from hip_helpers import _live_bytes
    model = ctx["pr"].model
    out, feats = model.inference([{"image": ctx["x"], "height": FRAME[0], "width": FRAME[1]}])
    assert isinstance(out, list) and len(out) == 1 and set(out[0]) == {"instances"}
    inst = out[0]["instances"]
    n, by_images = _live_bytes(model, model.last_results)
    assert list(feats.keys()) == ["p2", "p3", "p4", "p5", "p6"]
    p2 = feats["p2"].cpu()
    insts, feats2 = model.inference_frames(ctx["pr"]._upload([ctx["frame"]]))
    n2, by_frames = _live_bytes(model, model.last_results)
    assert n == n2 > 0 and by_images == by_frames                         # same bytes whichever entry fed the image
    assert torch.equal(p2, feats2["p2"].cpu())
    post = ctx["oracle"].inference(ctx["x"], *FRAME)
    assert n == post["boxes"].shape[0] and torch.equal(inst.pred_classes, post["classes"])
    db = float((inst.pred_boxes.tensor - post["boxes"]).abs().max())
    ds = float((inst.scores - post["scores"]).abs().max())
    df = float((p2 - post["features"]["p2"]).abs().max() / post["features"]["p2"].abs().max())
    _log(logdir, "inference", dict(n=n, box_max_abs=db, score_max_abs=ds, p2_rel=df))
    assert db < 2e-2 and ds < 1e-5 and df < 1e-4
    assert inst.image_size == FRAME and len(inst.pred_masks) == n
    assert tuple(inst.pred_masks[0].dense().shape) == FRAME              # the reference's N x H x W bool masks, on demand


def test_inference_with_detected_instances_vs_oracle_given_boxes(ctx, logdir):
    """track_rcnn.py:52-54: ``detected_instances`` (boxes in resized-image pixels + classes) skip the box branch."""
    from apse_uav_amd.structures.instances import Boxes, Instances
    model = ctx["pr"].model
    boxes = torch.tensor([[40.0, 30.0, 120.0, 90.0], [200.5, 100.25, 260.0, 180.75], [10.0, 150.0, 90.0, 240.0],
                          [300.0, 20.0, 440.0, 60.0]])
    classes = torch.tensor([0, 1, 3, 2])
    det = Instances((ctx["ih"], ctx["iw"]))
    det.pred_boxes = Boxes(boxes)
    det.pred_classes = classes
    out, _ = model.inference([{"image": ctx["x"], "height": FRAME[0], "width": FRAME[1]}], detected_instances=[det])
    inst = out[0]["instances"]
    post = ctx["oracle"].inference(ctx["x"], *FRAME, given_boxes=boxes, given_classes=classes)
    assert len(inst) == 4 == post["boxes"].shape[0]
    assert torch.equal(inst.pred_classes, post["classes"])
    db = float((inst.pred_boxes.tensor - post["boxes"]).abs().max())
    bad = tot = 0
    for k in range(4):
        m = inst.pred_masks[k]
        assert tuple(m.rect) == tuple(post["mask_rects"][k])
        bad += int((m.window().cpu() != post["mask_windows"][k]).sum())
        tot += int(post["mask_windows"][k].sum())
    _log(logdir, "given", dict(box_max_abs=db, mask_px_mismatch=bad, mask_px=tot))
    assert db < 1e-3 and bad <= 4


def test_forward_equals_the_five_stage_calls(ctx):
    """apse_forward = apse_backbone + apse_rpn + apse_box_head + apse_mask_tail + apse_embed on the same input."""
    from apse_uav_amd import _lib
    from hip_helpers import _live_bytes
    lib = _lib.load()
    model = ctx["pr"].model
    dev = ctx["pr"]._upload([ctx["frame"]])
    s = _lib.stream_ptr()
    model.preprocess_frames(dev)
    for fn in ("apse_backbone", "apse_rpn", "apse_box_head", "apse_mask_tail", "apse_embed"):
        _lib.check(getattr(lib, fn)(model._ctx, 1, s), model._ctx, fn)
    n1, staged = _live_bytes(model, model.read(1))
    model.preprocess_frames(dev)
    _lib.check(lib.apse_forward(model._ctx, 1, s), model._ctx, "apse_forward")
    n2, fused = _live_bytes(model, model.read(1))
    assert n1 == n2 > 0 and staged == fused
    props_a = model.debug_tensor("proposals").cpu()
    model.preprocess_frames(dev)
    model.run(1)                                                          # apse_rpn_levels(31) inside
    assert torch.equal(model.debug_tensor("proposals").cpu(), props_a)
    assert lib.apse_forward(model._ctx, 0, s) != 0 and lib.apse_forward(model._ctx, 99, s) != 0      # batch out of range
    assert b"batch" in lib.apse_last_error(model._ctx)


def test_mask_tail_twice_gives_the_same_record(ctx):
    """paste_masks adds into integer sums that only pack_detections used to clear; a staged-API caller repeating
    apse_mask_tail on one detection list (a timing loop) must not get doubled masses / moved centroids."""
    from apse_uav_amd import _lib
    from hip_helpers import _live_bytes
    lib = _lib.load()
    model = ctx["pr"].model
    dev = ctx["pr"]._upload([ctx["frame"]])
    s = _lib.stream_ptr()
    model.preprocess_frames(dev)
    for fn in ("apse_backbone", "apse_rpn", "apse_box_head", "apse_mask_tail", "apse_embed"):
        _lib.check(getattr(lib, fn)(model._ctx, 1, s), model._ctx, fn)
    n1, once = _live_bytes(model, model.read(1))
    mass1 = model.last_results.mass[:n1].copy()
    for _ in range(2):
        _lib.check(lib.apse_mask_tail(model._ctx, 1, s), model._ctx, "apse_mask_tail")
    _lib.check(lib.apse_embed(model._ctx, 1, s), model._ctx, "apse_embed")
    n2, again = _live_bytes(model, model.read(1))
    assert n1 == n2 > 0 and int(mass1.sum()) > 0
    assert np.array_equal(model.last_results.mass[:n2], mass1) and once == again


def test_profile_with_a_forward_between_the_read_halves(ctx):
    """apse_profile(1) together with the two-half read and a forward enqueued between _begin and _end (TrackPredictor
    run-ahead).  The event pool has two halves, so _end accounts exactly the forward it returns: the launch counts of two such
    steps are twice those of one plain profiled step."""
    from apse_uav_amd import _lib
    lib = _lib.load()
    model = ctx["pr"].model
    dev = ctx["pr"]._upload([ctx["frame"]])
    NCFG = 14

    def read_prof():
        pr = (C.c_double * (3 * NCFG))()
        lib.apse_profile_read(model._ctx, C.byref(pr), 1)
        return np.array(list(pr)).reshape(NCFG, 3)

    model.preprocess_frames(dev)
    model.run(1)
    model.read(1)                                      # context built, nothing pending
    lib.apse_profile(model._ctx, 1)
    model.preprocess_frames(dev)
    model.run(1)
    model.read(1)
    one = read_prof()
    model.preprocess_frames(dev)
    model.run(1)
    for k in range(2):
        model.read_begin(1)
        if k == 0:
            model.preprocess_frames(dev)
            model.run(1)                               # the next forward, enqueued between the two halves of this read
        model.read_end(1)
    two = read_prof()
    lib.apse_profile(model._ctx, 0)
    assert one[:, 2].sum() > 10
    assert np.array_equal(two[:, 2], 2 * one[:, 2])    # every launch of both forwards timed once, none dropped
    assert np.allclose(two[:, 1], 2 * one[:, 1])       # and attributed the right FLOPs
    assert (two[:, 0][one[:, 2] > 0] > 0).all()


def test_selective_scan_contract(ctx, logdir):
    """SelectiveMaskRCNN.scan: the reference's model-level entry of the Selective* predictor: proposals from the last
    pyramid level only; returns the post-processed list alone (no feature dict)."""
    from apse_uav_amd.engines.selective_predictor import SelectivePredictor
    sp = SelectivePredictor(_cfg(), state_dict=ctx["sd"])
    out = sp.model.scan([{"image": ctx["x"], "height": FRAME[0], "width": FRAME[1]}])
    assert isinstance(out, list) and set(out[0]) == {"instances"}
    inst = out[0]["instances"]
    P = int(sp.model.last_results.prop_count[0])
    props = sp.model.debug_tensor("proposals").cpu().view(-1, 4)[:P]
    post = ctx["oracle"].inference(ctx["x"], *FRAME, rpn_levels=[4])
    assert P == post["proposals"]["boxes"].shape[0]
    assert float((props - post["proposals"]["boxes"]).abs().max()) < 1e-2
    assert len(inst) == post["boxes"].shape[0]
    via_call = sp(ctx["frame"])["instances"]                              # the engine entry on the u8 frame: same result
    assert len(via_call) == len(inst)
    if len(inst):
        assert torch.equal(via_call.pred_boxes.tensor, inst.pred_boxes.tensor)
        assert float((inst.pred_boxes.tensor - post["boxes"]).abs().max()) < 5e-2
    _log(logdir, "scan", dict(P=P, n=len(inst)))
