"""-m gpu: the whole per-frame path (HIP, through the reference-shaped engines) against the CPU oracle.

Small configuration (one bottleneck per stage, 270x480 frames resized to 252x448) so the oracle
finishes in seconds; the full R-101 / 4K configuration is covered by test_gpu_fullsize.py.
Bars: indices / ids / classes exact; float tensors within the f32 tolerance stated per check.
"""
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
    with open(os.path.join(logdir, "detector_parity.log"), "a") as f:
        f.write(name + " " + json.dumps(obj) + "\n")


@pytest.fixture(scope="module")
def setup():
    from PIL import Image
    from apse_uav_amd.weights import synthetic_detector_state, synthetic_association_state
    from apse_uav_amd.synthetic import SyntheticSequence
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from apse_uav_amd.utils import resample
    from oracle.detector import DetectorOracle
    from oracle import tracker as otr
    sd = synthetic_detector_state(0, BLOCKS)
    asd = synthetic_association_state(1)
    seq = SyntheticSequence("dynamic", FRAME[0], FRAME[1])
    cfg = _cfg()
    tracker = RcnnTracker(cfg, FRAME, asd, detector_state=sd)
    oracle = DetectorOracle(sd, dict(depth_blocks=BLOCKS, min_size=256, max_size=448))
    ih, iw = resample.resize_shortest_edge(FRAME[0], FRAME[1], 256, 448)

    def oracle_frame(frame):
        img = np.asarray(Image.fromarray(frame).resize((iw, ih), Image.BILINEAR))
        x = torch.as_tensor(img.astype("float32").transpose(2, 0, 1))
        post = oracle.inference(x, FRAME[0], FRAME[1])
        p2 = post["features"]["p2"]
        rois = otr.features_rois(p2, post["boxes"], FRAME[1])
        emb = otr.association_head(rois, asd["fc.weight"], asd["fc.bias"])
        post["emb"] = emb
        post["assoc_rois"] = rois
        return post
    return dict(sd=sd, asd=asd, seq=seq, cfg=cfg, tracker=tracker, oracle=oracle, oracle_frame=oracle_frame, ih=ih, iw=iw)


def test_single_frame_stages(setup, logdir):
    from oracle import mask_utils as omu
    tr = setup["tracker"]
    frame = setup["seq"].frame(0)
    post = setup["oracle_frame"](frame)
    pred, feats = tr.predictor(frame)
    inst = pred["instances"]
    model = tr.predictor.model
    # ---- features (f32, rel. to max; accumulation-order noise through ~20 convolutions)
    for k in ("p2", "p3", "p4", "p5", "p6"):
        got = feats[k].cpu()
        ref = post["features"][k]
        d = float((got - ref).abs().max() / ref.abs().max())
        _log(logdir, "feat/" + k, dict(rel=d, shape=list(got.shape)))
        assert got.shape == ref.shape
        assert d < 5e-6, (k, d)               # [observed on MI355X, round 2: 2.3e-6 (p2), 6.5e-7 .. 8.1e-7 (p3..p6)]
    # ---- proposals
    res = model.last_results
    P = int(res.prop_count[0])
    props = model.debug_tensor("proposals").cpu().view(-1, 4)[:P]
    ref_props = post["proposals"]["boxes"]
    _log(logdir, "rpn", dict(P=P, ref_P=int(ref_props.shape[0])))
    assert P == ref_props.shape[0]
    dprop = float((props - ref_props).abs().max())
    _log(logdir, "rpn_boxes", dict(max_abs=dprop))
    assert dprop < 1.3e-4, dprop              # [observed 6.1e-5 px = 1 ulp at x ~ 400] pixels in the 252x448 image; same anchors selected in the same order
    # ---- detections
    n = len(inst)
    _log(logdir, "dets", dict(n=n, ref_n=int(post["boxes"].shape[0]), scores=[float(s) for s in inst.scores[:8]],
                              ref_scores=[float(s) for s in post["scores"][:8]]))
    assert n == post["boxes"].shape[0]
    assert torch.equal(inst.pred_classes, post["classes"])
    dbox = float((inst.pred_boxes.tensor - post["boxes"]).abs().max())
    dscore = float((inst.scores - post["scores"]).abs().max())
    _log(logdir, "dets_delta", dict(box_max_abs_px=dbox, score_max_abs=dscore))
    assert dbox < 1.3e-4, dbox                 # [observed 6.1e-5 frame px = 1 ulp at x ~ 400]; north_star: 1e-3 on pixel positions
    assert dscore < 2e-6, dscore               # [observed 9.5e-7]
    # ---- masks: identical pixel sets up to threshold-edge pixels; centroids equal or off by one
    bad_px = 0
    for k in range(n):
        m = inst.pred_masks[k]
        win = m.window().cpu()
        ref_win, ref_rect = post["mask_windows"][k], post["mask_rects"][k]
        assert tuple(m.rect) == tuple(ref_rect)
        bad_px += int((win != ref_win).sum())
        rc = omu.window_centroid(ref_win, ref_rect)
        if m.mass and not np.isnan(rc[0]):
            assert abs(m.centroid[0] - rc[0]) <= 1 and abs(m.centroid[1] - rc[1]) <= 1
    _log(logdir, "masks", dict(mismatched_pixels=bad_px, total=int(sum(int(m.mass) for m in inst.pred_masks))))
    assert bad_px <= 2                         # [observed 0 of 7505] >= 0.5 threshold on f32 bilinear values: an edge pixel may flip
    # ---- embeddings (unit vectors)
    if n:
        emb = torch.from_numpy(inst._record["embeddings"])
        de = float((emb - post["emb"]).abs().max())
        _log(logdir, "emb", dict(max_abs=de))
        assert de < 1e-6                       # [observed 4.9e-7] unit vectors


def test_sequence_ids_and_csv(setup, logdir, tmp_path):
    """Tracker over a short dynamic sequence: ids per frame and the CSV text must equal the oracle's."""
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from apse_uav_amd.utils import csv_log
    from oracle import tracker as otr
    tr = RcnnTracker(setup["cfg"], FRAME, setup["asd"], detector_state=setup["sd"])
    otk = otr.TrackerOracle()
    lines, olines = [], []
    host = 1
    for t in range(6):
        frame = setup["seq"].frame(t)
        rec = tr.next_frame(frame)
        post = setup["oracle_frame"](frame)
        det = dict(boxes=post["boxes"], scores=post["scores"], classes=post["classes"],
                   masks=list(zip(post["mask_windows"], post["mask_rects"])), emb=post["emb"])
        orec = otk.next_frame(det)
        _log(logdir, "seq/%d" % t, dict(ids=list(rec.ids) if len(rec) else [], ref_ids=orec["ids"]))
        assert (list(rec.ids) if len(rec) else []) == orec["ids"]
        line, hi = tr.log_line(rec, host, t)
        oline, ohi = otr.log_oneline(orec, host, t)
        lines.append(line)
        olines.append(oline)
    same = sum(1 for a, b in zip(lines, olines) if a == b)
    _log(logdir, "seq/csv", dict(same_lines=same, n=len(lines), sample=lines[0][:120], ref=olines[0][:120]))
    assert same == len(lines)                  # [observed 6 of 6] integer cells: every line equal to the oracle's text


@pytest.mark.parametrize("depth,want_masks", [(3, False), (2, True)])
def test_pipelined_tracker_equals_sequential(setup, logdir, depth, want_masks):
    """PipelinedRcnnTracker (several frames in flight on separate streams / contexts, association on the host in
    frame order) must give the ids, boxes, centroids and CSV lines of RcnnTracker.next_frame, frame by frame."""
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from apse_uav_amd.engines.pipelined_tracker import PipelinedRcnnTracker
    frames = [setup["seq"].frame(t) for t in range(7)]
    seq = RcnnTracker(setup["cfg"], FRAME, setup["asd"], detector_state=setup["sd"])

    def boxes_np(rec):
        if not len(rec):
            return None
        return np.stack([np.asarray((b.tensor if hasattr(b, "tensor") else torch.as_tensor(b)).cpu(), np.float32).reshape(-1)
                         for b in rec.pred_boxes])
    ref = []
    for t, fr in enumerate(frames):
        rec = seq.next_frame(fr)
        ref.append((list(rec.ids) if len(rec) else [], boxes_np(rec),
                    seq.log_line(rec, 1, t)[0], [m.dense().cpu().numpy() for m in rec.pred_masks] if len(rec) else []))
    drv = PipelinedRcnnTracker(setup["cfg"], FRAME, setup["asd"], depth=depth, want_masks=want_masks, detector_state=setup["sd"])
    n = 0
    for (t, rec), (ids, boxes, line, masks) in zip(drv.run(frames), ref):     # log_line belongs to the frame just returned
        assert t == n
        n += 1
        assert (list(rec.ids) if len(rec) else []) == ids
        if boxes is not None:
            assert np.array_equal(boxes_np(rec), boxes)
        assert drv.tracker.log_line(rec, 1, t)[0] == line
        if want_masks:
            for a, b in zip(rec.pred_masks, masks):
                assert np.array_equal(a.dense().cpu().numpy(), b)
    assert n == len(frames)
    _log(logdir, "pipelined/depth%d" % depth, dict(frames=len(frames), ids_last=ref[-1][0]))


@pytest.mark.parametrize("storage", [True], ids=["store16"])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_bf16_mode_vs_bf16_oracle(setup, logdir, dtype, storage):
    """cfg.APSE.DTYPE = "bf16": bf16 matrix cores, f32 accumulate / storage.  Checked against the oracle run
    with the same quantisation points (filters and layer inputs rounded to bf16).  A different f32
    accumulation order can flip the bf16 rounding of a next-layer input (2^-9 relative), so the float
    tolerance is wider than in f32 mode; detections are compared after matching by IoU."""
    from PIL import Image
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from oracle.detector import DetectorOracle
    cfg = _cfg()
    cfg.APSE.DTYPE = dtype
    cfg.APSE.STORAGE16 = storage
    tr = RcnnTracker(cfg, FRAME, setup["asd"], detector_state=setup["sd"])
    frame = setup["seq"].frame(0)
    pred, feats = tr.predictor(frame)
    inst = pred["instances"]
    oracle = DetectorOracle(setup["sd"], dict(depth_blocks=BLOCKS, min_size=256, max_size=448,
                                              bf16=("f16" if dtype == "f16" else True), storage16=storage))
    img = np.asarray(Image.fromarray(frame).resize((setup["iw"], setup["ih"]), Image.BILINEAR))
    post = oracle.inference(torch.as_tensor(img.astype("float32").transpose(2, 0, 1)), FRAME[0], FRAME[1])
    for k in ("p2", "p4", "p6"):
        got, ref = feats[k].cpu(), post["features"][k]
        d = float((got - ref).abs().max() / ref.abs().max())
        mean = float((got - ref).abs().mean() / ref.abs().mean())
        _log(logdir, dtype + ("/s16" if storage else "/s32") + "/feat/" + k, dict(rel_max=d, rel_mean=mean))
        assert d < 3e-2 and mean < 1e-2        # bf16 noise floor: ~2^-9 after the roundings decorrelate
    n, rn = len(inst), int(post["boxes"].shape[0])
    from hip_helpers import explain_frame
    rep, unexplained = explain_frame(tr.predictor.model, post)
    tag = dtype + ("/s16" if storage else "/s32")
    _log(logdir, tag + "/dets", dict(n=n, ref_n=rn, scores=[round(float(s), 4) for s in inst.scores],
                                     ref=[round(float(s), 4) for s in post["scores"]], matched=rep["box"]["matched"],
                                     only=rep["box"]["only"], rpn_only=len(rep["rpn"]["only"]), unexplained=unexplained,
                                     noise_q999_logit=rep["box"]["score_noise_q999"], noise_rms_logit=rep["box"]["score_noise_rms"]))
    # like the 4K test: the two runs keep the same detections except for candidates that sit inside the measured 16-bit noise
    # (99.9th percentile of the other candidates' deviations) of a discrete decision; every such case is proven one by one
    assert not unexplained, unexplained
    assert rep["box"]["matched"] >= 1 and rep["box"]["matched"] >= min(n, rn) - len(rep["box"]["only"])
    # [observed round 2: 6 of 6 kept in all four variants, scores within 2.3e-3 (bf16) / 4e-4 (f16)]
    assert rep["box"]["matched_score_max_abs"] < (5e-3 if dtype == "bf16" else 1e-3)


def test_no_detections_and_full_list(setup, logdir):
    """Edge cases of the packed detection list: zero detections (threshold never met) and the 100-detection
    cap (threshold ~0): both must run, agree with the oracle on the count and keep the tracker/CSV logic sane."""
    from PIL import Image
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from apse_uav_amd.utils import csv_log
    from oracle.detector import DetectorOracle
    frame = setup["seq"].frame(0)
    img = np.asarray(Image.fromarray(frame).resize((setup["iw"], setup["ih"]), Image.BILINEAR))
    x = torch.as_tensor(img.astype("float32").transpose(2, 0, 1))
    for thr, expect in ((0.9999, "zero"), (0.02, "cap")):
        cfg = _cfg()
        cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = thr
        tr = RcnnTracker(cfg, FRAME, setup["asd"], detector_state=setup["sd"])
        objs = tr.next_frame(frame)
        post = DetectorOracle(setup["sd"], dict(depth_blocks=BLOCKS, min_size=256, max_size=448, score_thresh=thr)).inference(
            x, FRAME[0], FRAME[1])
        n, rn = len(objs), int(post["boxes"].shape[0])
        _log(logdir, "edge/" + expect, dict(n=n, ref_n=rn))
        assert n == rn
        if expect == "zero":
            assert n == 0 and tr.log_line(objs, 1, 0) == ("", 0)
            objs2 = tr.next_frame(frame)                       # still nothing tracked, ids untouched
            assert len(objs2) == 0 and tr.objects.get_new_id() == 1
        else:
            assert n == 100 and list(objs.ids) == list(range(1, 101))
            assert torch.equal(torch.stack([c for c in objs.pred_classes]), post["classes"])
            got = torch.stack([b.tensor[0] for b in objs.pred_boxes])
            assert float((got - post["boxes"]).abs().max()) < 5e-2
            line, hi = tr.log_line(objs, 1, 0)
            assert hi == 100 and len(line.split(",")) == 1 + 4 * 100
            objs2 = tr.next_frame(frame)                       # same frame again: every object re-associated
            assert sorted(objs2.ids) == list(range(1, 101))


def test_selective_predictor_last_level_only(setup, logdir):
    """SelectivePredictor: proposals from p6 only, then the standard ROI heads."""
    from PIL import Image
    from apse_uav_amd.engines.selective_predictor import SelectivePredictor
    from oracle.detector import DetectorOracle
    frame = setup["seq"].frame(0)
    pr = SelectivePredictor(_cfg(), state_dict=setup["sd"])
    inst = pr(frame)["instances"]
    P = int(pr.model.last_results.prop_count[0])
    oracle = DetectorOracle(setup["sd"], dict(depth_blocks=BLOCKS, min_size=256, max_size=448))
    img = np.asarray(Image.fromarray(frame).resize((setup["iw"], setup["ih"]), Image.BILINEAR))
    post = oracle.inference(torch.as_tensor(img.astype("float32").transpose(2, 0, 1)), FRAME[0], FRAME[1], rpn_levels=[4])
    ref_props = post["proposals"]["boxes"]
    props = pr.model.debug_tensor("proposals").cpu().view(-1, 4)[:P]
    _log(logdir, "selective", dict(P=P, ref_P=int(ref_props.shape[0]), n=len(inst), ref_n=int(post["boxes"].shape[0])))
    assert P == ref_props.shape[0] and P <= 4 * 7 * 3
    assert float((props - ref_props).abs().max()) < 1e-2
    assert len(inst) == post["boxes"].shape[0]
    if len(inst):
        assert float((inst.pred_boxes.tensor - post["boxes"]).abs().max()) < 5e-2


def test_sharded_replay_equals_sequential(setup, logdir, tmp_path):
    """BASELINE config 4 semantics on one GPU: detect two contiguous frame shards independently (as two
    ranks would), concatenate their records (what the gather delivers) and replay the association: ids
    and CSV lines must equal the frame-by-frame tracker run.  The collective itself is covered by the
    2-rank gloo test."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import run_sequence as rs
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from apse_uav_amd.sharding import pack_record, shard_frames, unpack_record
    from apse_uav_amd.utils import csv_log
    n = 6
    seq = setup["seq"]
    seq_tr = RcnnTracker(_cfg(), FRAME, setup["asd"], detector_state=setup["sd"])
    ref_lines = []
    for t in range(n):
        objs = seq_tr.next_frame(seq.frame(t))
        ref_lines.append(seq_tr.log_line(objs, 1, t)[0])
    recs = []
    for rank in range(2):
        lo, hi = shard_frames(n, rank, 2)
        w = RcnnTracker(_cfg(), FRAME, setup["asd"], detector_state=setup["sd"])
        part = rs.detect_range(w, seq.frame, lo, hi, 1)
        recs += [unpack_record(pack_record(r, 100, 128), 100, 128) for r in part]      # through the wire format
    rep = RcnnTracker(_cfg(), FRAME, setup["asd"], detector_state=setup["sd"])
    lines, max_id = rs.replay(rep, recs, 1)
    _log(logdir, "sharded", dict(equal=sum(a == b for a, b in zip(lines, ref_lines)), n=n))
    assert lines == ref_lines
    p = tmp_path / "a.csv"
    csv_log.write_consumer_csv(str(p), lines, 1, [2, 3, 4])
    assert len(csv_log.read_centroid_data(str(p))) == n


def test_config3_like_batch4_bf16_with_preproc(setup, logdir, golden_dir):
    """BASELINE config 3 in miniature: dynamic sequence, batch 4, bf16 matrix cores, undistort + gamma HIP
    pre-processing in front of the resize.  Checked against the oracle fed with oracle/preproc.py's frames and
    run in bf16-emulation mode; detections are matched by position (16-bit noise floor)."""
    from PIL import Image
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from oracle import preproc as op
    from oracle.detector import DetectorOracle
    with open(os.path.join(golden_dir, "cam_params.json")) as f:
        cam = json.load(f)
    s = FRAME[1] / 3840.0                                   # intrinsics of the 4K camera scaled to the test frame
    mtx = np.asarray(cam["mtx"], np.float64)
    mtx[0] *= s
    mtx[1] *= s
    cam_small = dict(mtx=mtx.tolist(), dist=cam["dist"])
    cfg = _cfg()
    cfg.APSE.DTYPE = "bf16"
    cfg.APSE.MAX_BATCH = 4
    tr = RcnnTracker(cfg, FRAME, setup["asd"], detector_state=setup["sd"])
    tr.predictor.set_camera(cam_small)
    frames = [setup["seq"].frame(t) for t in range(4)]
    out = tr.predictor.predict_batch(frames, want_masks=False)[0]
    oracle = DetectorOracle(setup["sd"], dict(depth_blocks=BLOCKS, min_size=256, max_size=448, bf16=True, storage16=True))
    tot = matched = only = 0
    for b in range(4):
        pre = op.preprocess_img(frames[b], cam_small["mtx"], cam_small["dist"])
        img = np.asarray(Image.fromarray(pre).resize((setup["iw"], setup["ih"]), Image.BILINEAR))
        post = oracle.inference(torch.as_tensor(img.astype("float32").transpose(2, 0, 1)), FRAME[0], FRAME[1])
        inst = out[b]["instances"]
        n, rn = len(inst), int(post["boxes"].shape[0])
        from hip_helpers import explain_frame
        rep, unexplained = explain_frame(tr.predictor.model, post, b=b)
        _log(logdir, "config3/img%d" % b, dict(n=n, ref_n=rn, matched=rep["box"]["matched"], only=rep["box"]["only"], unexplained=unexplained,
                                                noise_q999_logit=rep["box"]["score_noise_q999"], noise_rms_logit=rep["box"]["score_noise_rms"],
                                                noise_largest=rep["box"]["score_noise_largest"]))
        assert not unexplained, (b, unexplained)       # every difference of the kept sets sits inside the measured bf16 noise of a decision
        tot += n
        matched += rep["box"]["matched"]
        only += len(rep["box"]["only"])
    _log(logdir, "config3", dict(dets=tot, matched=matched, only_one_run=only))
    assert tot > 0 and matched >= tot - only


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("size", [(256, 448), (270, 480)], ids=["tiles_whole", "tiles_ragged"])
def test_fused_stem_pool_equals_two_kernel_form(setup, dtype, size, monkeypatch):
    """16-bit storage modes run the stem convolution + ReLU + max-pool as one kernel (csrc/stem_pool16.hip): the pooled map and
    everything behind it must be the two-kernel form's bits (APSE_NO_STEM_FUSE, read when a context is built).
    270x480 -> 72x120 pooled cells: the last tile column is half outside the map."""
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from hip_helpers import _live_bytes
    frame = setup["seq"].frame(1)
    got = []
    for unfused in (False, True):
        if unfused:
            monkeypatch.setenv("APSE_NO_STEM_FUSE", "1")
        else:
            monkeypatch.delenv("APSE_NO_STEM_FUSE", raising=False)
        cfg = _cfg()
        cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = size
        cfg.APSE.DTYPE = dtype
        tr = RcnnTracker(cfg, FRAME, setup["asd"], detector_state=setup["sd"])
        tr.predictor(frame)
        model = tr.predictor.model
        got.append((model.debug_tensor("stem").cpu(), model.debug_tensor("p2").cpu(), _live_bytes(model, model.last_results)))
    assert got[0][0].shape == got[1][0].shape and got[0][0].numel() > 0
    assert torch.equal(got[0][0].view(torch.int16), got[1][0].view(torch.int16))
    assert torch.equal(got[0][1].view(torch.int16), got[1][1].view(torch.int16))
    assert got[0][2] == got[1][2]


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("size", [(256, 448), (270, 480), (200, 333)], ids=["tiles_whole", "tiles_ragged", "odd"])
def test_fused_bottleneck_equals_three_kernel_form(setup, dtype, size, monkeypatch, logdir):
    """16-bit storage modes run every res2 bottleneck (conv1 1x1 -> conv2 3x3 -> conv3 1x1 + residual) as ONE kernel with the two
    64-channel intermediates in LDS (csrc/bottleneck16.hip).  Everything behind it -- the output of each res2 block, p2, and every
    live byte of the frame's results -- must be the three-kernel form's bits (APSE_NO_BNECK_FUSE, read when a context is built).
    256x448 -> a 64x112 map (8x16 tiles fit exactly); 270x480 -> 68x120 and 200x333 -> 50x84: ragged last tile rows / columns,
    halo pixels outside the map on every border.  Block 0 takes its residual from the projection shortcut and has 64 input
    channels, blocks 1 / 2 are the 256-channel identity form."""
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from hip_helpers import _live_bytes
    frame = setup["seq"].frame(2)
    got = []
    for unfused in (False, True):
        if unfused:
            monkeypatch.setenv("APSE_NO_BNECK_FUSE", "1")
        else:
            monkeypatch.delenv("APSE_NO_BNECK_FUSE", raising=False)
        cfg = _cfg()
        cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = size
        cfg.APSE.DTYPE = dtype
        tr = RcnnTracker(cfg, FRAME, setup["asd"], detector_state=setup["sd"])
        tr.predictor(frame)
        model = tr.predictor.model
        got.append((model.debug_tensor("res2").cpu(), model.debug_tensor("p2").cpu(), _live_bytes(model, model.last_results)))
    assert got[0][0].shape == got[1][0].shape and got[0][0].numel() > 0
    assert int(got[0][0].view(torch.int16).ne(0).sum()) > 0
    same = torch.equal(got[0][0].view(torch.int16), got[1][0].view(torch.int16))
    _log(logdir, "bneck_fused/%s/%dx%d" % (dtype, size[0], size[1]), dict(res2_equal=same, shape=list(got[0][0].shape)))
    assert same
    assert torch.equal(got[0][1].view(torch.int16), got[1][1].view(torch.int16))
    assert got[0][2] == got[1][2]


def test_assoc_fc_sliced_equals_conv_form(setup, logdir, monkeypatch):
    """The association FC as K slices + ordered reduction + normalise (roi.hip, apse_k_assoc_fc) against the split-K convolution +
    reduce + l2_normalize kernels (APSE_NO_ASSOC_FC, read when a context is built): same f32 products, another summation order."""
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    frame = setup["seq"].frame(2)
    got = []
    for conv_form in (False, True):
        if conv_form:
            monkeypatch.setenv("APSE_NO_ASSOC_FC", "1")
        else:
            monkeypatch.delenv("APSE_NO_ASSOC_FC", raising=False)
        tr = RcnnTracker(_cfg(), FRAME, setup["asd"], detector_state=setup["sd"])
        tr.predictor(frame)
        rec = tr.predictor.model.last_results.record(0)
        got.append((rec["embeddings"].copy(), rec["boxes"].copy()))
    assert got[0][0].shape == got[1][0].shape and got[0][0].shape[0] > 0
    assert np.array_equal(got[0][1], got[1][1])
    d = float(np.abs(got[0][0] - got[1][0]).max())
    _log(logdir, "assoc_fc_sliced_vs_conv", dict(n=int(got[0][0].shape[0]), max_abs=d))
    assert d < 1e-6                                        # unit-norm 128-vectors: a few f32 ulps
    assert np.allclose(np.linalg.norm(got[0][0], axis=1), 1.0, atol=1e-6)


def test_results_independent_of_history(setup, logdir):
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from hip_helpers import history_independence
    tr = RcnnTracker(_cfg(), FRAME, setup["asd"], detector_state=setup["sd"])
    outs = history_independence(tr, setup["seq"].frame(0), (setup["ih"], setup["iw"]))
    _log(logdir, "history", dict(n=[o[0] for o in outs], nbytes=[len(o[1]) for o in outs]))
    assert outs[0][0] > 0
    assert outs[0][1] == outs[1][1] == outs[2][1]


def test_fused_preproc_equals_two_kernel_form(setup, golden_dir):
    """undistort + gamma fused into the horizontal resize staging (apse_set_camera) must build the same network input, byte for
    byte, as the stand-alone apse_undistort_gamma kernel followed by the plain resize -- at the test size and at 3840x2160."""
    from apse_uav_amd.engines.track_predictor import TrackPredictor
    from apse_uav_amd.synthetic import SyntheticSequence
    with open(os.path.join(golden_dir, "cam_params.json")) as f:
        cam = json.load(f)
    for (hw, cfg) in ((FRAME, _cfg()), ((2160, 3840), None)):
        if cfg is None:
            from apse_uav_amd.config import setup_cfg
            cfg = setup_cfg()
        s = hw[1] / 3840.0
        mtx = np.asarray(cam["mtx"], np.float64)
        mtx[0] *= s
        mtx[1] *= s
        cam_s = dict(mtx=mtx.tolist(), dist=cam["dist"])
        frame = SyntheticSequence("dynamic", *hw).frame(3)
        got = []
        for fused in (True, False):
            pr = TrackPredictor(cfg, state_dict=setup["sd"])
            pr.set_camera(cam_s, fused=fused)
            assert (pr.frame_preprocessor is None) == fused
            dev = pr._upload([frame])
            pr.model.preprocess_frames(dev)
            got.append(pr.model.debug_tensor("input").cpu())
        plain = TrackPredictor(cfg, state_dict=setup["sd"])
        plain.model.preprocess_frames(plain._upload([frame]))
        assert torch.equal(got[0], got[1])
        assert not torch.equal(got[0], plain.model.debug_tensor("input").cpu())      # the camera model does change the pixels
    # round 4 (compact remap table, 12-byte gathers, the batch's rows grouped per XCD): a batch of three different frames at the small
    # size (270 rows: not a multiple of the 8-row groups), and a camera whose distortion moves pixels by more than the table's
    # +-1023 px at 4K -- it must fall back to the per-pixel model and still give the stand-alone operator's bytes
    cfg3 = _cfg()
    cfg3.APSE.MAX_BATCH = 3
    frames = [SyntheticSequence("dynamic", *FRAME).frame(t) for t in (0, 9, 31)]
    s = FRAME[1] / 3840.0
    mtx = np.asarray(cam["mtx"], np.float64)
    mtx[0] *= s
    mtx[1] *= s
    got = []
    for fused in (True, False):
        pr = TrackPredictor(cfg3, state_dict=setup["sd"])
        pr.set_camera(dict(mtx=mtx.tolist(), dist=cam["dist"]), fused=fused)
        pr.model.preprocess_frames(pr._upload(frames))
        got.append(pr.model.debug_tensor("input").cpu())
    assert torch.equal(got[0], got[1]) and not torch.equal(got[0][:got[0].numel() // 3], got[0][got[0].numel() // 3:2 * (got[0].numel() // 3)])
    from apse_uav_amd.config import setup_cfg
    wild = dict(mtx=cam["mtx"], dist=[-1.5, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])      # k1 = -1.5: 378 157 pixels come from > 1023 px away, inside the frame
    frame = SyntheticSequence("dynamic", 2160, 3840).frame(3)
    got = []
    for fused in (True, False):
        pr = TrackPredictor(setup_cfg(), state_dict=setup["sd"])
        pr.set_camera(wild, fused=fused)
        pr.model.preprocess_frames(pr._upload([frame]))
        got.append(pr.model.debug_tensor("input").cpu())
    assert torch.equal(got[0], got[1])
