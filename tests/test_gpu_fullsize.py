"""-m gpu: BASELINE.json full-size configuration (3840x2160 frames, R-101-FPN, f32) on the HIP path.

* one frame against the CPU oracle (features, proposals, detections, masks, embeddings);
* size-independent properties: determinism (same frame twice -> identical results block), batch 2 ==
  2 x batch 1 per image, given-boxes mode CSV through the consumer's parser (config 1/2 plumbing).
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
FRAME = (2160, 3840)


def _threads():
    from apse_uav_amd.utils.hostinfo import usable_cpus
    return max(1, min(32, usable_cpus()))          # the box's CPU quota (16), not the 256 cores it shows


def _log(logdir, name, obj):
    with open(os.path.join(logdir, "fullsize_parity.log"), "a") as f:
        f.write(name + " " + json.dumps(obj) + "\n")


@pytest.fixture(scope="module")
def env():
    from apse_uav_amd.config import setup_cfg
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from apse_uav_amd.synthetic import SyntheticSequence
    from apse_uav_amd.weights import UAV4K_R101_CLS_BIAS, synthetic_association_state, synthetic_detector_state
    sd = synthetic_detector_state(0, cls_bias=UAV4K_R101_CLS_BIAS)
    asd = synthetic_association_state(1)
    cfg = setup_cfg()
    cfg.APSE.MAX_BATCH = 2
    tr = RcnnTracker(cfg, FRAME, asd, detector_state=sd)
    seq = SyntheticSequence("dynamic", *FRAME)
    return dict(sd=sd, asd=asd, cfg=cfg, tr=tr, seq=seq)


def test_full_frame_vs_oracle(env, logdir):
    from PIL import Image
    from oracle import tracker as otr
    from oracle.detector import DetectorOracle, resize_shape
    tr = env["tr"]
    frame = env["seq"].frame(0)
    pred, feats = tr.predictor(frame)
    inst = pred["instances"]
    ih, iw = resize_shape(*FRAME)
    assert (ih, iw) == (750, 1333)
    img = np.asarray(Image.fromarray(frame).resize((iw, ih), Image.BILINEAR))
    torch.set_num_threads(_threads())
    post = DetectorOracle(env["sd"]).inference(torch.as_tensor(img.astype("float32").transpose(2, 0, 1)), *FRAME)
    # Bars: each limit is about 2x what this test observes on MI355X (logged below; observations in brackets); index results are
    # exact; the one float that misses north_star's 1e-3 (rescaled box corners, 1.22e-3 px) says so in its assert.
    for k in ("p2", "p4", "p6"):
        got, ref = feats[k].cpu(), post["features"][k]
        d = float((got - ref).abs().max() / ref.abs().max())
        _log(logdir, "feat/" + k, dict(rel=d))
        assert d < 6e-6                                   # [2.8e-6] f32 through 104 convolutions, different sum order
    model = tr.predictor.model
    P = int(model.last_results.prop_count[0])
    ref_props = post["proposals"]["boxes"]
    props = model.debug_tensor("proposals").cpu().view(-1, 4)[:P]
    assert P == ref_props.shape[0]
    row_err = (props - ref_props[:P]).abs().max(dim=1).values
    same_p = int(row_err.lt(1e-3).sum())
    # rows that differ are ORDER swaps of near-tied objectness logits around the rank-1000 cut: as a set the proposals agree
    d2 = torch.cdist(props.double(), ref_props.double(), p=float("inf"))
    set_err = float(torch.maximum(d2.min(dim=1).values.max(), d2.min(dim=0).values.max()))
    _log(logdir, "rpn", dict(P=P, ref_P=int(ref_props.shape[0]), rows_equal=same_p, rows_max_abs_when_equal=float(row_err[row_err < 1e-3].max()),
                             set_max_abs=set_err))
    assert same_p >= P - 12                               # [994 of 1000 rows in place, 6 swapped]
    assert float(row_err[row_err < 1e-3].max()) < 1e-3 and set_err < 1.4e-3    # [6.7e-4 px on coordinates up to 1333: 11 ulp]
    n = len(inst)
    from hip_helpers import explain_frame
    rep, unexplained = explain_frame(model, post)
    db = float((inst.pred_boxes.tensor - post["boxes"]).abs().max()) if n == post["boxes"].shape[0] else -1.0
    ds = float((inst.scores - post["scores"]).abs().max()) if n == post["boxes"].shape[0] else -1.0
    _log(logdir, "dets", dict(n=n, ref_n=int(post["boxes"].shape[0]), box_max_abs_px=db, score_max_abs=ds, analysis=rep,
                              unexplained=unexplained, scores=[round(float(s), 5) for s in inst.scores],
                              ref=[round(float(s), 5) for s in post["scores"]]))
    assert n == post["boxes"].shape[0] and not rep["box"]["only"]         # same detection set: ids / box indices exact
    assert not rep["rpn"]["only"] and not unexplained                     # same proposal set (only the ORDER of near-ties differs)
    assert torch.equal(inst.pred_classes, post["classes"])
    # [1.22e-3 px] box corners in 4K frame pixels = 4.3e-4 px in the resized image x 2.88: 5 f32 ulps at x ~ 3000.  north_star's
    # 1e-3 is met in the resized image the network works in and missed by 0.2e-3 px on the rescaled corners; the pixel
    # positions the CSV carries (integer centroids / closest points) are exact (test_4k_sequence_ids_and_csv_vs_oracle).
    assert db < 2.5e-3, ("box corners differ from the oracle by %.3e frame px; observed 1.22e-3 in round 2, which already MISSES "
                         "north_star's 1e-3 by 0.22e-3 (5 f32 ulps at x ~ 3000); this bar is 2x that observation" % db)
    assert ds < 4e-6                                      # [1.6e-6]
    assert rep["box"]["eps_score"] < 6e-5 and rep["box"]["eps_box_px"] < 2e-3     # [2.6e-5 logits, 9.2e-4 px] every candidate above 0.02, not only the kept ones
    bad = tot = 0
    for k in range(n):
        m = inst.pred_masks[k]
        assert tuple(m.rect) == tuple(post["mask_rects"][k])
        bad += int((m.window().cpu() != post["mask_windows"][k]).sum())
        tot += int(m.mass)
    _log(logdir, "masks", dict(mismatched=bad, total=tot))
    assert bad <= 8                                       # [4 of 130 557] >= 0.5 on f32 bilinear values: edge pixels may flip
    if n:
        rois = otr.features_rois(post["features"]["p2"], post["boxes"], FRAME[1])
        emb = otr.association_head(rois, env["asd"]["fc.weight"], env["asd"]["fc.bias"])
        de = float((torch.from_numpy(inst._record["embeddings"]) - emb).abs().max())
        _log(logdir, "emb", dict(max_abs=de))
        assert de < 1.3e-6                                # [6.1e-7] unit vectors


def test_determinism_and_batch_equivalence(env):
    tr = env["tr"]
    pr = tr.predictor
    f0, f1 = env["seq"].frame(0), env["seq"].frame(5)
    a = pr.predict_batch([f0], want_masks=False)[0][0]["instances"]
    b = pr.predict_batch([f0], want_masks=False)[0][0]["instances"]
    assert torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and np.array_equal(a._record["embeddings"], b._record["embeddings"])
    c = pr.predict_batch([f1], want_masks=False)[0][0]["instances"]
    two = pr.predict_batch([f0, f1], want_masks=False)[0]
    for one, bt in ((a, two[0]["instances"]), (c, two[1]["instances"])):
        assert torch.equal(one.pred_boxes.tensor, bt.pred_boxes.tensor)
        assert torch.equal(one.pred_classes, bt.pred_classes)
        assert np.array_equal(one._record["centroids"], bt._record["centroids"])
        assert np.array_equal(one._record["closest"], bt._record["closest"])
        assert np.allclose(one._record["embeddings"], bt._record["embeddings"], atol=1e-6)


def test_given_boxes_csv_roundtrip(env, tmp_path):
    """Configs 1/2 plumbing: the K synthetic vehicles as given boxes -> tracker -> consumer CSV ->
    aruco_detect.readCentroidData-style parse; centroids must fall inside their boxes."""
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from apse_uav_amd.utils import csv_log, resample
    tr = RcnnTracker(env["cfg"], FRAME, env["asd"], detector_state=env["sd"])
    seq = env["seq"]
    ih, iw = resample.resize_shortest_edge(*FRAME)
    sx, sy = iw / FRAME[1], ih / FRAME[0]
    lines = []
    for t in (0, 1, 2, 22):
        boxes = seq.boxes(t) * np.array([sx, sy, sx, sy], np.float32)
        given = (boxes, np.zeros(len(boxes), np.int32), np.array([len(boxes)], np.int32))
        out = tr.predictor.predict_batch([seq.frame(t)], given=given)[0][0]["instances"]
        assert len(out) == len(boxes)
        tr.frame_count += 1
        objs = tr._finish_frame(out, None)
        line, hi = tr.log_line(objs, 1, t)
        lines.append(line)
        for k in range(len(out)):
            cx, cy = out.pred_masks[k].centroid
            x0, y0, x1, y1 = out.pred_boxes.tensor[k].tolist()
            if out.pred_masks[k].mass:
                assert x0 - 2 <= cx <= x1 + 2 and y0 - 2 <= cy <= y1 + 2
    path = tmp_path / "seq_dcnn_data.csv"
    csv_log.write_consumer_csv(str(path), lines, host_id=1, vehicle_ids=[2, 3, 4])
    data = csv_log.read_centroid_data(str(path))
    assert len(data) == 4 and all(len(r) == 17 for r in data)
    assert data[0][0] == 0 and data[3][0] == 22


def test_4k_sequence_ids_and_csv_vs_oracle(env, logdir):
    """BASELINE config 2 in short form: a 4-frame 4K dynamic sequence through RcnnTracker.next_frame; track
    ids per frame and the CSV line text must equal the CPU oracle's tracker (ids / integer cells exact)."""
    from PIL import Image
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from oracle import tracker as otr
    from oracle.detector import DetectorOracle, resize_shape
    torch.set_num_threads(_threads())
    tr = RcnnTracker(env["cfg"], FRAME, env["asd"], detector_state=env["sd"])
    oracle = DetectorOracle(env["sd"])
    otk = otr.TrackerOracle()
    ih, iw = resize_shape(*FRAME)
    same = 0
    for t in range(4):
        frame = env["seq"].frame(3 * t)
        objs = tr.next_frame(frame)
        line, _ = tr.log_line(objs, 1, t)
        img = np.asarray(Image.fromarray(frame).resize((iw, ih), Image.BILINEAR))
        post = oracle.inference(torch.as_tensor(img.astype("float32").transpose(2, 0, 1)), *FRAME)
        rois = otr.features_rois(post["features"]["p2"], post["boxes"], FRAME[1])
        emb = otr.association_head(rois, env["asd"]["fc.weight"], env["asd"]["fc.bias"])
        orec = otk.next_frame(dict(boxes=post["boxes"], scores=post["scores"], classes=post["classes"],
                                   masks=list(zip(post["mask_windows"], post["mask_rects"])), emb=emb))
        oline, _ = otr.log_oneline(orec, 1, t)
        _log(logdir, "seq4k/%d" % t, dict(ids=list(objs.ids) if len(objs) else [], ref_ids=orec["ids"], same_line=line == oline))
        assert (list(objs.ids) if len(objs) else []) == orec["ids"]
        same += int(line == oline)
    assert same == 4                          # [4 of 4] integer cells: every CSV line equal to the oracle's text


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_full_frame_16bit_vs_oracle(env, logdir, dtype):
    """BASELINE configs 3 / 5 precision at full size: 16-bit matrix cores and 16-bit activation storage (this is where the
    256x128 tile and the deep-K 16-bit shapes are live) against the oracle with the same rounding points
    (operands AND stored tensors rounded to the 16-bit type).  A different f32 accumulation order can flip a 16-bit
    rounding (2^-9 / 2^-11 relative) of a next-layer input, so features are compared on mean error.  Detections: the two
    runs must keep the SAME set (then ids / box indices are exact) except for candidates that sit within the measured
    16-bit noise of a discrete decision -- score 0.5, NMS IoU 0.5 / 0.7, the rank-1000 cuts, the order of two near-tied
    overlapping candidates, floor() of the FPN level of a proposal -- which hip_helpers.explain_frame proves one by one, for
    the RPN and the box stage (noise measured on all the OTHER candidates of the same frame; consequences of an explained
    flip through greedy NMS are followed); any other difference fails the test.
    Parity unpinned like the f32 detector (oracle restates detectron2)."""
    from PIL import Image
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from hip_helpers import explain_frame
    from oracle.detector import DetectorOracle, resize_shape
    cfg = env["cfg"].clone()
    cfg.APSE.MAX_BATCH = 1
    cfg.APSE.DTYPE = dtype
    cfg.APSE.STORAGE16 = True
    tr = RcnnTracker(cfg, FRAME, env["asd"], detector_state=env["sd"])
    frame = env["seq"].frame(0)
    pred, feats = tr.predictor(frame)
    inst = pred["instances"]
    model = tr.predictor.model
    ih, iw = resize_shape(*FRAME)
    img = np.asarray(Image.fromarray(frame).resize((iw, ih), Image.BILINEAR))
    x = torch.as_tensor(img.astype("float32").transpose(2, 0, 1))
    torch.set_num_threads(_threads())
    oracle = DetectorOracle(env["sd"], dict(bf16=("f16" if dtype == "f16" else True), storage16=True))
    post = oracle.inference(x, *FRAME)
    lim_max, lim_mean = (6e-2, 1.5e-2) if dtype == "bf16" else (1e-2, 2.5e-3)
    for k in ("p2", "p4", "p6"):
        got, ref = feats[k].cpu(), post["features"][k]
        d = float((got - ref).abs().max() / ref.abs().max())
        mean = float((got - ref).abs().mean() / ref.abs().mean())
        _log(logdir, dtype + "/feat/" + k, dict(rel_max=d, rel_mean=mean))
        assert d < lim_max and mean < lim_mean
    rep, unexplained = explain_frame(model, post, dump=os.path.join(logdir, "analysis_%s_vs_oracle.json" % dtype))
    _log(logdir, dtype + "/dets", dict(analysis=rep, unexplained=unexplained, scores=[round(float(s), 4) for s in inst.scores],
                                       ref=[round(float(s), 4) for s in post["scores"]]))
    assert not unexplained, unexplained
    assert rep["box"]["matched"] >= 1 and rep["box"]["matched"] >= min(rep["box"]["nA"], rep["box"]["nB"]) - 2
    # For the record (no bar: this compares two QUANTISATIONS, not two implementations): the same analysis against the f32
    # oracle.  Round 2: f16 differs from f32 only inside its noise band; bf16 drops one detection that f32 scores 0.5079 (bf16:
    # 0.489, a shift of 0.019 where the other candidates move by <= 0.007) -- the price of 8-bit mantissas, equally in the oracle.
    post32 = DetectorOracle(env["sd"]).inference(x, *FRAME)
    rep32, un32 = explain_frame(model, post32)
    _log(logdir, dtype + "/dets_vs_f32_oracle", dict(analysis=rep32, unexplained=un32))
    # 16-bit noise of what both runs keep (resized-image pixels; x 2.88 in the 4K frame): logged, bounded loosely
    assert rep["box"]["matched_score_max_abs"] < (2e-2 if dtype == "bf16" else 5e-3)
    assert rep["box"]["matched_box_max_abs"] < (6.0 if dtype == "bf16" else 1.0)


def test_4k_results_independent_of_history(env, logdir):
    """R-101 at 3840x2160: frame A after 0-, 8- and 100-detection forwards -> identical bytes (hip_helpers)."""
    from apse_uav_amd.utils import resample
    from hip_helpers import history_independence
    outs = history_independence(env["tr"], env["seq"].frame(0), resample.resize_shortest_edge(*FRAME))
    _log(logdir, "history4k", dict(n=[o[0] for o in outs], nbytes=[len(o[1]) for o in outs]))
    assert outs[0][0] > 0
    assert outs[0][1] == outs[1][1] == outs[2][1]


def test_4k_shards_and_pipeline_equal_sequential(env, logdir):
    """"single ordered CSV identical to the 1-GPU run".  Six 4K frames through (a) RcnnTracker.next_frame,
    (b) two independent shards [0,3) / [3,6) (fresh contexts, as two ranks would hold) -> wire format -> replay,
    (c) PipelinedRcnnTracker(depth=3): the CSV text must be equal byte for byte, the records bit for bit."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import run_sequence as rs
    from apse_uav_amd.engines.pipelined_tracker import PipelinedRcnnTracker
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from apse_uav_amd.sharding import pack_record, shard_frames, unpack_record
    cfg = env["cfg"].clone()
    cfg.APSE.MAX_BATCH = 1
    n = 6
    frames = [env["seq"].frame(7 * t) for t in range(n)]          # the dynamic sequence moves: detections differ per frame
    seq_tr = RcnnTracker(cfg, FRAME, env["asd"], detector_state=env["sd"])
    ref_lines, ref_recs = [], []
    for t in range(n):
        objs = seq_tr.next_frame(frames[t])
        ref_lines.append(seq_tr.log_line(objs, 1, t)[0])
        ref_recs.append(pack_record(seq_tr._last_record, 100, 128).tobytes())
    del seq_tr
    recs = []
    for rank in range(2):
        lo, hi = shard_frames(n, rank, 2)
        w = RcnnTracker(cfg, FRAME, env["asd"], detector_state=env["sd"])
        recs += [pack_record(r, 100, 128) for r in rs.detect_range(w, lambda k: frames[k], lo, hi, 1)]
        del w
    same_recs = sum(a.tobytes() == b for a, b in zip(recs, ref_recs))
    rep = RcnnTracker(cfg, FRAME, env["asd"], detector_state=env["sd"])
    lines, _ = rs.replay(rep, [unpack_record(r, 100, 128) for r in recs], 1)
    lines_native, _ = rs.replay(rep, [unpack_record(r, 100, 128) for r in recs], 1, fast=True)
    del rep
    drv = PipelinedRcnnTracker(cfg, FRAME, env["asd"], depth=3, detector_state=env["sd"])
    plines = [drv.tracker.log_line(objs, 1, t)[0] for t, objs in drv.run(frames)]
    _log(logdir, "shards4k", dict(records_equal=same_recs, sharded_lines_equal=sum(a == b for a, b in zip(lines, ref_lines)),
                                  pipelined_lines_equal=sum(a == b for a, b in zip(plines, ref_lines)), n=n,
                                  sample=ref_lines[-1][:100]))
    assert any(ref_lines)
    assert same_recs == n
    assert lines == ref_lines and lines_native == ref_lines
    assert plines == ref_lines


def _record_bytes(inst):
    r = inst._record
    return b"".join(np.ascontiguousarray(r[k]).tobytes() for k in ("boxes", "scores", "classes", "centroids", "mass", "rects",
                                                                   "closest", "embeddings"))


def _batch_equals_singles(cfg, env, frames, logdir, tag, camera=None):
    """One context (MAX_BATCH = len(frames)): the batch forward and the per-frame forwards must give the same BYTES per
    image (boxes, scores, classes, centroids, masses, rects, closest-point table, embeddings) -- a frame's results do not
    depend on its batch neighbours -- and therefore the same track ids and CSV text."""
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    tr = RcnnTracker(cfg, FRAME, env["asd"], detector_state=env["sd"])
    if camera is not None:
        tr.predictor.set_camera(camera)
    batch = [o["instances"] for o in tr.predictor.predict_batch(frames, want_masks=False)[0]]
    singles = [tr.predictor.predict_batch([f], want_masks=False)[0][0]["instances"] for f in frames]
    same = [(_record_bytes(a) == _record_bytes(b)) for a, b in zip(batch, singles)]
    lines = []
    for dets in (batch, singles):
        tr.reset_tracker()
        out = []
        for t, d in enumerate(dets):
            tr.frame_count += 1
            objs = tr._finish_frame(d, None, host_replay=True)
            out.append((list(objs.ids) if len(objs) else [], tr.log_line(objs, 1, t)[0]))
        lines.append(out)
    _log(logdir, tag, dict(n=[len(b) for b in batch], bytes_equal=same, ids=[l[0] for l in lines[0]]))
    assert all(len(b) > 0 for b in batch)
    assert all(same)
    assert lines[0] == lines[1]


def test_config3_bf16_batch4_with_preproc_equals_batch1(env, logdir, golden_dir):
    """BASELINE configs[2] at full size: dynamic 3840x2160 frames, batch 4, bf16 matrix cores + storage, undistort + gamma HIP
    pre-processing (data/cam_params.json) in front of the resize."""
    with open(os.path.join(golden_dir, "cam_params.json")) as f:
        cam = json.load(f)
    cfg = env["cfg"].clone()
    cfg.APSE.MAX_BATCH = 4
    cfg.APSE.DTYPE = "bf16"
    cfg.APSE.STORAGE16 = True
    frames = [env["seq"].frame(t) for t in (0, 12, 27, 45)]           # vehicle 1 is out of the picture in the third
    _batch_equals_singles(cfg, env, frames, logdir, "config3_b4_bf16", camera=cam)


def test_config5_f16_batch8_equals_batch1(env, logdir):
    """BASELINE configs[4] on one GPU: synthetic 3840x2160 stream, batch 8 per GPU, fp16."""
    cfg = env["cfg"].clone()
    cfg.APSE.MAX_BATCH = 8
    cfg.APSE.DTYPE = "f16"
    cfg.APSE.STORAGE16 = True
    frames = [env["seq"].frame(6 * t) for t in range(8)]
    _batch_equals_singles(cfg, env, frames, logdir, "config5_b8_f16")


def test_config2_static64_csv_vs_oracle(env, logdir, tmp_path, golden_dir):
    """BASELINE configs[1] as SURVEY 8d restates it: the "static" 64-frame 3840x2160 sequence, batch 1 f32, through
    RcnnTracker.next_frame; CSV text against the oracle's CSV, then the consumer layout against the shipped header.
    The static sequence has zero motion, so its frames are identical arrays: the oracle DETECTOR runs once, the
    oracle TRACKER (ids, association, log line) runs all 64 steps."""
    from PIL import Image
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from apse_uav_amd.synthetic import SyntheticSequence
    from apse_uav_amd.utils import csv_log
    from oracle import tracker as otr
    from oracle.detector import DetectorOracle, resize_shape
    seq = SyntheticSequence("static", *FRAME)
    f0 = seq.frame(0)
    assert np.array_equal(f0, seq.frame(63))
    cfg = env["cfg"].clone()
    cfg.APSE.MAX_BATCH = 1
    tr = RcnnTracker(cfg, FRAME, env["asd"], detector_state=env["sd"])
    lines, max_id = [], 0
    for t in range(64):
        objs = tr.next_frame(seq.frame(t), upcoming=None)
        line, hi = tr.log_line(objs, 1, t)
        lines.append(line)
        max_id = max(max_id, hi)
    torch.set_num_threads(_threads())
    ih, iw = resize_shape(*FRAME)
    img = np.asarray(Image.fromarray(f0).resize((iw, ih), Image.BILINEAR))
    post = DetectorOracle(env["sd"]).inference(torch.as_tensor(img.astype("float32").transpose(2, 0, 1)), *FRAME)
    rois = otr.features_rois(post["features"]["p2"], post["boxes"], FRAME[1])
    emb = otr.association_head(rois, env["asd"]["fc.weight"], env["asd"]["fc.bias"])
    otk = otr.TrackerOracle()
    olines = []
    for t in range(64):
        orec = otk.next_frame(dict(boxes=post["boxes"], scores=post["scores"], classes=post["classes"],
                                   masks=list(zip(post["mask_windows"], post["mask_rects"])), emb=emb))
        olines.append(otr.log_oneline(orec, 1, t)[0])
    same = sum(a == b for a, b in zip(lines, olines))
    _log(logdir, "config2_static64", dict(same_lines=same, n=64, max_id=max_id, sample=lines[63][:96]))
    assert same == 64
    path = tmp_path / "static_dcnn_data.csv"
    csv_log.write_consumer_csv(str(path), lines, host_id=1, vehicle_ids=[2, 3, 4])
    with open(path) as f, open(os.path.join(golden_dir, "static_dcnn_data_head.csv")) as g:
        got, ref = f.read().split("\n"), g.read().split("\n")
    assert got[0].split(",")[0].startswith("Host id:") and len(got[0].split(",")) == len(ref[0].split(","))
    assert [c.split(" ")[-1] for c in got[1].split(",")] == [c.split(" ")[-1] for c in ref[1].split(",")]      # frame, cent_x, cent_y, clos_x, ...
    data = csv_log.read_centroid_data(str(path))
    assert len(data) == 64 and all(len(r) == 17 for r in data) and [r[0] for r in data] == list(range(64))


def _ids_have_gap(ids_per_frame):
    """ids that are present, then absent for at least one frame, then present again (re-association after absence)."""
    out = []
    for i in sorted({i for ids in ids_per_frame for i in ids}):
        seen = [i in ids for ids in ids_per_frame]
        first, last = seen.index(True), len(seen) - 1 - seen[::-1].index(True)
        if not all(seen[first:last + 1]):
            out.append(i)
    return out


def test_4k_dynamic16_departure_and_return_vs_oracle(env, logdir):
    """A 16-frame 3840x2160 dynamic run (every 4th frame of the synthetic sequence: vehicle 1 is out of the picture in frames
    5..9 and back from frame 10) through RcnnTracker.next_frame against the oracle's detector + tracker: track ids per frame
    and the CSV text -- including the BLANK cells of an absent id and the id an object gets back when it is re-associated after
    its absence (visualize_uav.py:131-141, rcnn_tracker.py:136-147; the shipped data/dynamic_dcnn_data.csv has 1 536 blank
    cells) -- must be equal.  Oracle parity itself is unpinned for the detector rows."""
    from PIL import Image
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from oracle import tracker as otr
    from oracle.detector import DetectorOracle, resize_shape
    torch.set_num_threads(_threads())
    cfg = env["cfg"].clone()
    cfg.APSE.MAX_BATCH = 1
    tr = RcnnTracker(cfg, FRAME, env["asd"], detector_state=env["sd"])
    oracle = DetectorOracle(env["sd"])
    otk = otr.TrackerOracle()
    ih, iw = resize_shape(*FRAME)
    ids_hip, ids_ref, lines, olines = [], [], [], []
    for k in range(16):
        frame = env["seq"].frame(4 * k)
        objs = tr.next_frame(frame)
        lines.append(tr.log_line(objs, 1, k)[0])
        ids_hip.append(list(objs.ids) if len(objs) else [])
        img = np.asarray(Image.fromarray(frame).resize((iw, ih), Image.BILINEAR))
        post = oracle.inference(torch.as_tensor(img.astype("float32").transpose(2, 0, 1)), *FRAME)
        rois = otr.features_rois(post["features"]["p2"], post["boxes"], FRAME[1])
        emb = otr.association_head(rois, env["asd"]["fc.weight"], env["asd"]["fc.bias"])
        orec = otk.next_frame(dict(boxes=post["boxes"], scores=post["scores"], classes=post["classes"],
                                   masks=list(zip(post["mask_windows"], post["mask_rects"])), emb=emb))
        olines.append(otr.log_oneline(orec, 1, k)[0])
        ids_ref.append(orec["ids"])
    gaps = _ids_have_gap(ids_ref)
    blank_cells = sum(1 for ln in olines for c in ln.split(",")[1:] if c == "")
    same = sum(a == b for a, b in zip(lines, olines))
    # cell-level comparison: blanks must coincide exactly; numeric cells (integer pixel positions) may move where a mask-edge
    # pixel sits on the >= 0.5 threshold (f32 noise of the mask logits: test_full_frame_vs_oracle sees 4 of 130 557 pixels flip)
    diffs, blank_pattern_equal, cells = [], True, 0
    for k, (a, b) in enumerate(zip(lines, olines)):
        ca, cb = a.split(","), b.split(",")
        if len(ca) != len(cb) or [c == "" for c in ca] != [c == "" for c in cb]:
            blank_pattern_equal = False
            continue
        for j, (u, v) in enumerate(zip(ca, cb)):
            cells += int(u != "")
            if u != v:
                diffs.append(dict(frame=k, cell=j, id=(j - 1) // 4 + 1, what=("cent_x", "cent_y", "clos_x", "clos_y")[(j - 1) % 4],
                                  hip=u, oracle=v, delta=abs(float(u) - float(v))))
    _log(logdir, "dyn16_4k", dict(ids=ids_hip, ref_ids=ids_ref, same_lines=same, ids_with_gap=gaps, blank_cells=blank_cells,
                                  numeric_cells=cells, differing_cells=diffs))
    assert ids_hip == ids_ref                  # ids exact in every frame, through departures and returns
    assert blank_pattern_equal                 # the blank cells of absent ids coincide
    assert gaps and blank_cells >= 4           # the run does contain departures, blank cells, and re-associations after absence
    # [observed round 3: 14 of 16 lines identical text; 3 of 560 numeric cells differ: two centroid coordinates by one pixel, and
    # one closest point that jumps 14 px ALONG a mask edge facing the host -- the pixel the oracle picks sits on the >= 0.5
    # threshold and is not in the HIP mask; both points are at the same distance from the host centroid within a pixel]
    assert same >= 12 and len(diffs) <= 6, diffs
    for d in diffs:
        if d["what"].startswith("cent"):
            assert d["delta"] <= 1, d
        else:
            ca, cb = lines[d["frame"]].split(","), olines[d["frame"]].split(",")
            j = 1 + 4 * (d["id"] - 1)
            hx, hy = float(cb[1]), float(cb[2])                                  # host = id 1: its centroid in the oracle's line
            da = ((float(ca[j + 2]) - hx) ** 2 + (float(ca[j + 3]) - hy) ** 2) ** 0.5
            db_ = ((float(cb[j + 2]) - hx) ** 2 + (float(cb[j + 3]) - hy) ** 2) ** 0.5
            assert abs(da - db_) <= 1.5, (d, da, db_)                            # both are closest points up to one edge pixel


def _run_sequence_mode(env, frames, dtype, batch, camera):
    """frames -> per frame dict(ids, line, boxes [n,4], cent [n,2]) with one context of the given mode (frames in batches)."""
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    cfg = env["cfg"].clone()
    cfg.APSE.MAX_BATCH = batch
    cfg.APSE.DTYPE = dtype
    cfg.APSE.STORAGE16 = dtype != "f32"
    tr = RcnnTracker(cfg, FRAME, env["asd"], detector_state=env["sd"])
    if camera is not None:
        tr.predictor.set_camera(camera)
    out = []
    for lo in range(0, len(frames), batch):
        dets = [o["instances"] for o in tr.predictor.predict_batch(frames[lo:lo + batch], want_masks=False)[0]]
        for j, d in enumerate(dets):
            rec = d._record
            tr.frame_count += 1
            objs = tr._finish_frame(d, None, host_replay=True)
            ids = list(objs.ids) if len(objs) else []
            out.append(dict(ids=sorted(ids), line=tr.log_line(objs, 1, lo + j)[0], boxes=np.asarray(rec["boxes"], np.float64).reshape(-1, 4),
                            cent=np.asarray(rec["centroids"], np.float64).reshape(-1, 2), cls=np.asarray(rec["classes"]).reshape(-1)))
    return out


def _match_frames(a, b, iou_thr=0.9):
    """detections of one frame in two runs paired by class and IoU >= iou_thr -> (pairs, n_a, n_b)."""
    from hip_helpers import _iou_matrix
    na, nb = len(a["boxes"]), len(b["boxes"])
    if not na or not nb:
        return [], na, nb
    u = _iou_matrix(torch.from_numpy(a["boxes"]), torch.from_numpy(b["boxes"])).numpy()
    u[a["cls"][:, None] != b["cls"][None, :]] = 0.0
    pairs, used = [], set()
    for i in np.argsort(-u.max(axis=1)):
        j = int(np.argmax(u[i]))
        if u[i, j] >= iou_thr and j not in used:
            used.add(j)
            pairs.append((int(i), j))
    return pairs, na, nb


def test_16bit_modes_sequence_drift_vs_f32(env, logdir, golden_dir):
    """What bf16 / fp16 do over a SEQUENCE (BASELINE configs[2] / [4] against configs[1]'s precision): the 64-frame 3840x2160
    dynamic sequence through HIP-bf16 (batch 4, undistort + gamma fused) and HIP-fp16 (batch 8), each against HIP-f32 on the
    same input (f32 is the oracle-checked mode).  Per mode (gpurun_out/seq_drift.json):
      detections paired by class and box IoU >= 0.9 (independent of ids): frames whose detection SETS agree, share of paired
        detections, largest / median centroid difference of a pair;
      ids: frames whose id set equals f32's, first frame where it does not, frames whose CSV line is the same text.
    The synthetic weights put every score within ~0.1 of the 0.5 threshold (weights.py tunes the class bias so that ~8 of 1000
    proposals pass), so a detection near the threshold comes and goes under 16-bit rounding and the ids drift from the first such
    event on; trained weights separate the scores.  A characterisation with floors at about half the observed agreement."""
    with open(os.path.join(golden_dir, "cam_params.json")) as f:
        cam = json.load(f)
    frames = [env["seq"].frame(t) for t in range(64)]
    table = {}
    for tag, dtype, batch, camera in (("bf16_b4_preproc", "bf16", 4, cam), ("f16_b8", "f16", 8, None)):
        ref = _run_sequence_mode(env, frames, "f32", 1, camera)
        got = _run_sequence_mode(env, frames, dtype, batch, camera)
        paired = tot_a = tot_b = same_sets = 0
        dc = []
        for a, b in zip(got, ref):
            pairs, na, nb = _match_frames(a, b)
            paired += len(pairs)
            tot_a += na
            tot_b += nb
            same_sets += int(len(pairs) == na == nb)
            dc += [float(np.abs(a["cent"][i] - b["cent"][j]).max()) for i, j in pairs]
        table[tag] = dict(frames=64, detections_mode=tot_a, detections_f32=tot_b, paired_iou90=paired,
                          paired_share_of_f32=round(paired / max(tot_b, 1), 4), frames_same_detection_set=same_sets,
                          paired_centroid_delta_max_px=max(dc) if dc else None,
                          paired_centroid_delta_median_px=float(np.median(dc)) if dc else None,
                          paired_centroid_delta_p99_px=float(np.quantile(dc, 0.99)) if dc else None,
                          frames_same_id_set=sum(a["ids"] == b["ids"] for a, b in zip(got, ref)),
                          first_frame_with_other_ids=next((t for t, (a, b) in enumerate(zip(got, ref)) if a["ids"] != b["ids"]), None),
                          frames_same_csv_line=sum(a["line"] == b["line"] for a, b in zip(got, ref)))
    with open(os.path.join(logdir, "seq_drift.json"), "w") as f:
        json.dump(table, f, indent=1)
    _log(logdir, "seq_drift", table)
    # floors at about half the observed disagreement [round 3: fp16 98.7 % of f32's detections paired, 55 of 64 frames with the same
    # detection set, paired centroids within 6 px; bf16 + gamma (~40 threshold-level detections per frame) 81.2 % paired, p99 15 px]
    assert table["f16_b8"]["paired_share_of_f32"] > 0.97 and table["f16_b8"]["frames_same_detection_set"] >= 46, table["f16_b8"]
    assert table["f16_b8"]["paired_centroid_delta_max_px"] <= 12, table["f16_b8"]
    assert table["bf16_b4_preproc"]["paired_share_of_f32"] > 0.62 and table["bf16_b4_preproc"]["paired_centroid_delta_p99_px"] <= 30, table["bf16_b4_preproc"]


def _given_run(env, frames, seq, dtype, batch, camera):
    """The sequence driver's own loop (tools/run_sequence.py: detect_range on the software-pipelined loop + the reference-shaped
    replay) in given-boxes mode -> (CSV lines, ids per frame, records)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import run_sequence as rs
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    cfg = env["cfg"].clone()
    cfg.APSE.MAX_BATCH = batch
    cfg.APSE.DTYPE = dtype
    tr = RcnnTracker(cfg, FRAME, env["asd"], detector_state=env["sd"])
    if camera is not None:
        tr.predictor.set_camera(camera)
    recs = rs.detect_range(tr, lambda t: frames[t], 0, len(frames), batch, rs.synthetic_given_fn(seq, *FRAME))
    lines, ids = [], []
    for k, rec in enumerate(recs):
        objs = tr.next_record(rec)
        lines.append(tr.log_line(objs, 1, k)[0])
        ids.append(list(objs.ids) if len(objs) else [])
    return lines, ids, recs


def _cell_deltas(lines, ref_lines):
    """(blank pattern equal, identical lines, max |delta| of centroid cells, of closest-point cells, number of differing cells)."""
    same_blank, same, dc, dp, nd = True, 0, 0.0, 0.0, 0
    for a, b in zip(lines, ref_lines):
        ca, cb = a.split(","), b.split(",")
        same += int(a == b)
        if len(ca) != len(cb) or [c == "" for c in ca] != [c == "" for c in cb]:
            same_blank = False
            continue
        for j, (u, v) in enumerate(zip(ca, cb)):
            if j == 0 or u == "" or u == v:
                continue
            nd += 1
            d = abs(float(u) - float(v))
            if (j - 1) % 4 < 2:
                dc = max(dc, d)
            else:
                dp = max(dp, d)
    return same_blank, same, dc, dp, nd


def test_given_boxes_sequence64_ids_16bit_vs_f32_vs_oracle(env, logdir, golden_dir):
    """SURVEY 8d's deterministic form of configs 1-4: the 64-frame dynamic 3840x2160 sequence in GIVEN-BOXES mode
    (the synthetic vehicles' boxes through the reference's detected_instances entry, track_rcnn.py:52-54), so the score-threshold
    lottery of the synthetic weights is out of the picture and what is compared is the tail that produces the CSV: mask head, paste,
    centroids, closest points, roi_pool, association FC, distance < 0.6, id allocation (rcnn_tracker.py:126-147).
      (a) HIP f32 batch 1 against the oracle's given_boxes run: ids per frame and CSV text;
      (b) HIP bf16 batch 4 + fused undistort / gamma, and HIP fp16 batch 8, each against HIP f32 on the same input: ids identical in
          64 / 64 frames, blank pattern identical; centroid / closest-point deltas logged, bars at ~2x the observation.
    Vehicle 1 is out of the picture in frames 20..39 (blank cells, re-association on return)."""
    from PIL import Image
    from oracle import tracker as otr
    from oracle.detector import DetectorOracle, resize_shape
    with open(os.path.join(golden_dir, "cam_params.json")) as f:
        cam = json.load(f)
    seq = env["seq"]
    frames = [seq.frame(t) for t in range(64)]
    ref_lines, ref_ids, _ = _given_run(env, frames, seq, "f32", 1, None)
    table = {}
    # (a) oracle
    torch.set_num_threads(_threads())
    oracle = DetectorOracle(env["sd"])
    otk = otr.TrackerOracle()
    ih, iw = resize_shape(*FRAME)
    sc = torch.tensor([iw / FRAME[1], ih / FRAME[0], iw / FRAME[1], ih / FRAME[0]], dtype=torch.float32)
    olines, oids = [], []
    for t in range(64):
        img = np.asarray(Image.fromarray(frames[t]).resize((iw, ih), Image.BILINEAR))
        gb = torch.from_numpy(seq.boxes(t)) * sc
        post = oracle.inference(torch.as_tensor(img.astype("float32").transpose(2, 0, 1)), *FRAME, given_boxes=gb,
                                given_classes=torch.zeros(len(gb), dtype=torch.int64))
        rois = otr.features_rois(post["features"]["p2"], post["boxes"], FRAME[1])
        emb = otr.association_head(rois, env["asd"]["fc.weight"], env["asd"]["fc.bias"])
        orec = otk.next_frame(dict(boxes=post["boxes"], scores=post["scores"], classes=post["classes"],
                                   masks=list(zip(post["mask_windows"], post["mask_rects"])), emb=emb))
        olines.append(otr.log_oneline(orec, 1, t)[0])
        oids.append(orec["ids"])
    blank, same, dc, dp, nd = _cell_deltas(ref_lines, olines)
    table["f32_b1_vs_oracle"] = dict(frames_same_ids=sum(a == b for a, b in zip(ref_ids, oids)), blank_pattern_equal=blank,
                                     identical_csv_lines=same, differing_cells=nd, centroid_delta_max_px=dc, closest_delta_max_px=dp,
                                     blank_cells=sum(1 for ln in olines for c in ln.split(",")[1:] if c == ""), max_id=max(max(i) for i in oids if i))
    # (b) the 16-bit configurations against HIP f32 on the same input
    ref_cam = _given_run(env, frames, seq, "f32", 1, cam)
    for tag, dtype, batch, camera, ref in (("bf16_b4_preproc_vs_f32_preproc", "bf16", 4, cam, ref_cam),
                                           ("f16_b8_vs_f32", "f16", 8, None, (ref_lines, ref_ids, None))):
        lines, ids, _ = _given_run(env, frames, seq, dtype, batch, camera)
        blank, same, dc, dp, nd = _cell_deltas(lines, ref[0])
        table[tag] = dict(frames_same_ids=sum(a == b for a, b in zip(ids, ref[1])), blank_pattern_equal=blank, identical_csv_lines=same,
                          differing_cells=nd, centroid_delta_max_px=dc, closest_delta_max_px=dp)
    with open(os.path.join(logdir, "given_boxes_seq64.json"), "w") as f:
        json.dump(table, f, indent=1)
    _log(logdir, "given_boxes_seq64", table)
    a = table["f32_b1_vs_oracle"]
    # [round 4, MI355X: f32 vs the oracle 64 / 64 frames with the same ids, 64 / 64 identical CSV lines, 0 of 1 200 numeric cells differ,
    #  80 blank cells; bf16 batch 4 + undistort / gamma vs f32 on the same input: ids 64 / 64, centroids within 7 px, closest points
    #  within 245 px (a closest point slides along a mask edge when an edge pixel flips); fp16 batch 8: ids 64 / 64, 3 px / 35 px]
    assert a["frames_same_ids"] == 64 and a["blank_pattern_equal"] and a["blank_cells"] >= 40, a
    assert a["identical_csv_lines"] == 64 and a["differing_cells"] == 0, a
    for tag, cent_bar, clos_bar in (("bf16_b4_preproc_vs_f32_preproc", 14, 490), ("f16_b8_vs_f32", 6, 70)):      # 2x the observation
        assert table[tag]["frames_same_ids"] == 64 and table[tag]["blank_pattern_equal"], (tag, table[tag])
        assert table[tag]["centroid_delta_max_px"] <= cent_bar and table[tag]["closest_delta_max_px"] <= clos_bar, (tag, table[tag])


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_4k_fused_bottleneck_equals_three_kernel_form(env, logdir, dtype, monkeypatch):
    """csrc/bottleneck16.hip at the size it was built for: 3840x2160 frames (res2 maps of 192 x 336: 24 x 21 tiles of 8 x 16, 504 per
    image, persistent blocks walking ~2 tiles each at batch 2), batch 2, against the three-kernel form (APSE_NO_BNECK_FUSE):
    res2, p2 and the per-image records must be the same bits."""
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    frames = [env["seq"].frame(0), env["seq"].frame(9)]
    got = []
    for unfused in (False, True):
        if unfused:
            monkeypatch.setenv("APSE_NO_BNECK_FUSE", "1")
        else:
            monkeypatch.delenv("APSE_NO_BNECK_FUSE", raising=False)
        cfg = env["cfg"].clone()
        cfg.APSE.MAX_BATCH = 2
        cfg.APSE.DTYPE = dtype
        cfg.APSE.STORAGE16 = True
        tr = RcnnTracker(cfg, FRAME, env["asd"], detector_state=env["sd"])
        out = [o["instances"] for o in tr.predictor.predict_batch(frames, want_masks=False)[0]]
        model = tr.predictor.model
        got.append((model.debug_tensor("res2").cpu(), model.debug_tensor("p2").cpu(), [_record_bytes(o) for o in out], [len(o) for o in out]))
        del tr
    _log(logdir, "bneck_fused_4k/" + dtype, dict(n=got[0][3], res2_equal=bool(torch.equal(got[0][0].view(torch.int16), got[1][0].view(torch.int16)))))
    assert all(n > 0 for n in got[0][3])
    assert torch.equal(got[0][0].view(torch.int16), got[1][0].view(torch.int16))
    assert torch.equal(got[0][1].view(torch.int16), got[1][1].view(torch.int16))
    assert got[0][2] == got[1][2]


def test_4k_context_resize_equals_pillow(env):
    """The context's own resize path (apse_preprocess_frames: tap-major coefficient table, 16-byte row pitch of the intermediate
    image, dword vertical pass -- csrc/elementwise.hip) against Pillow itself, batch 2: the network input (f32 NHWC4, mean
    subtracted, zero padded) must be Pillow's bytes minus the mean, exactly."""
    from PIL import Image
    from oracle.detector import resize_shape
    tr = env["tr"]
    model = tr.predictor.model
    frames = [env["seq"].frame(2), env["seq"].frame(13)]
    model.preprocess_frames(torch.from_numpy(np.stack(frames)).cuda())
    ih, iw = resize_shape(*FRAME)
    ph, pw = (ih + 31) // 32 * 32, (iw + 31) // 32 * 32
    got = model.debug_tensor("input").cpu().view(2, ph, pw, 4)
    mean = torch.tensor(list(tr.predictor.cfg.MODEL.PIXEL_MEAN), dtype=torch.float32)
    for b, f in enumerate(frames):
        ref = torch.from_numpy(np.asarray(Image.fromarray(f).resize((iw, ih), Image.BILINEAR)).astype(np.float32)) - mean
        assert torch.equal(got[b, :ih, :iw, :3], ref)
        assert float(got[b, ih:].abs().sum()) == 0.0 and float(got[b, :, iw:].abs().sum()) == 0.0 and float(got[b, ..., 3].abs().sum()) == 0.0
