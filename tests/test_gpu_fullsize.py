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
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    post = DetectorOracle(env["sd"]).inference(torch.as_tensor(img.astype("float32").transpose(2, 0, 1)), *FRAME)
    for k in ("p2", "p4", "p6"):
        got, ref = feats[k].cpu(), post["features"][k]
        d = float((got - ref).abs().max() / ref.abs().max())
        _log(logdir, "feat/" + k, dict(rel=d))
        assert d < 2e-4                                   # f32 through 104 convolutions, different sum order
    model = tr.predictor.model
    P = int(model.last_results.prop_count[0])
    ref_props = post["proposals"]["boxes"]
    props = model.debug_tensor("proposals").cpu().view(-1, 4)[:P]
    same_p = int((props - ref_props[:P]).abs().max(dim=1).values.lt(0.05).sum()) if P == ref_props.shape[0] else -1
    _log(logdir, "rpn", dict(P=P, ref_P=int(ref_props.shape[0]), rows_equal=same_p))
    assert P == ref_props.shape[0]
    assert same_p >= P - 20                               # near-tied logits at rank ~1000 may swap a few rows
    n = len(inst)
    _log(logdir, "dets", dict(n=n, ref_n=int(post["boxes"].shape[0]), scores=[round(float(s), 5) for s in inst.scores],
                              ref=[round(float(s), 5) for s in post["scores"]]))
    assert n == post["boxes"].shape[0]
    assert torch.equal(inst.pred_classes, post["classes"])
    assert float((inst.pred_boxes.tensor - post["boxes"]).abs().max()) < 0.1        # 4K frame pixels
    assert float((inst.scores - post["scores"]).abs().max()) < 1e-4
    bad = tot = 0
    for k in range(n):
        m = inst.pred_masks[k]
        assert tuple(m.rect) == tuple(post["mask_rects"][k])
        bad += int((m.window().cpu() != post["mask_windows"][k]).sum())
        tot += int(m.mass)
    _log(logdir, "masks", dict(mismatched=bad, total=tot))
    assert bad <= max(16, tot // 5000)            # >= 0.5 threshold on f32 bilinear values: edge pixels may flip
    if n:
        rois = otr.features_rois(post["features"]["p2"], post["boxes"], FRAME[1])
        emb = otr.association_head(rois, env["asd"]["fc.weight"], env["asd"]["fc.bias"])
        de = float((torch.from_numpy(inst._record["embeddings"]) - emb).abs().max())
        _log(logdir, "emb", dict(max_abs=de))
        assert de < 2e-3


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
    torch.set_num_threads(min(32, os.cpu_count() or 1))
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
        if line != oline:                     # a threshold-edge pixel may move one integer cell by 1
            a, b = line.split(","), oline.split(",")
            assert len(a) == len(b)
            assert all(x == y or abs(float(x) - float(y)) <= 1.0 for x, y in zip(a, b))
    assert same >= 3


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_full_frame_16bit_vs_oracle(env, logdir, dtype):
    """BASELINE configs 3 / 5 precision at full size: 16-bit matrix cores and 16-bit activation storage (this is where the
    256x128 tile and the deep-K 16-bit shapes are live) against the oracle with the same rounding points
    (operands AND stored tensors rounded to the 16-bit type).  A different f32 accumulation order can flip a 16-bit
    rounding (2^-9 / 2^-11 relative) of a next-layer input, so features are compared on mean error and the detections
    after matching by box; parity unpinned like the f32 detector (oracle restates detectron2)."""
    from PIL import Image
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from oracle.detector import DetectorOracle, resize_shape
    cfg = env["cfg"].clone()
    cfg.APSE.MAX_BATCH = 1
    cfg.APSE.DTYPE = dtype
    cfg.APSE.STORAGE16 = True
    tr = RcnnTracker(cfg, FRAME, env["asd"], detector_state=env["sd"])
    frame = env["seq"].frame(0)
    pred, feats = tr.predictor(frame)
    inst = pred["instances"]
    ih, iw = resize_shape(*FRAME)
    img = np.asarray(Image.fromarray(frame).resize((iw, ih), Image.BILINEAR))
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    oracle = DetectorOracle(env["sd"], dict(bf16=("f16" if dtype == "f16" else True), storage16=True))
    post = oracle.inference(torch.as_tensor(img.astype("float32").transpose(2, 0, 1)), *FRAME)
    lim_max, lim_mean = (6e-2, 1.5e-2) if dtype == "bf16" else (1e-2, 2.5e-3)
    for k in ("p2", "p4", "p6"):
        got, ref = feats[k].cpu(), post["features"][k]
        d = float((got - ref).abs().max() / ref.abs().max())
        mean = float((got - ref).abs().mean() / ref.abs().mean())
        _log(logdir, dtype + "/feat/" + k, dict(rel_max=d, rel_mean=mean))
        assert d < lim_max and mean < lim_mean
    n, rn = len(inst), int(post["boxes"].shape[0])
    matched = 0
    for k in range(n):
        dd = (post["boxes"] - inst.pred_boxes.tensor[k]).abs().max(dim=1).values if rn else torch.tensor([])
        j = int(dd.argmin()) if rn else -1
        if rn and float(dd[j]) < 8.0 and int(post["classes"][j]) == int(inst.pred_classes[k]):      # 4K pixels
            matched += 1
    _log(logdir, dtype + "/dets", dict(n=n, ref_n=rn, matched=matched, scores=[round(float(s), 4) for s in inst.scores],
                                       ref=[round(float(s), 4) for s in post["scores"]]))
    assert abs(n - rn) <= 2 and matched >= min(n, rn) - 2


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
