"""-m gpu: the ingest path of SURVEY 8(f) rank 1 at full size (3840x2160, R-101-FPN, f32) against the plain call.

The reference loop is /root/reference/dcnn/scripts/tests/visualize_uav.py:172-190 (read a frame, `tracker.next_frame(frame)`).
The build's extension `next_frame(frame, upcoming=next)` / `TrackPredictor.prefetch` stages the NEXT frame through pinned
memory on a copy stream while the current one computes (engines/track_predictor.py: `h2d_done` / `consumed` events, identity
match of the announced array, fallback when a prefetch is never used).  That is cross-stream code: every variant below must
give the per-frame records of the plain loop BYTE FOR BYTE and the same ids / CSV text.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
FRAME = (2160, 3840)
N = 6


@pytest.fixture(scope="module")
def env():
    from apse_uav_amd.config import setup_cfg
    from apse_uav_amd.synthetic import SyntheticSequence
    from apse_uav_amd.weights import UAV4K_R101_CLS_BIAS, synthetic_association_state, synthetic_detector_state
    sd = synthetic_detector_state(0, cls_bias=UAV4K_R101_CLS_BIAS)
    asd = synthetic_association_state(1)
    cfg = setup_cfg()
    cfg.APSE.MAX_BATCH = 1
    seq = SyntheticSequence("dynamic", *FRAME)
    frames = [seq.frame(7 * t) for t in range(N)]           # the dynamic sequence moves: every frame differs
    return dict(sd=sd, asd=asd, cfg=cfg, frames=frames)


def _tracker(env):
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    return RcnnTracker(env["cfg"], FRAME, env["asd"], detector_state=env["sd"])


def _step(tr, frame, t, **kw):
    """ids, CSV line, packed record bytes, and the bytes of every detection's mask window (copied from the bit planes AFTER the
    results were read: with an announced frame the next forward is already enqueued by then, and the library's two alternating
    sets of bit planes must still hand out THIS frame's masks)."""
    from apse_uav_amd.sharding import pack_record
    objs = tr.next_frame(frame, **kw)
    wins = b"".join(bytes(m.rect.__repr__(), "ascii") + m.window().cpu().numpy().tobytes() for m in objs.pred_masks) if len(objs) else b""
    return (list(objs.ids) if len(objs) else [], tr.log_line(objs, 1, t)[0], pack_record(tr._last_record, 100, 128).tobytes(), wins)


@pytest.fixture(scope="module")
def plain(env):
    tr = _tracker(env)
    out = [_step(tr, f, t) for t, f in enumerate(env["frames"])]
    assert any(o[0] for o in out) and len({o[2] for o in out}) == N        # detections, and the frames really differ
    return out


def test_announced_next_frame_equals_plain(env, plain):
    """next_frame(f_t, upcoming=f_{t+1}): the upload of frame t+1 runs on the copy stream under frame t's kernels."""
    tr = _tracker(env)
    fr = env["frames"]
    got = []
    for t in range(N):
        got.append(_step(tr, fr[t], t, upcoming=fr[t + 1] if t + 1 < N else None))
        # the announced frame was uploaded, its resize staged behind this forward and its network enqueued ahead (run_ahead):
        # the next call only reads -- or, after the last frame, nothing is left in flight
        rt = tr.predictor.model._running_tag
        assert (rt is not None and rt[0][0] is fr[t + 1]) if t + 1 < N else rt is None
    assert [g[2] == p[2] for g, p in zip(got, plain)] == [True] * N
    assert [(g[0], g[1]) for g in got] == [(p[0], p[1]) for p in plain]
    assert [g[3] == p[3] for g, p in zip(got, plain)] == [True] * N and len({p[3] for p in plain}) == N      # this frame's masks, not the next one's
    # and every prefetch was actually consumed (identity match), not re-uploaded
    assert tr.predictor._prefetched is None and tr.predictor.model._input_tag is None


def test_announced_frame_differs_from_given(env, plain):
    """The caller announces one array and then passes ANOTHER (a copy of the right frame, or a different frame altogether):
    the stale prefetch must be dropped and the given frame uploaded afresh -- never the announced bytes."""
    tr = _tracker(env)
    fr = env["frames"]
    got = []
    for t in range(N):
        if t % 2 == 0:
            announce = fr[(t + 3) % N]                      # a different frame than the one that will come
        else:
            announce = fr[(t + 1) % N].copy()               # the right content in a different array object
        got.append(_step(tr, fr[t], t, upcoming=announce))
    assert [g[2] == p[2] for g, p in zip(got, plain)] == [True] * N
    assert [(g[0], g[1]) for g in got] == [(p[0], p[1]) for p in plain]


def test_prefetch_never_consumed(env, plain):
    """A prefetch that no call ever uses (the caller changed its mind, or the sequence ended): later plain calls, a
    second unused prefetch on top of it, and a prefetch issued between two plain calls must not disturb any frame."""
    tr = _tracker(env)
    fr = env["frames"]
    got = []
    for t in range(N):
        if t in (1, 2):
            tr.predictor.prefetch(fr[(t + 2) % N])          # announced, never passed
        if t == 2:
            tr.predictor.prefetch([fr[0]])                  # a second one over the first
        got.append(_step(tr, fr[t], t))
    tr.predictor.prefetch(fr[0])                            # left dangling at the end of the sequence
    torch.cuda.synchronize()
    assert [g[2] == p[2] for g, p in zip(got, plain)] == [True] * N
    assert [(g[0], g[1]) for g in got] == [(p[0], p[1]) for p in plain]


def test_mutating_the_consumed_frame_buffer_is_safe(env, plain):
    """The reference loop reuses its frame variable (`ret, frame = video.read()`): once next_frame has returned, the
    caller may overwrite the array it passed, with or without a prefetch in flight for the NEXT frame."""
    tr = _tracker(env)
    fr = [f.copy() for f in env["frames"]]
    got = []
    for t in range(N):
        got.append(_step(tr, fr[t], t, upcoming=fr[t + 1] if t + 1 < N else None))
        fr[t][:] = 0                                         # the caller's buffer is its own again
    assert [g[2] == p[2] for g, p in zip(got, plain)] == [True] * N
    assert [g[3] == p[3] for g, p in zip(got, plain)] == [True] * N


def test_start_frame_dynamic_equals_tracker_started_there(env, tmp_path):
    """tools/run_sequence.py --start-frame 3 on the 9-frame DYNAMIC sequence: equal to a fresh tracker fed frames 3..8 with
    absolute frame numbers (the reference semantics: the tracker's first frame is frame S).  The static-sequence form of this
    check (rows == rows 3.. of the full run) is in test_gpu_a_multirank.py."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import run_sequence as rs
    from apse_uav_amd.synthetic import SyntheticSequence
    from apse_uav_amd.utils import csv_log
    out = str(tmp_path / "part.csv")
    rs.main(["--frames", "9", "--kind", "dynamic", "--size", "2160x3840", "--start-frame", "3", "--out", out])
    seq = SyntheticSequence("dynamic", *FRAME)
    tr = _tracker(env)
    lines = []
    for t in range(3, 9):
        objs = tr.next_frame(seq.frame(t))
        lines.append(tr.log_line(objs, 1, t)[0])
    ref = str(tmp_path / "ref.csv")
    csv_log.write_consumer_csv(ref, lines, 1, [2, 3, 4], first_frame=3)
    with open(out) as f, open(ref) as g:
        a, b = f.read(), g.read()
    assert a == b
    assert [r.split(",")[0] for r in a.split("\n")[2:-1]] == ["3", "4", "5", "6", "7", "8"]


def test_run_ahead_forward_discarded_when_the_caller_changes_course(env, plain):
    """A forward enqueued ahead for the announced frame must not leak into anything else: the caller announces frame t + 1 but then
    (a) asks the predictor for the same frame t again, (b) uses the model-level entry with given boxes, (c) carries on with the
    sequence -- every result must be the plain loop's bytes / its own re-run's."""
    from apse_uav_amd.sharding import pack_record
    tr = _tracker(env)
    fr = env["frames"]
    a = _step(tr, fr[0], 0, upcoming=fr[1])
    assert a[2] == plain[0][2] and tr.predictor.model._running_tag is not None
    # (a) the predictor is asked for frame 0 again (not the announced one): fresh upload + forward, identical record
    inst = tr.predictor(fr[0])[0]["instances"]
    assert pack_record(inst._record, 100, 128).tobytes() == plain[0][2]
    assert tr.predictor.model._running_tag is None
    # (b) announce again, then a given-boxes forward of another frame through predict_batch
    tr2 = _tracker(env)
    _step(tr2, fr[0], 0, upcoming=fr[1])
    boxes = np.array([[100.0, 100.0, 300.0, 260.0]], np.float32)
    given = (boxes, np.zeros(1, np.int32), np.array([1], np.int32))
    g1 = tr2.predictor.predict_batch([fr[2]], given=given)[0][0]["instances"]
    g2 = _tracker(env).predictor.predict_batch([fr[2]], given=given)[0][0]["instances"]
    assert pack_record(g1._record, 100, 128).tobytes() == pack_record(g2._record, 100, 128).tobytes()
    # (c) the sequence continues from frame 1 after the detour of (a): ids / records as in the plain loop
    b = _step(tr, fr[1], 1, upcoming=fr[2])
    c = _step(tr, fr[2], 2)
    assert (b[2], c[2]) == (plain[1][2], plain[2][2]) and (b[0], c[0]) == (plain[1][0], plain[2][0])


def test_announced_loop_over_a_longer_dynamic_run(env):
    """24 frames of the dynamic sequence (births, a departure and a return) through next_frame(f_t, upcoming=f_{t+1}) -- every forward
    enqueued one call ahead, the two sets of mask bit planes alternating 12 times -- against the plain loop: ids, CSV lines, packed
    records and mask windows of every frame."""
    from apse_uav_amd.synthetic import SyntheticSequence
    seq = SyntheticSequence("dynamic", *FRAME)
    fr = [seq.frame(4 * t) for t in range(24)]
    a, b = _tracker(env), _tracker(env)
    for t in range(24):
        p = _step(a, fr[t], t)
        g = _step(b, fr[t], t, upcoming=fr[t + 1] if t + 1 < 24 else None)
        assert g == p, "frame %d differs" % t


def test_announced_frame_of_another_size(env, plain):
    """``upcoming=`` with a frame whose H x W differs from the current one.  Pre-staging it would rebuild the
    context between read_begin and read_end and lose the current frame's results; the predictor must skip the pre-stage (and the
    run-ahead) and let the next call take the normal upload path.  Frame 0 (4K) announcing a 1080p frame, then that frame, then
    4K again: the 4K records equal the plain loop's."""
    from apse_uav_amd.synthetic import SyntheticSequence
    tr = _tracker(env)
    fr = env["frames"]
    small = SyntheticSequence("dynamic", 1080, 1920).frame(3)
    a = _step(tr, fr[0], 0, upcoming=small)
    assert a[2] == plain[0][2] and a[3] == plain[0][3]
    assert tr.predictor.model._running_tag is None and tr.predictor.model._input_tag is None     # nothing staged, nothing run ahead
    pred, _ = tr.predictor(small)                       # the announced frame: resident (prefetched), context rebuilt for 1080p here
    assert pred["instances"].image_size == (1080, 1920)
    tr2 = _tracker(env)
    ref_small, _ = tr2.predictor(small)
    assert torch.equal(pred["instances"].pred_boxes.tensor, ref_small["instances"].pred_boxes.tensor)
    b, _ = tr.predictor(fr[1])                          # back to 4K
    from apse_uav_amd.sharding import pack_record
    ref = _tracker(env).predictor(fr[1])[0]["instances"]
    assert pack_record(b["instances"]._record).tobytes() == pack_record(ref._record).tobytes()
