"""CPU (-m "not gpu"): FastReplay (rank-0 replay of gathered records) must reproduce the reference-shaped
RcnnTracker.next_record + log_line path line for line, including births, deaths after 100 unseen frames,
absent host, empty frames and empty masks."""
import numpy as np
import pytest
import torch


def _stream(nframes, seed=0):
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((12, 128)).astype(np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    recs = []
    for t in range(nframes):
        alive = [k for k in range(12) if not (k == 3 and 40 <= t < 170) and not (k == 0 and 20 <= t < 30) and (k < 8 or t % 37 == k)]
        if t in (5, 6):
            alive = []
        rng.shuffle(alive)
        n = len(alive)
        emb = base[alive] + 0.02 * rng.standard_normal((n, 128)).astype(np.float32)
        emb /= np.maximum(np.linalg.norm(emb, axis=1, keepdims=True), 1e-12)
        cent = rng.integers(1, 3000, (n, 2)).astype(np.int32)
        if n and t == 11:
            cent[0] = -1                                  # empty mask
        recs.append(dict(boxes=(rng.random((n, 4)) * 1000).astype(np.float32), scores=rng.random(n).astype(np.float32),
                         classes=rng.integers(0, 4, n).astype(np.int64), centroids=cent,
                         mass=rng.integers(1, 9000, n).astype(np.int32), rects=rng.integers(0, 3000, (n, 4)).astype(np.int32),
                         closest=rng.integers(1, 3000, (n, n, 2)).astype(np.int32), embeddings=emb.astype(np.float32),
                         packed_index=np.arange(n)))
    return recs


def test_fast_replay_equals_tracker_path():
    from apse_uav_amd.config import setup_cfg
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from apse_uav_amd.engines.replay import FastReplay
    from apse_uav_amd.weights import synthetic_association_state
    tr = RcnnTracker(setup_cfg(device="cpu"), (2160, 3840), synthetic_association_state(1), detector_state={})
    for host in (1, 4):
        tr.reset_tracker()
        fr = FastReplay(host)
        for t, rec in enumerate(_stream(260, seed=host)):
            objs = tr.next_record(rec)
            line, hi = tr.log_line(objs, host, t)
            fline, ids = fr.step(rec, t)
            assert fline == line, (t, fline, line)
            assert ids == (list(objs.ids) if len(objs) else [])
        assert fr.max_id >= 10 and fr.next_id == tr.objects.get_new_id()


def test_native_replay_equals_fast_replay():
    """apse_replay_* (C++ in libapse_hip.so, host-only) vs FastReplay (scipy Hungarian) on long random streams."""
    from apse_uav_amd.engines.replay import FastReplay, NativeReplay
    for host in (1, 4, 9):
        fr, nr = FastReplay(host), NativeReplay(host)
        for t, rec in enumerate(_stream(400, seed=10 + host)):
            a, ids_a = fr.step(rec, t)
            b, ids_b = nr.step(rec, t)
            assert a == b, (t, a, b)
            assert sorted(ids_a) == ids_b
        assert fr.max_id == nr.max_id and fr.next_id == nr.next_id


def test_native_replay_packed_wire_format():
    """One C call over a whole shard in the gather's wire format == per-record FastReplay."""
    from apse_uav_amd.engines.replay import FastReplay, NativeReplay
    from apse_uav_amd.sharding import pack_record
    recs = _stream(120, seed=5)
    packed = np.concatenate([pack_record(r, 100, 128) for r in recs])
    fr, nr = FastReplay(2), NativeReplay(2)
    ref = [fr.step(r, 7 + t)[0] for t, r in enumerate(recs)]
    got = nr.run_packed(packed, len(recs), kd=100, first_frame=7, chunk=50)       # 120 records: three chunks, ids carried across them
    assert got == ref and nr.max_id == fr.max_id
    assert NativeReplay(2).run_packed((packed, len(recs)), first_frame=7) == ref
    with pytest.raises(RuntimeError):
        NativeReplay(2).run_packed(packed[:-5], len(recs))                        # truncated payload: refused, not read past
