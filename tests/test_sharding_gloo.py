"""CPU (-m "not gpu"): the N > 1 path -- frame sharding + the single gather of per-frame records --
with two gloo ranks, and the host association replayed from gathered records."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _records(lo, hi):
    out = []
    for t in range(lo, hi):
        rng = np.random.default_rng(100 + t)
        n = int(rng.integers(0, 5))
        emb = rng.standard_normal((n, 128)).astype(np.float32)
        emb /= np.maximum(np.linalg.norm(emb, axis=1, keepdims=True), 1e-12)
        out.append(dict(boxes=(rng.random((n, 4)) * 1000).astype(np.float32), scores=rng.random(n).astype(np.float32),
                        classes=rng.integers(0, 4, n).astype(np.int64), centroids=rng.integers(1, 3000, (n, 2)).astype(np.int32),
                        mass=rng.integers(1, 90000, n).astype(np.int32), rects=rng.integers(0, 3000, (n, 4)).astype(np.int32),
                        closest=rng.integers(1, 3000, (n, n, 2)).astype(np.int32), embeddings=emb))
    return out


def _worker(rank, world, port, nframes, q):
    import torch.distributed as dist
    from apse_uav_amd.sharding import gather_records, shard_frames
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_frames(nframes, rank, world)
    got = gather_records(_records(lo, hi), rank, world, torch.device("cpu"))
    if rank == 0:
        q.put([{k: np.asarray(v).tolist() for k, v in r.items() if k != "packed_index"} for r in got])
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_records_two_ranks_equals_single():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    nframes = 7                                   # uneven shards: 4 + 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nframes, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _records(0, nframes)
    assert len(got) == nframes
    for g, r in zip(got, ref):
        for k in ("boxes", "scores", "classes", "centroids", "mass", "rects", "closest", "embeddings"):
            assert np.array_equal(np.asarray(g[k], dtype=np.asarray(r[k]).dtype).reshape(np.asarray(r[k]).shape), r[k]), k


def test_pack_unpack_roundtrip_and_shards():
    from apse_uav_amd.sharding import pack_record, record_len, shard_frames, unpack_record
    r = _records(3, 4)[0]
    v = pack_record(r, 100, 128)
    assert v.shape == (record_len(100, 128),)
    u = unpack_record(v, 100, 128)
    for k in ("boxes", "scores", "classes", "centroids", "mass", "rects", "closest", "embeddings"):
        assert np.array_equal(u[k], r[k]), k
    cover = []
    for rank in range(8):
        lo, hi = shard_frames(2734, rank, 8)
        cover += list(range(lo, hi))
    assert cover == list(range(2734))
