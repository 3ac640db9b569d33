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
    n = len(r["scores"])
    assert v.shape == (record_len(n, 128),) == (1 + 13 * n + 2 * n * n + 128 * n,)      # count-prefixed: a frame costs what it holds
    assert record_len(4) * 4 < 2500 and record_len(100) == 34101                       # ~2.3 KB typical, 136 KB worst case
    u = unpack_record(v, 100, 128)
    assert u.pop("wire_floats") == v.size
    for k in ("boxes", "scores", "classes", "centroids", "mass", "rects", "closest", "embeddings"):
        assert np.array_equal(u[k], r[k]), k
    cover = []
    for rank in range(8):
        lo, hi = shard_frames(2734, rank, 8)
        cover += list(range(lo, hi))
    assert cover == list(range(2734))


def _bench(args, env_extra, timeout=180):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, lines, p.stderr


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` without a launcher must run TWO ranks and say n_gpus 2 in rank 0's single JSON line
    (the GPU-free --rehearse-spawn body: same rendezvous / gather / max-over-ranks timing as the real run)."""
    rc, lines, err = _bench(["--gpus", "2", "--steps", "3", "--rehearse-spawn"], {"APSE_DIST_BACKEND": "gloo"})
    assert rc == 0, err[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["records_gathered"] == 6 and lines[0]["steps"] == 3


def test_bench_gpus_flag_must_match_the_launcher():
    rc, lines, err = _bench(["--gpus", "4", "--rehearse-spawn"], {"WORLD_SIZE": "2", "RANK": "0", "APSE_DIST_BACKEND": "gloo"})
    assert rc != 0 and not lines and "WORLD_SIZE=2" in err
    rc, lines, err = _bench(["--rehearse-spawn"], {"WORLD_SIZE": "2", "RANK": "0", "APSE_DIST_BACKEND": "gloo"})
    assert rc != 0 and not lines           # torchrun with 2 ranks but --gpus left at 1: refuse, never print n_gpus 1


def test_bench_a_dead_rank_fails_the_job():
    rc, lines, err = _bench(["--gpus", "2", "--steps", "2", "--rehearse-spawn"],
                            {"APSE_DIST_BACKEND": "gloo", "APSE_REHEARSE_FAIL_RANK": "1"})
    assert rc != 0 and not lines
