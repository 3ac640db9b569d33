"""Frame sharding across GPUs and the single gather of per-frame records.

The reference is single-process.  Per-frame GPU work (detector, masks, centroids, closest-point
table, embeddings) is a pure function of one frame, so frames shard embarrassingly across ranks;
only the id assignment (Hungarian + track store, /root/reference/dcnn/engines/rcnn_tracker.py:122-147)
is sequential.  Each rank therefore emits a fixed-layout record per frame and rank 0 receives all of
them with ONE collective at sequence end (``torch.distributed`` gather: RCCL over xGMI with the
"nccl" backend on GPUs, gloo in the CPU tests), then replays the association in frame order.
"""
import numpy as np
import torch

_FIELDS = (("boxes", np.float32, 4), ("scores", np.float32, 1), ("classes", np.int64, 1), ("centroids", np.int32, 2),
           ("mass", np.int32, 1), ("rects", np.int32, 4))


def shard_frames(n_frames, rank, world):
    """Contiguous frame range [lo, hi) of this rank."""
    per = (n_frames + world - 1) // world
    lo = min(rank * per, n_frames)
    return lo, min(lo + per, n_frames)


def pack_record(rec, kd, edim):
    """Record -> flat f32 vector of fixed length (so one tensor gather moves every frame)."""
    n = len(rec["scores"])
    out = np.zeros(record_len(kd, edim), np.float32)
    out[0] = n
    o = 1
    for name, _, width in _FIELDS:
        a = np.asarray(rec[name]).reshape(n, width).astype(np.float32)    # all values < 2^24: exact in f32
        out[o:o + n * width] = a.reshape(-1)
        o += kd * width
    cl = np.asarray(rec["closest"], np.float32).reshape(n, n, 2)
    blk = np.zeros((kd, kd, 2), np.float32)
    blk[:n, :n] = cl
    out[o:o + kd * kd * 2] = blk.reshape(-1)
    o += kd * kd * 2
    emb = np.zeros((kd, edim), np.float32)
    emb[:n] = rec["embeddings"]
    out[o:o + kd * edim] = emb.reshape(-1)
    return out


def record_len(kd, edim):
    return 1 + kd * sum(w for _, _, w in _FIELDS) + kd * kd * 2 + kd * edim


def unpack_record(vec, kd, edim):
    n = int(vec[0])
    rec = {}
    o = 1
    for name, dt, width in _FIELDS:
        a = vec[o:o + n * width].reshape(n, width).astype(dt)
        rec[name] = a[:, 0] if width == 1 else a
        o += kd * width
    rec["closest"] = vec[o:o + kd * kd * 2].reshape(kd, kd, 2)[:n, :n].astype(np.int32)
    o += kd * kd * 2
    rec["embeddings"] = vec[o:o + kd * edim].reshape(kd, edim)[:n].astype(np.float32).copy()
    rec["packed_index"] = np.arange(n)
    return rec


def gather_records(records, rank, world, device, kd=100, edim=128, unpack=True):
    """All ranks call this once; rank 0 gets the records of every rank in rank order (frame order for
    contiguous shards), the others get None.  Ranks may hold different numbers of frames.

    One collective on the data path: ``all_gather_into_tensor`` of the padded record block (RCCL over xGMI
    with the "nccl" backend, gloo on CPU).  The payload is KBs..MBs per rank, so gathering to every rank
    instead of only rank 0 costs nothing measurable and uses the best-supported RCCL primitive."""
    import torch.distributed as dist
    L = record_len(kd, edim)
    cnt = torch.tensor([len(records)], device=device, dtype=torch.int64)
    cnts = torch.zeros((world,), device=device, dtype=torch.int64)
    dist.all_gather_into_tensor(cnts, cnt)
    counts = [int(v) for v in cnts.cpu().tolist()]
    mx = max(max(counts), 1)
    buf = torch.zeros((mx, L), dtype=torch.float32)
    for i, r in enumerate(records):
        buf[i] = torch.from_numpy(pack_record(r, kd, edim))
    buf = buf.to(device)
    out = torch.empty((world * mx, L), dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(out, buf)
    if rank != 0:
        return None
    h = out.cpu().numpy().reshape(world, mx, L)
    if not unpack:
        # wire format kept: rank-ordered [frames, L] array for NativeReplay.run_packed
        return np.concatenate([h[r, :counts[r]] for r in range(world)], axis=0)
    res = []
    for r in range(world):
        for i in range(counts[r]):
            res.append(unpack_record(h[r, i], kd, edim))
    return res


def spawn_local_ranks(script, argv, n, extra_env=None):
    """Starts ``n`` rank processes of ``script`` on this node (one per GPU) and returns the job's exit code.

    For ``bench.py --gpus N`` / ``tools/run_sequence.py --gpus N`` started WITHOUT a launcher (no WORLD_SIZE in the
    environment).  Plain child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, the
    rendezvous on 127.0.0.1; the calling process must not have touched the GPU (it only relays: rank 0's stdout is
    passed through line by line, the other ranks' stdout goes to stderr).  Any rank failing ends the job: the
    remaining children are terminated (by pid) and the first non-zero exit code is returned."""
    import os
    import socket
    import subprocess
    import sys
    import time
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    def relay():
        for line in procs[0].stdout:             # ends when rank 0 closes its stdout (exit)
            sys.stdout.write(line)
            sys.stdout.flush()
    import threading
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:                # a dead rank leaves the others blocked in a collective
                    q.terminate()
        time.sleep(0.05)
    th.join(timeout=5)
    return rc
