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
_ROW = sum(w for _, _, w in _FIELDS)          # 13 floats per detection in front of the closest-point table and the embedding


def shard_frames(n_frames, rank, world):
    """Contiguous frame range [lo, hi) of this rank."""
    per = (n_frames + world - 1) // world
    lo = min(rank * per, n_frames)
    return lo, min(lo + per, n_frames)


def record_len(n, edim=128):
    """Floats of the compact record of a frame with ``n`` detections: count, n x 13 fields, n x n x 2 closest points,
    n x edim embeddings -- 597 floats (2.4 KB) at n = 4, 1257 (5 KB) at n = 8, 34 101 (136 KB) at the n = 100 cap."""
    return 1 + n * _ROW + n * n * 2 + n * edim


def pack_record(rec, kd=100, edim=128):
    """Record -> count-prefixed flat f32 vector (the wire format of the gather; SURVEY 8e: "fixed-size record per frame" is
    relaxed to a count-prefixed one so a frame costs what it holds).  Every integer field is < 2^24: exact in f32."""
    n = len(rec["scores"])
    if n > kd:
        raise ValueError("record with %d detections, cap %d" % (n, kd))
    out = np.empty(record_len(n, edim), np.float32)
    out[0] = n
    o = 1
    for name, _, width in _FIELDS:
        out[o:o + n * width] = np.asarray(rec[name]).reshape(n * width)
        o += n * width
    out[o:o + n * n * 2] = np.asarray(rec["closest"], np.float32).reshape(n * n * 2)
    o += n * n * 2
    out[o:o + n * edim] = np.asarray(rec["embeddings"], np.float32).reshape(n * edim)
    return out


def unpack_record(vec, kd=100, edim=128):
    """One compact record (a vector that starts with it) -> record dict; ``rec["wire_floats"]`` = floats consumed."""
    n = int(vec[0])
    if n < 0 or n > kd or record_len(n, edim) > len(vec):
        raise ValueError("corrupt record: count %d, %d floats left" % (n, len(vec)))
    rec = {}
    o = 1
    for name, dt, width in _FIELDS:
        a = vec[o:o + n * width].reshape(n, width).astype(dt)
        rec[name] = a[:, 0] if width == 1 else a
        o += n * width
    rec["closest"] = vec[o:o + n * n * 2].reshape(n, n, 2).astype(np.int32)
    o += n * n * 2
    rec["embeddings"] = vec[o:o + n * edim].reshape(n, edim).astype(np.float32).copy()
    rec["packed_index"] = np.arange(n)
    rec["wire_floats"] = o + n * edim
    return rec


def unpack_stream(flat, nrec, kd=100, edim=128):
    """``nrec`` compact records laid end to end -> list of record dicts."""
    out, o = [], 0
    for _ in range(nrec):
        r = unpack_record(flat[o:], kd, edim)
        o += r.pop("wire_floats")
        out.append(r)
    return out


def gather_records(records, rank, world, device, kd=100, edim=128, unpack=True):
    """All ranks call this once; rank 0 gets the records of every rank in rank order (frame order for contiguous shards),
    the others get None.  Ranks may hold different numbers of frames and frames different numbers of detections.

    The exchange step of the path (visualize_uav.py:186-233 run frame-sharded): each rank lays its compact records end to
    end; a 16-byte all-gather tells every rank the (frames, floats) of the others, then ONE ``dist.gather`` moves the
    payloads to rank 0 only (RCCL over xGMI with the "nccl" backend, gloo on CPU) -- a 2 734-frame run on 8 GPUs at N = 8
    moves 1.7 MB per rank instead of the 47 MB per rank to every rank that worst-case records cost.
    ``unpack=False``: (flat f32 array, number of records) for ``NativeReplay.run_packed``."""
    import torch.distributed as dist
    flat = np.concatenate([pack_record(r, kd, edim) for r in records]) if records else np.zeros((0,), np.float32)
    mine = torch.tensor([len(records), flat.size], device=device, dtype=torch.int64)
    sizes = torch.zeros((world * 2,), device=device, dtype=torch.int64)
    dist.all_gather_into_tensor(sizes, mine)
    sizes = sizes.cpu().view(world, 2).tolist()
    mx = max(max(int(s[1]) for s in sizes), 1)
    buf = torch.zeros((mx,), dtype=torch.float32)
    buf[:flat.size] = torch.from_numpy(flat)
    buf = buf.to(device)
    import os
    if os.environ.get("APSE_GATHER", "gather") == "all_gather":
        # diagnostic switch: the same payload with all_gather_into_tensor (every rank receives it) instead of a gather to rank 0
        allb = torch.empty((world * mx,), dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(allb, buf)
        parts = list(allb.view(world, mx)) if rank == 0 else None
    else:
        parts = [torch.empty((mx,), dtype=torch.float32, device=device) for _ in range(world)] if rank == 0 else None
        dist.gather(buf, parts, dst=0)
    if rank != 0:
        return None
    h = np.concatenate([parts[r].cpu().numpy()[:int(sizes[r][1])] for r in range(world)]) if world else flat
    nrec = sum(int(s[0]) for s in sizes)
    if not unpack:
        return h, nrec
    return unpack_stream(h, nrec, kd, edim)


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
