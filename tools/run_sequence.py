#!/usr/bin/env python3
"""Sequence driver: the counterpart of /root/reference/dcnn/scripts/tests/visualize_uav.py:156-235
(video loop -> tracker.next_frame -> generate_log_oneline -> log_file.csv), without the GUI / PNG output.

Frames come from the synthetic generator (no video ships with the reference) or from a directory of
images readable by PIL.  Single process:   python tools/run_sequence.py --frames 64 --out seq.csv
Frame-sharded over N GPUs (BASELINE config 4), either form:
    python tools/run_sequence.py --gpus N --frames 64 --out seq.csv          (starts the N ranks itself)
    torchrun --nproc-per-node N tools/run_sequence.py --gpus N --frames 64 --out seq.csv
--start-frame S skips the first S frames like the reference loop does (visualize_uav.py:172-190): they are read and
dropped, the CSV's frame column keeps the absolute frame index.
Every rank detects its contiguous frame range; ONE gather of per-frame records; rank 0 replays the
association in frame order and writes the same CSV a single-GPU run writes.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def frame_source(args, H, W):
    if args.images:
        from PIL import Image
        names = sorted(n for n in os.listdir(args.images) if n.split(".")[-1].lower() in ("jpg", "png", "bmp"))
        names = names[: args.frames] if args.frames else names
        return len(names), lambda t: np.ascontiguousarray(np.asarray(Image.open(os.path.join(args.images, names[t])).convert("RGB"))[:, :, ::-1])
    from apse_uav_amd.synthetic import SyntheticSequence
    seq = SyntheticSequence(args.kind, H, W)
    return args.frames, seq.frame


def detect_range(tracker, get_frame, lo, hi, batch, given_fn=None):
    """Stateless GPU part for frames [lo, hi): returns their records."""
    recs = []
    t = lo
    while t < hi:
        n = min(batch, hi - t)
        frames = [get_frame(k) for k in range(t, t + n)]
        given = given_fn(range(t, t + n)) if given_fn else None
        dev = tracker.predictor._upload(frames)
        model = tracker.predictor.model
        model.preprocess_frames(dev)
        model.run(n, given)
        res = model.read(n)
        recs += [res.record(b) for b in range(n)]
        t += n
    return recs


def replay(tracker, records, host_id, first_frame=0, fast=False):
    """Sequential id assignment over records.  fast=True: apse_uav_amd.engines.replay.NativeReplay (same rules,
    C++ in libapse_hip.so); default: the reference-shaped RcnnTracker.next_record path."""
    if fast:
        from apse_uav_amd.engines.replay import NativeReplay
        fr = NativeReplay(host_id)
        lines = [fr.step(rec, first_frame + k)[0] for k, rec in enumerate(records)]
        return lines, fr.max_id
    lines, max_id = [], 0
    for k, rec in enumerate(records):
        objs = tracker.next_record(rec)
        line, hi = tracker.log_line(objs, host_id, first_frame + k)
        lines.append(line)
        max_id = max(max_id, hi)
    return lines, max_id


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--kind", default="dynamic", choices=["static", "dynamic"])
    ap.add_argument("--size", default="2160x3840")
    ap.add_argument("--images", default=None)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--blocks", default="3,4,23,3")
    ap.add_argument("--weights", default="", help="detector .pth (default: seeded synthetic weights)")
    ap.add_argument("--assoc-weights", default="")
    ap.add_argument("--host-id", type=int, default=1)
    ap.add_argument("--vehicle-ids", default="2,3,4")
    ap.add_argument("--out", default="seq_dcnn_data.csv")
    ap.add_argument("--raw-out", default="")
    ap.add_argument("--gpus", type=int, default=1, help="ranks (one per GPU); without a launcher the ranks are started here")
    ap.add_argument("--start-frame", type=int, default=0,
                    help="first frame to process (visualize_uav.py:172-190 START_FROM_FRAME); earlier frames are skipped, "
                         "the CSV keeps absolute frame indices")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="single process: frames in flight on separate streams (engines/pipelined_tracker.py); 1 = plain loop")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:        # no launcher: start the ranks (nothing has touched the GPU yet)
        from apse_uav_amd.sharding import spawn_local_ranks
        sys.exit(spawn_local_ranks(os.path.abspath(__file__), sys.argv[1:] if argv is None else argv, args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("run_sequence.py: --gpus %d but WORLD_SIZE=%d ranks were started" % (args.gpus, world))
    backend = os.environ.get("APSE_DIST_BACKEND", "nccl")           # "nccl" is RCCL on ROCm; gloo: one-GPU rehearsal
    ndev = torch.cuda.device_count()
    if ndev < 1:
        sys.exit("run_sequence.py: no GPU visible (the apse_uav hot path has no CPU fallback)")
    if backend == "nccl" and world > ndev:
        sys.exit("run_sequence.py: --gpus %d needs %d GPUs, %d visible (one-GPU rehearsal: APSE_DIST_BACKEND=gloo)" % (world, world, ndev))
    local = local % ndev
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    from apse_uav_amd.config import setup_cfg
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from apse_uav_amd.sharding import gather_records, shard_frames
    from apse_uav_amd.utils import csv_log
    from apse_uav_amd.weights import (UAV4K_R101_CLS_BIAS, load_association_file, load_detector_file,
                                      synthetic_association_state, synthetic_detector_state)
    H, W = [int(v) for v in args.size.split("x")]
    blocks = tuple(int(v) for v in args.blocks.split(","))
    sd = load_detector_file(args.weights) if args.weights else synthetic_detector_state(
        0, blocks, cls_bias=UAV4K_R101_CLS_BIAS if blocks == (3, 4, 23, 3) else None)
    asd = load_association_file(args.assoc_weights) if args.assoc_weights else synthetic_association_state(1)
    cfg = setup_cfg(device="cuda:%d" % local)
    cfg.APSE.MAX_BATCH = args.batch
    cfg.APSE.DTYPE = args.dtype
    n, get_frame = frame_source(args, H, W)
    first = min(max(args.start_frame, 0), n)
    if world == 1 and args.in_flight > 1:
        import time
        from apse_uav_amd.engines.pipelined_tracker import PipelinedRcnnTracker
        assert args.batch == 1, "--in-flight drives batch-1 frames"
        drv = PipelinedRcnnTracker(cfg, (H, W), asd, depth=args.in_flight, detector_state=sd)
        lines, max_id = [], 0
        t0 = time.perf_counter()
        for t, objs in drv.run(get_frame(k) for k in range(first, n)):
            line, hi_id = drv.tracker.log_line(objs, args.host_id, first + t)
            lines.append(line)
            max_id = max(max_id, hi_id)
        dt = time.perf_counter() - t0
        csv_log.write_consumer_csv(args.out, lines, args.host_id, [int(v) for v in args.vehicle_ids.split(",")], first_frame=first)
        if args.raw_out:
            csv_log.write_raw_csv(args.raw_out, lines, args.host_id, max_id)
        print("wrote %s: %d frames, %d track ids, %d frames in flight, %.1f frames/s incl. frame generation and upload"
              % (args.out, len(lines), max_id, args.in_flight, (n - first) / dt))
        return
    tracker = RcnnTracker(cfg, (H, W), asd, detector_state=sd)
    lo, hi = shard_frames(n - first, rank, world)
    recs = detect_range(tracker, get_frame, first + lo, first + hi, args.batch)
    if world > 1:
        recs = gather_records(recs, rank, world, torch.device("cuda", local) if backend == "nccl" else torch.device("cpu"))
    if rank == 0:
        lines, max_id = replay(tracker, recs, args.host_id, first_frame=first, fast=(world > 1))
        csv_log.write_consumer_csv(args.out, lines, args.host_id, [int(v) for v in args.vehicle_ids.split(",")], first_frame=first)
        if args.raw_out:
            csv_log.write_raw_csv(args.raw_out, lines, args.host_id, max_id)
        print("wrote %s: %d frames, %d track ids" % (args.out, len(lines), max_id))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
