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
--preproc runs preprocess_img (undistort + Lab gamma, visualize_uav.py:56-71) fused into the resize; --given-boxes feeds the
synthetic vehicles' boxes through the reference's detected_instances entry (track_rcnn.py:52-54).  BASELINE configs[2] / [3]:
    python tools/run_sequence.py [--gpus 8] --kind dynamic --frames 64 --dtype bf16 --batch 4 --preproc [--given-boxes]
Every rank detects its contiguous frame range on the software-pipelined loop; ONE gather of compact per-frame records to
rank 0, which replays the association in frame order and writes the same CSV a single-GPU run writes.
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


def load_camera(width, path=None):
    """Camera of preprocess_img (visualize_uav.py:56-71): the reference's data/cam_params.json (fixture copy under
    tests/golden/), its matrix scaled to the frame width like bench.py does for frames that are not 3840 wide."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(path or os.path.join(root, "tests", "golden", "cam_params.json")) as fh:
        cam = json.load(fh)
    sc = width / 3840.0
    cam["mtx"] = [[v * sc for v in cam["mtx"][0]], [v * sc for v in cam["mtx"][1]], cam["mtx"][2]]
    return cam


def synthetic_given_fn(seq, H, W, min_size=800, max_size=1333):
    """Given-boxes mode (TrackRCNN.inference(detected_instances=...), dcnn/networks/track_rcnn.py:52-54): the synthetic
    vehicles' boxes of frame t, class 0, in RESIZED-image pixels -> (boxes, classes, counts) of a batch of frame indices.  The box
    branch is skipped; masks, centroids, closest points, embeddings and ids are then a deterministic function of the frames."""
    from apse_uav_amd.utils import resample
    ih, iw = resample.resize_shortest_edge(H, W, min_size, max_size)
    sc = np.array([iw / W, ih / H, iw / W, ih / H], np.float32)

    def given(ts):
        per = [seq.boxes(t) * sc for t in ts]
        boxes = np.concatenate(per).astype(np.float32).reshape(-1, 4) if per else np.zeros((0, 4), np.float32)
        return boxes, np.zeros(len(boxes), np.int32), np.asarray([len(b) for b in per], np.int32)
    return given


def detect_range(tracker, get_frame, lo, hi, batch, given_fn=None, stats=None):
    """Stateless GPU part for frames [lo, hi): returns their records.  The loop is the software-pipelined single-stream loop of
    bench.py / TrackPredictor run-ahead: batch k + 1 is uploaded (copy stream), resized and its whole network enqueued BEHIND
    batch k's results copy (apse_read_results_begin), then the host waits for that copy only (_end) and builds batch k's records
    while the card is already on batch k + 1.  Same bits as the plain loop (tests/test_gpu_ingest.py).  ``stats`` (dict) receives
    the frame count and wall time."""
    import time
    pred = tracker.predictor
    model = pred.model
    spans = [(t, min(batch, hi - t)) for t in range(lo, hi, batch)]
    recs = []

    def stage(k):
        t, n = spans[k]
        frames = [get_frame(j) for j in range(t, t + n)]
        pred.prefetch(frames)                 # host copy + H2D on the copy stream: overlaps the forward that is running
        dev = pred._upload(frames)
        model.preprocess_frames(dev)
        pred._frames_consumed()
        model.run(n, given_fn(range(t, t + n)) if given_fn else None)

    t0 = time.perf_counter()
    if spans:
        stage(0)
    for k, (t, n) in enumerate(spans):
        model.read_begin(n)
        if k + 1 < len(spans):
            stage(k + 1)
        res = model.read_end(n)
        recs += [res.record(b) for b in range(n)]
    if stats is not None:
        stats["frames"] = hi - lo
        stats["seconds"] = time.perf_counter() - t0
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
    ap.add_argument("--preproc", action="store_true",
                    help="undistort + Lab gamma of preprocess_img (visualize_uav.py:56-71,191) fused into the resize, camera from "
                         "--cam-params (BASELINE configs[2] / [3]: --dtype bf16 --batch 4 --preproc)")
    ap.add_argument("--cam-params", default="", help="camera JSON in the layout of the reference's data/cam_params.json "
                                                     "(default: the fixture copy tests/golden/cam_params.json)")
    ap.add_argument("--given-boxes", action="store_true",
                    help="synthetic sequences only: feed the vehicles' boxes as detected_instances (track_rcnn.py:52-54), skipping the "
                         "RPN / box branch -- the deterministic form of configs 1-4 (SURVEY 8d)")
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
    given_fn = None
    if args.given_boxes:
        if args.images:
            sys.exit("run_sequence.py: --given-boxes needs the synthetic sequence (it knows its vehicles' boxes)")
        from apse_uav_amd.synthetic import SyntheticSequence
        given_fn = synthetic_given_fn(SyntheticSequence(args.kind, H, W), H, W, cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST)
    cam = load_camera(W, args.cam_params or None) if args.preproc else None
    if world == 1 and args.in_flight > 1:
        import time
        from apse_uav_amd.engines.pipelined_tracker import PipelinedRcnnTracker
        assert args.batch == 1 and not given_fn, "--in-flight drives batch-1 frames through the detector"
        drv = PipelinedRcnnTracker(cfg, (H, W), asd, depth=args.in_flight, detector_state=sd)
        if cam is not None:
            drv.set_camera(cam)
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
    if cam is not None:
        tracker.predictor.set_camera(cam)
    lo, hi = shard_frames(n - first, rank, world)
    stats = {}
    recs = detect_range(tracker, get_frame, first + lo, first + hi, args.batch, given_fn, stats)
    print("rank %d: frames [%d, %d) in %.3f s = %.1f frames/s incl. frame generation / decode and upload"
          % (rank, first + lo, first + hi, stats["seconds"], (hi - lo) / max(stats["seconds"], 1e-9)), file=sys.stderr)
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
