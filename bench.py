#!/usr/bin/env python3
"""Benchmark of the dcnn hot path on MI355X: 4K UAV frames/s (whole job) + p50 per-frame latency.

Workload (BASELINE.json configs[1]): synthetic "static" 3840x2160 sequence, batch 1, f32, Mask R-CNN
R-101-FPN with seeded synthetic weights (no weights or video ship with the reference), frames resident in
HBM before the timed region.  One step = one batch through the whole per-frame path: PIL-exact resize +
normalise, backbone + FPN, RPN + proposal selection, box head + NMS, mask head + paste + centroid/closest
points, roi_pool + association embedding, D2H of the results block, host association (Hungarian) and the
CSV line.  N > 1: one process per GPU, frames sharded by rank (weak scaling), a single gather of the
per-frame records to rank 0 after the last step, which then runs the sequential id assignment.

Prints ONE JSON line (metric, value, unit, n_gpus, steps, warmup, ms_per_step, scaling, dtype, data, config, ...), plus
  roofline     : the dominant kernel's algorithmic FLOP/s from HIP events recorded (inside libapse_hip.so, on the launch stream)
                 around every convolution launch of a SEPARATE pass of --probe-steps steps right after the timed region
                 -- the timed region itself carries no instrumentation;
  modes        : (default N = 1 invocation) short un-instrumented runs of BASELINE configs[2] (bf16, batch 4, fused undistort +
                 gamma) and configs[4] on one GPU (fp16, batch 8), each with its MFMA and HBM whole-path roofline fractions;
  cpu_baseline : the CPU oracle (oracle/, PyTorch CPU f32) timed on this host, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG_NAMES = ["conv_igemm<128x128>", "conv_igemm<64x64>", "conv_igemm<128x32>", "conv_igemm<128x64>",
             "conv_igemm<64x64,k32>", "conv_igemm<128x32,k32>", "conv_igemm<64x64,8 waves,k64>", "conv_igemm<64x64,8 waves,k128>",
             "conv_igemm<256x128>", "conv1x1_stream", "bottleneck64_fused16", "conv_glds16<256x128>", "stem_s2d_pool16", "conv_skinny16"]
# template arguments <WM, WN, TM, TN, KS, XT, WK, PR> of conv_igemm_f32 behind each tile shape (f32 path, f32 activations):
# the kernel names rocprofv3 reports, used to look the dominant kernel up in the committed PMC summary
CFG_TEMPLATE = ["<2, 2, 2, 2, 1, 0, 1, 0>", "<2, 2, 1, 1, 2, 0, 1, 0>", "<4, 1, 1, 1, 2, 0, 1, 0>", "<4, 1, 1, 2, 1, 0, 1, 0>",
                "<2, 2, 1, 1, 1, 0, 1, 0>", "<4, 1, 1, 1, 1, 0, 1, 0>", "<2, 2, 1, 1, 2, 0, 2, 0>", "<2, 2, 1, 1, 4, 0, 2, 0>", None, None, None, None, None, None]
NCFG = len(CFG_NAMES)
PMC_TRAFFIC_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic_latest.json")
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: bf16 dense


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--frame", type=str, default="2160x3840")
    ap.add_argument("--blocks", type=str, default="3,4,23,3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=3)
    ap.add_argument("--no-events", action="store_true", help="skip the per-kernel HIP-event pass (no roofline block)")
    ap.add_argument("--probe-steps", type=int, default=3,
                    help="steps of the SEPARATE instrumented pass that follows the timed region: HIP-event pairs around every "
                         "convolution launch (each pair costs ~5 us of stream time, ~1.5 ms per instrumented frame, which is why "
                         "the timed region itself carries none)")
    ap.add_argument("--no-prestage", action="store_true",
                    help="plain loop (resize, network, read, host work strictly one after the other) instead of the software-pipelined "
                         "single-stream loop")
    ap.add_argument("--no-extra-modes", action="store_true",
                    help="skip the short runs of BASELINE configs[2] (bf16, batch 4, fused undistort + gamma) and configs[4] "
                         "(fp16, batch 8) that the default N = 1 invocation appends as `modes`")
    ap.add_argument("--mode-steps", type=int, default=12)
    ap.add_argument("--cpu-threads", type=int, default=64, help="cap on the CPU baseline's all-cores run (further capped by the container's CPU quota)")
    ap.add_argument("--from-host", action="store_true",
                    help="frames start in pinned host memory and are uploaded inside the timed region on a copy stream "
                         "(double-buffered, overlapped with compute): the PCIe-inclusive rate noted in DESIGN.md, never "
                         "the headline value")
    ap.add_argument("--throughput-depth", type=int, default=4,
                    help="after the timed region, also measure a pipelined run with this many frames in flight and "
                         "report it as `throughput_mode` (0 = skip)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="frames in flight per GPU: >1 runs consecutive steps on separate HIP streams / contexts so the "
                         "small-grid layers of one frame overlap with the next frame's (detection is stateless per "
                         "frame; the host association still consumes frames in order)")
    ap.add_argument("--dtype", type=str, default="f32", choices=["f32", "bf16", "f16"],
                    help="f32 = exact-f32 MFMA (reference numerics, BASELINE configs[1]); bf16 = bf16 matrix cores with "
                         "f32 accumulate/storage for the bulk GEMMs (configs[2]/[4] style)")
    ap.add_argument("--bg-bias", type=float, default=None)
    ap.add_argument("--preproc", action="store_true",
                    help="undistort (data/cam_params.json model, fixture copy in tests/golden/) + Lab gamma fused into the resize "
                         "(BASELINE configs[2] style: --dtype bf16 --batch 4 --preproc)")
    ap.add_argument("--no-entrypoint", action="store_true", help="skip the RcnnTracker.next_frame(np.ndarray) measurement")
    ap.add_argument("--entry-steps", type=int, default=24)
    ap.add_argument("--rehearse-spawn", action="store_true",
                    help="no GPU work: rendezvous, the record gather and the max-over-ranks timing only (CPU test of the "
                         "--gpus N launch path; use with APSE_DIST_BACKEND=gloo)")
    args = ap.parse_args()

    # ---- N ranks: one process per GPU.  Under a launcher (torchrun: WORLD_SIZE set) this process IS a rank; without one
    # `--gpus N` starts the N ranks itself -- before anything touches the GPU -- and only relays rank 0's JSON line.
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from apse_uav_amd.sharding import spawn_local_ranks
        sys.exit(spawn_local_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; pass --gpus %d (the reported n_gpus "
                 "must be the number of ranks that ran)" % (args.gpus, world, world))
    backend = os.environ.get("APSE_DIST_BACKEND", "nccl")       # "nccl" is RCCL on ROCm
    if args.rehearse_spawn:
        return rehearse_spawn(args, rank, world, backend)
    ndev = torch.cuda.device_count()
    if ndev < 1:
        sys.exit("bench.py: no GPU visible (the apse_uav hot path has no CPU fallback)")
    if backend == "nccl" and world > ndev:
        sys.exit("bench.py: --gpus %d needs %d GPUs on this node, %d visible (a one-GPU rehearsal of the launch path: "
                 "APSE_DIST_BACKEND=gloo)" % (world, world, ndev))
    dev_index = local_rank % ndev            # rehearsals may put several ranks on one GPU (APSE_DIST_BACKEND=gloo)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    else:
        dist = None
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    torch.cuda.set_device(dev)

    from apse_uav_amd import _lib
    from apse_uav_amd.config import setup_cfg
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from apse_uav_amd.utils.hostinfo import usable_cpus
    # host threads: torch's intra-op pool defaults to every core it sees (256 on the GPU box) while the container's CPU
    # quota is 16; oversubscribed, spinning OpenMP workers get the process throttled.  Size the pool by the real share.
    torch.set_num_threads(max(1, min(torch.get_num_threads(), usable_cpus() // max(world, 1) or 1)))
    from apse_uav_amd.sharding import gather_records
    from apse_uav_amd.synthetic import SyntheticSequence
    from apse_uav_amd.weights import synthetic_association_state, synthetic_detector_state

    H, W = [int(v) for v in args.frame.split("x")]
    blocks = tuple(int(v) for v in args.blocks.split(","))
    B = args.batch
    from apse_uav_amd.weights import UAV4K_R101_CLS_BIAS
    if args.bg_bias is None and blocks == (3, 4, 23, 3):
        sd = synthetic_detector_state(0, blocks, cls_bias=UAV4K_R101_CLS_BIAS)
    else:
        sd = synthetic_detector_state(0, blocks, bg_bias=args.bg_bias or 0.0)
    asd = synthetic_association_state(1)
    cfg = setup_cfg(device="cuda:%d" % dev_index)
    cfg.APSE.MAX_BATCH = B
    cfg.APSE.DTYPE = args.dtype
    tracker = RcnnTracker(cfg, (H, W), asd, detector_state=sd)
    model = tracker.predictor.model
    if args.preproc:
        with open(os.path.join(ROOT, "tests", "golden", "cam_params.json")) as fh:
            cam = json.load(fh)
        sc = W / 3840.0
        cam["mtx"] = [[v * sc for v in cam["mtx"][0]], [v * sc for v in cam["mtx"][1]], cam["mtx"][2]]
        tracker.predictor.set_camera(cam)

    # frames of this rank's shard, resident in HBM before timing (8 distinct frames, cycled)
    seq = SyntheticSequence("static", H, W)
    nres = 8
    host_frames = [seq.frame(rank * 1000 + i) for i in range(nres)]
    frames = torch.stack([torch.from_numpy(f) for f in host_frames]).to(dev)

    lib = _lib.load()
    records = []
    from apse_uav_amd.engines.replay import NativeReplay
    replay = NativeReplay(host_id=1)        # same association rules as RcnnTracker.next_record (tests/test_replay.py)

    # optional software pipeline over frames: depth contexts on depth streams (weights replicated)
    depth = max(1, args.pipeline)
    models = [model]
    if depth > 1:
        from apse_uav_amd.networks.track_rcnn import TrackRCNN
        for _ in range(depth - 1):
            m2 = TrackRCNN(cfg)
            m2.load_state_dict(sd)
            m2.attach_association_head(tracker.association_head)
            m2.set_camera(model._camera)
            models.append(m2)
    # every slot of a software pipeline gets its own non-default stream
    streams = [torch.cuda.current_stream()] if depth == 1 else [torch.cuda.Stream() for _ in range(depth)]

    pinned = torch.stack([torch.from_numpy(f) for f in host_frames]).pin_memory() if args.from_host else None
    copy_stream = torch.cuda.Stream() if args.from_host else None
    upload = {}

    def prefetch(i):
        """H2D of step i's frames on the copy stream (upload overlapped with compute)."""
        idx = [(i * B + j) % nres for j in range(B)]
        with torch.cuda.stream(copy_stream):
            d = torch.stack([pinned[j] for j in idx]).to(dev, non_blocking=True) if B > 1 else pinned[idx[0]:idx[0] + 1].to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        upload[i] = (d, ev)

    def submit(i):
        k = i % depth
        idx = [(i * B + j) % nres for j in range(B)]
        if args.from_host:
            if i not in upload:
                prefetch(i)
            batch, ev = upload.pop(i)
            streams[k].wait_event(ev)
            batch.record_stream(streams[k])       # allocated on the copy stream, consumed here: keep the allocator off it until then
            prefetch(i + 1)                       # next step's upload overlaps this step's compute
        else:
            # frames of a step that are consecutive in HBM are handed over as a view (a gather of B 4K frames is 2 x 25 MB
            # of copies per frame that a real ingest never makes: frames arrive one by one into the batch buffer)
            batch = frames[idx[0]:idx[0] + B] if idx == list(range(idx[0], idx[0] + B)) else frames[idx]
        with torch.cuda.stream(streams[k]):
            models[k].preprocess_frames(batch)
            models[k].run(B)

    def collect(i, timed):
        k = i % depth
        with torch.cuda.stream(streams[k]):
            res = models[k].read(B)                   # D2H of the results block + sync of that stream
        for b in range(B):
            rec = res.record(b)
            if world == 1:
                replay.step(rec, i * B + b)
            elif timed:
                records.append(rec)
        return res

    def step(i, timed):
        submit(i)
        return collect(i, timed)

    # a context is built on its first forward (~0.2 s: weight packing + upload): prime every pipeline slot before the W warm-up
    # steps, or slots beyond W are created INSIDE the timed region (round 1's "depth 6 collapses to 45 frames/s")
    for k in range(1, depth):
        with torch.cuda.stream(streams[k]):
            models[k].preprocess_frames(frames[0:B])
            models[k].run(B)
            models[k].read(B)
    for i in range(args.warmup):
        res = step(i, False)
    torch.cuda.synchronize()
    if dist is not None:
        # warm-up of the exchange step too: the first collective of a size creates RCCL channels / buffers (hundreds of ms), which is
        # start-up cost like the context build, not part of a frame
        gather_records([], rank, world, coll_dev, unpack=False)
    lib.apse_profile(model._ctx, 0)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    lat = []
    lat_enq = []
    P_sum = N_sum = 0
    n_instr = 0
    t0 = time.perf_counter()
    def account(res):
        nonlocal P_sum, N_sum
        P_sum += int(res.prop_count[:B].sum())
        N_sum += res.total

    inflight = []          # (step index, submit time)
    trace = [] if os.environ.get("APSE_BENCH_TRACE") else None      # (submit ms, collect ms) per step -> stderr
    # Single stream, one context (the headline form): the loop is software-pipelined the way TrackPredictor runs it with an announced
    # next frame -- frame i + 1's resize + normalise is enqueued BEHIND frame i's results copy (apse_read_results_begin / _end: the
    # copy is waited for by event, not the stream) and frame i + 1's network right behind that (it needs nothing of frame i; stream order
    # keeps the copy in front of what it overwrites), so the card never waits for the host; the host association + CSV line of frame i
    # run while the GPU already works on frame i + 1.  Every frame's work is inside the
    # timed region; nothing runs concurrently on the GPU; results are the same bits (tests/test_gpu_ingest.py).
    fast = depth == 1 and not args.from_host and not args.no_prestage
    if fast:
        m0 = models[0]

        def frames_of(i):
            idx = [(i * B + j) % nres for j in range(B)]
            return frames[idx[0]:idx[0] + B] if idx == list(range(idx[0], idx[0] + B)) else frames[idx]

        def post(i, res):
            for b in range(B):
                rec = res.record(b)
                if world == 1:
                    replay.step(rec, i * B + b)
                else:
                    records.append(rec)
            account(res)
        first = args.warmup
        ts = time.perf_counter()
        m0.preprocess_frames(frames_of(first))
        t_enq = time.perf_counter()              # when this frame's network was handed to the stream
        m0.run(B)
        for i in range(args.steps):
            j = args.warmup + i
            m0.read_begin(B)
            more = i + 1 < args.steps
            if more:
                m0.preprocess_frames(frames_of(j + 1))      # behind frame j's network and its results copy
                ts_next = time.perf_counter()
                t_enq_next = ts_next
                m0.run(B)                                   # frame j + 1's network right behind them: the stream keeps frame j's
                                                            # results copy in front of everything this forward overwrites, and the
                                                            # card does not idle while the host wakes up on the copy's event
            res = m0.read_end(B)
            t_ready = time.perf_counter()
            t_post = time.perf_counter()
            post(j, res)
            # per-frame latency: from the moment the card is free for this frame (its network was enqueued while the previous frame
            # was still running, so: the later of that enqueue and the previous frame's results) to its results on the host, + this
            # frame's host association / CSV line (which runs while the GPU is already on the next frame)
            lat.append((t_ready - ts) + (time.perf_counter() - t_post))
            lat_enq.append((t_ready - t_enq) + (time.perf_counter() - t_post))      # what a caller of this loop sees: enqueue -> results + host part
            if more:
                ts = max(ts_next, t_ready)
                t_enq = t_enq_next
    for i in range(0 if fast else args.steps):     # the timed region carries NO instrumentation (no HIP events, no profiling calls)
        inflight.append((args.warmup + i, time.perf_counter()))
        t_a = time.perf_counter()
        submit(args.warmup + i)
        t_b = time.perf_counter()
        if len(inflight) == depth:
            j, ts = inflight.pop(0)
            account(collect(j, True))
            lat.append(time.perf_counter() - ts)
        if trace is not None:
            trace.append((round(1e3 * (t_b - t_a), 2), round(1e3 * (time.perf_counter() - t_b), 2)))
    while inflight:
        j, ts = inflight.pop(0)
        account(collect(j, True))
        lat.append(time.perf_counter() - ts)
    if trace is not None:
        print("submit/collect ms per step:", trace, file=sys.stderr)
    if dist is not None:
        packed = gather_records(records, rank, world, coll_dev, unpack=False)   # the single exchange step (RCCL over xGMI)
        if rank == 0:
            replay.run_packed(packed, kd=100)                              # (flat, nrec): sequential id assignment + CSV lines (C++)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # ---- separate instrumented pass (NOT part of `value`): HIP-event pairs around every convolution launch, on the stream the
    # kernels run on (recorded inside libapse_hip.so), for the roofline block
    if not args.no_events:
        for i in range(max(args.probe_steps, 0)):
            k = (args.warmup + args.steps + i) % depth
            lib.apse_profile(models[k]._ctx, 1)
            step(args.warmup + args.steps + i, False)
            lib.apse_profile(models[k]._ctx, 0)
            n_instr += 1
        torch.cuda.synchronize()
    import ctypes as C
    prof = np.zeros((NCFG, 3))
    for m in models:
        pr = (C.c_double * (3 * NCFG))()
        lib.apse_profile_read(m._ctx, C.byref(pr), 1)
        lib.apse_profile(m._ctx, 0)
        prof += np.array(list(pr)).reshape(NCFG, 3)

    if rank == 0:
        frames_total = args.steps * B * world
        fps = frames_total / elapsed
        dom = int(np.argmax(prof[:, 0]))
        ms, fl, nl = prof[dom]
        achieved = (fl / (ms * 1e-3)) / 1e12 if ms > 0 else 0.0
        peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
        total_conv_ms = float(prof[:, 0].sum())
        # HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside this process, so this is
        # the committed summary of the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same workload
        # (tools/gpu_profile_round.sh; read side doubled per the gfx950 note of MI355X_MICROARCH.md)
        traffic, traffic_src = None, None
        build = lib.apse_version().decode()
        if args.dtype == "f32" and B == 1 and os.path.exists(PMC_TRAFFIC_FILE):
            try:
                with open(PMC_TRAFFIC_FILE) as fh:
                    pmj = json.load(fh)
                pm = pmj.get("conv_igemm_f32" + CFG_TEMPLATE[dom]) if CFG_TEMPLATE[dom] else None
                if pmj.get("__build__") != build:
                    traffic_src = ("null: profiles/pmc_traffic_latest.json was recorded for build '%s', the loaded library is "
                                   "'%s' (re-run tools/gpu_profile_round.sh)" % (pmj.get("__build__"), build))
                elif pm:
                    traffic = int(pm["fetch_bytes"] + pm["write_bytes"])
                    traffic_src = "profiles/pmc_traffic_latest.json (same build): rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes, bytes per launch"
            except (OSError, ValueError, KeyError):
                pass
        flops_frame = model.flops(1, P_sum / max(args.steps * B, 1), N_sum / max(args.steps * B, 1))
        out = {
            "metric": "4K UAV frames/sec (whole node)", "value": round(fps, 3), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic" + (" (uploaded from pinned host memory inside the timed region)" if args.from_host else ""),
            "p50_ms_per_frame": round(1000.0 * float(np.median(lat)) / B, 3),
            "p50_ms_enqueue_to_results": round(1000.0 * float(np.median(lat_enq)), 3) if lat_enq else None,
            "latency_definitions": ("p50_ms_per_frame: from the moment the card is free for a frame (the later of its enqueue and the previous "
                                    "frame's results) to its results on the host + its host association, per frame of the batch; "
                                    "p50_ms_enqueue_to_results: from the call that enqueued the frame's network to the same point, per batch -- "
                                    "in the software-pipelined loop a frame is enqueued one step ahead, so this is about two steps") if fast
                                   else "p50_ms_per_frame: submit -> results on the host + host association, per frame of the batch",
            "config": {"workload": "static synthetic 3840x2160 sequence, batch=%d %s, Mask R-CNN R-%s-FPN, %d GPU(s), "
                                   "frames sharded per rank" % (B, args.dtype, "101" if blocks == (3, 4, 23, 3) else str(blocks), world),
                       "loop": ("single stream, software-pipelined: next frame's resize and network enqueued behind this frame's results copy, host "
                                "association overlapped with the next frame's network" if fast else "plain"),
                       "frame": "%dx%d" % (W, H), "preproc": "undistort + gamma fused into the resize" if args.preproc else "none", "batch_per_gpu": B, "frames_in_flight": depth, "proposals_per_frame": P_sum / max(args.steps * B, 1),
                       "detections_per_frame": N_sum / max(args.steps * B, 1),
                       "gflop_per_frame_algorithmic": round(flops_frame / 1e9, 2)},
            "roofline": {"bound": "mfma", "kernel": CFG_NAMES[dom], "achieved": round(achieved, 3),
                         "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": round(ms / max(nl, 1), 5), "launches": int(nl),
                         "all_kernels": {CFG_NAMES[k]: {"ms": round(float(prof[k, 0]), 3),
                                                        "tflops": round(float(prof[k, 1] / max(prof[k, 0], 1e-9) / 1e9), 3),
                                                        "launches": int(prof[k, 2])} for k in range(NCFG) if prof[k, 2] > 0},
                         "conv_ms_per_frame": round(total_conv_ms / max(n_instr * B, 1), 3), "instrumented_steps": n_instr,
                         "measured": "separate pass of %d instrumented steps right after the timed region (the timed region has no "
                                     "events); HIP events recorded in-library on the launch stream, marker overhead calibrated out" % n_instr,
                         "whole_path_tflops": round(flops_frame * fps / world / 1e12, 3),
                         "whole_path_frac": round(flops_frame * fps / world / 1e12 / peak, 4)},
        }
        if world == 1 and depth == 1 and args.throughput_depth > 1 and not args.from_host:
            out["throughput_mode"] = throughput_mode(cfg, sd, tracker, model, frames, nres, B, args.throughput_depth, replay)
        if (world == 1 and depth == 1 and B == 1 and args.dtype == "f32" and not args.preproc and not args.from_host
                and not args.no_extra_modes and blocks == (3, 4, 23, 3)):
            out["modes"] = {}
            for tag, kw in (("configs[2]: bf16 batch 4, undistort + gamma fused", dict(dtype="bf16", batch=4, preproc=True)),
                            ("configs[4] on one GPU: f16 batch 8", dict(dtype="f16", batch=8, preproc=False, in_flight=3))):
                out["modes"][tag] = extra_mode(lib, sd, asd, frames, nres, H, W, dev_index, args.mode_steps, 0 if args.no_events else 2, **kw)
        out["build"] = build
        out["association"] = "C++ Hungarian + track store of csrc/replay.hip (NativeReplay; equal to scipy linear_sum_assignment on the tests' streams)"
        if world == 1 and depth == 1 and B == 1 and not args.no_entrypoint:
            out["entrypoint"] = entrypoint_mode(tracker, host_frames, args.entry_steps)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, asd, host_frames, blocks, H, W, args.cpu_frames, args.cpu_threads)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


# algorithmic-minimum HBM traffic of one 3840x2160 frame through R-101-FPN: every convolution output written
# once and read once (390.2 M elements) + the ~63 M weights once per BATCH
ACT_ELEMS_4K_R101 = 390.2e6
WEIGHT_ELEMS_R101 = 63.0e6
HBM_ACHIEVABLE_TBS = 6.29             # MI355X_MICROARCH.md: measured copy bandwidth (8.0 TB/s data sheet)
SUSTAINED_16BIT_TFLOPS = 1810.0       # tools/micro/mfma_peak.hip on random operands (profiles/r01b_mfma_peak.txt)


def extra_mode(lib, sd, asd, frames, nres, H, W, dev_index, steps, probe_steps, dtype, batch, preproc, warmup=3, in_flight=1):
    """A short un-instrumented run of another BASELINE configuration on the same resident frames: its own context
    (`batch` frames per step, 16-bit matrix cores + 16-bit activation storage, optionally the fused undistort + gamma), the
    whole per-frame path incl. D2H of the results block and the host association, timed like the headline (synchronise,
    `steps` steps, synchronise).  Reports BOTH roofline fractions of the whole path: MFMA (algorithmic FLOP/s / the dense
    2.5 PFLOP/s peak, and / the 1.81 PFLOP/s the card sustains on random operands) and HBM (SURVEY 8d's algorithmic bytes /
    time / 6.29 TB/s).  A separate instrumented pass names the dominant kernel."""
    import ctypes as C
    from apse_uav_amd.config import setup_cfg
    from apse_uav_amd.engines.rcnn_tracker import RcnnTracker
    from apse_uav_amd.engines.replay import NativeReplay
    cfg = setup_cfg(device="cuda:%d" % dev_index)
    cfg.APSE.MAX_BATCH = batch
    cfg.APSE.DTYPE = dtype
    tr = RcnnTracker(cfg, (H, W), asd, detector_state=sd)
    model = tr.predictor.model
    if preproc:
        with open(os.path.join(ROOT, "tests", "golden", "cam_params.json")) as fh:
            cam = json.load(fh)
        sc = W / 3840.0
        cam["mtx"] = [[v * sc for v in cam["mtx"][0]], [v * sc for v in cam["mtx"][1]], cam["mtx"][2]]
        tr.predictor.set_camera(cam)
    replay = NativeReplay(host_id=1)
    P = N = 0

    def step(i, count):
        nonlocal P, N
        lo = (i * batch) % nres
        b = frames[lo:lo + batch] if lo + batch <= nres else frames[[(lo + j) % nres for j in range(batch)]]
        model.preprocess_frames(b)
        model.run(batch)
        res = model.read(batch)
        for k in range(batch):
            replay.step(res.record(k), i * batch + k)
        if count:
            P += int(res.prop_count[:batch].sum())
            N += res.total
    lib.apse_profile(model._ctx, 0)
    for i in range(warmup):
        step(i, False)
    torch.cuda.synchronize()
    lat = []

    def batch_of(i):
        lo = (i * batch) % nres
        return frames[lo:lo + batch] if lo + batch <= nres else frames[[(lo + j) % nres for j in range(batch)]]
    t0 = time.perf_counter()
    # the same software-pipelined single-stream loop as the headline (next batch's resize behind this batch's results copy)
    ts = time.perf_counter()
    model.preprocess_frames(batch_of(warmup))
    model.run(batch)
    for i in range(steps):
        model.read_begin(batch)
        more = i + 1 < steps
        if more:
            model.preprocess_frames(batch_of(warmup + i + 1))
            ts_next = time.perf_counter()
            model.run(batch)
        res = model.read_end(batch)
        t_ready = time.perf_counter()
        t_post = time.perf_counter()
        for k in range(batch):
            replay.step(res.record(k), (warmup + i) * batch + k)
        P += int(res.prop_count[:batch].sum())
        N += res.total
        lat.append((t_ready - ts) + (time.perf_counter() - t_post))
        if more:
            ts = max(ts_next, t_ready)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nfr = steps * batch
    fps = nfr / dt
    flops_frame = model.flops(1, P / nfr, N / nfr)
    tfl = flops_frame * fps / 1e12
    out = {"value": round(fps, 3), "unit": "frames/s", "steps": steps, "warmup": warmup, "batch": batch, "dtype": dtype,
           "preproc": "undistort + gamma fused into the resize" if preproc else "none",
           "ms_per_step": round(1e3 * dt / steps, 3), "p50_ms_per_frame": round(1e3 * float(np.median(lat)) / batch, 3),
           "proposals_per_frame": P / nfr, "detections_per_frame": N / nfr,
           "whole_path_tflops": round(tfl, 2),
           "mfma_frac_of_dense_peak_2500": round(tfl / PEAK_BF16_MFMA_TFLOPS, 4),
           "mfma_frac_of_sustained_1810": round(tfl / SUSTAINED_16BIT_TFLOPS, 4)}
    if (H, W) == (2160, 3840):
        bytes_frame = 2.0 * ACT_ELEMS_4K_R101 * 2 + WEIGHT_ELEMS_R101 * 2 / batch
        out["hbm_algorithmic_bytes_per_frame"] = int(bytes_frame)
        out["hbm_tbs_algorithmic"] = round(bytes_frame * fps / 1e12, 3)
        out["hbm_frac_of_achievable_6.29"] = round(bytes_frame * fps / 1e12 / HBM_ACHIEVABLE_TBS, 4)
    if probe_steps > 0:
        for i in range(probe_steps):
            lib.apse_profile(model._ctx, 1)
            step(warmup + steps + i, False)
        lib.apse_profile(model._ctx, 0)
        pr = (C.c_double * (3 * NCFG))()
        lib.apse_profile_read(model._ctx, C.byref(pr), 1)
        prof = np.array(list(pr)).reshape(NCFG, 3)
        dom = int(np.argmax(prof[:, 0]))
        ms, fl, nl = prof[dom]
        out["dominant_kernel"] = {"kernel": CFG_NAMES[dom], "tflops": round(fl / max(ms, 1e-9) / 1e9, 2),
                                  "frac_of_dense_peak_2500": round(fl / max(ms, 1e-9) / 1e9 / PEAK_BF16_MFMA_TFLOPS, 4),
                                  "avg_launch_ms": round(ms / max(nl, 1), 5), "launches_per_step": int(nl / probe_steps),
                                  "conv_ms_per_step": round(float(prof[:, 0].sum()) / probe_steps, 3),
                                  "measured": "separate pass of %d instrumented steps after the timed steps" % probe_steps}
    if in_flight > 1:
        # the same batches with `in_flight` of them in flight on separate streams / contexts (weights replicated): what a STREAM
        # configuration (configs[4]) sustains when the host keeps the GPU queue full; per-frame results are those of the single-stream
        # run (a frame's bits do not depend on what runs beside it: tests/test_gpu_fullsize.py), latency is in_flight x a step
        from apse_uav_amd.networks.track_rcnn import TrackRCNN
        models = [model]
        for _ in range(in_flight - 1):
            m2 = TrackRCNN(cfg)
            m2.load_state_dict(sd)
            m2.attach_association_head(tr.association_head)
            m2.set_camera(model._camera)
            models.append(m2)
        streams = [torch.cuda.Stream() for _ in range(in_flight)]

        def submit(i):
            k = i % in_flight
            lo = (i * batch) % nres
            bt = frames[lo:lo + batch] if lo + batch <= nres else frames[[(lo + j) % nres for j in range(batch)]]
            with torch.cuda.stream(streams[k]):
                models[k].preprocess_frames(bt)
                models[k].run(batch)

        def collect(i):
            k = i % in_flight
            with torch.cuda.stream(streams[k]):
                res = models[k].read(batch)
            for j in range(batch):
                replay.step(res.record(j), i * batch + j)
        torch.cuda.synchronize()
        for k in range(in_flight):                      # every slot builds its context before the timed part
            submit(k)
            collect(k)
        torch.cuda.synchronize()
        lat2, pend = [], []
        t0 = time.perf_counter()
        for i in range(steps):
            pend.append((i, time.perf_counter()))
            submit(i)
            if len(pend) == in_flight:
                j, ts = pend.pop(0)
                collect(j)
                lat2.append(time.perf_counter() - ts)
        while pend:
            j, ts = pend.pop(0)
            collect(j)
            lat2.append(time.perf_counter() - ts)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        fps2 = steps * batch / dt2
        out["batches_in_flight_%d" % in_flight] = {
            "value": round(fps2, 3), "unit": "frames/s", "steps": steps, "p50_ms_per_batch": round(1e3 * float(np.median(lat2)), 3),
            "whole_path_tflops": round(flops_frame * fps2 / 1e12, 2),
            "mfma_frac_of_dense_peak_2500": round(flops_frame * fps2 / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
            "hbm_frac_of_achievable_6.29": round(out["hbm_algorithmic_bytes_per_frame"] * fps2 / 1e12 / HBM_ACHIEVABLE_TBS, 4) if "hbm_algorithmic_bytes_per_frame" in out else None,
            "what": "%d batches of %d frames in flight on separate HIP streams / contexts, fill and drain inside the timed part" % (in_flight, batch)}
        del models, streams
    del tr, model
    torch.cuda.empty_cache()
    return out


def rehearse_spawn(args, rank, world, backend):
    """The multi-rank skeleton of main() without GPU work: rendezvous on MASTER_ADDR/PORT, barrier, `steps` synthetic
    per-frame records per rank, the ONE gather + rank-0 replay, max-over-ranks timing, one JSON line from rank 0."""
    import torch.distributed as dist
    from apse_uav_amd.sharding import gather_records
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1 and backend == "nccl":
        sys.exit("bench.py --rehearse-spawn runs without a GPU: set APSE_DIST_BACKEND=gloo")
    if os.environ.get("APSE_REHEARSE_FAIL_RANK") == str(rank):
        sys.exit(3)                              # tests: a rank that dies must end the whole job with a non-zero code
    if world > 1:
        dist.init_process_group(backend)
        dist.barrier()
    t0 = time.perf_counter()
    rng = np.random.RandomState(rank)
    recs = []
    for i in range(args.steps):
        n = 3
        e = rng.randn(n, 128).astype(np.float32)
        recs.append(dict(boxes=rng.rand(n, 4).astype(np.float32), scores=rng.rand(n).astype(np.float32),
                         classes=np.zeros(n, np.int64), centroids=rng.randint(1, 2000, (n, 2)).astype(np.int32),
                         mass=np.full(n, 10, np.int32), rects=np.zeros((n, 4), np.int32),
                         closest=rng.randint(1, 2000, (n, n, 2)).astype(np.int32),
                         embeddings=e / np.linalg.norm(e, axis=1, keepdims=True)))
    frames_seen = len(recs)
    if world > 1:
        got = gather_records(recs, rank, world, torch.device("cpu"))
        frames_seen = len(got) if rank == 0 else 0
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        print(json.dumps({"metric": "4K UAV frames/sec (whole node)", "value": None, "unit": "frames/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "rehearsal": True, "records_gathered": frames_seen,
                          "ms_per_step": round(1000.0 * elapsed / max(args.steps, 1), 3)}))
    if world > 1:
        dist.destroy_process_group()


def throughput_mode(cfg, sd, tracker, model, frames, nres, B, depth, replay, steps=96, warmup=4):
    """Same frames, same per-frame results, `depth` frames in flight on separate streams / contexts
    (apse_uav_amd.engines.pipelined_tracker.PipelinedRcnnTracker: detector per frame on its own stream, association
    on the host in frame order): the small-grid layers of one frame (res4/res5 at batch 1 fill ~1 block per CU)
    overlap with other frames'.  Informational: the headline value is the single-stream run above.
    The timed part starts and ends with an EMPTY pipeline (fill and drain included); 96 frames keep that share small."""
    warmup = max(warmup, depth + 1)          # every slot's context is built (first forward) before the timed part
    from apse_uav_amd.engines.pipelined_tracker import PipelinedRcnnTracker
    if B != 1:
        return None
    H, W = frames.shape[1:3]
    drv = PipelinedRcnnTracker(cfg, (H, W), tracker.association_head.state_dict(), depth=depth, detector_state=sd)
    lat = []
    t0 = None
    stamps = {}
    done = 0
    for i in range(warmup + steps):
        if i == warmup:
            while drv._inflight:
                drv.collect()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if len(drv._inflight) == depth:
            j, _ = drv.collect()
            if j in stamps:
                lat.append(time.perf_counter() - stamps.pop(j))
                done += 1
        stamps[drv._submitted] = time.perf_counter() if i >= warmup else None
        if stamps[drv._submitted] is None:
            stamps.pop(drv._submitted)
        drv.submit(frames[i % nres])
    while drv._inflight:
        j, _ = drv.collect()
        if j in stamps:
            lat.append(time.perf_counter() - stamps.pop(j))
            done += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"frames_in_flight": depth, "value": round(done / dt, 3), "unit": "frames/s",
            "p50_ms_per_frame": round(1000.0 * float(np.median(lat)), 3), "steps": done,
            "engine": "PipelinedRcnnTracker"}


def entrypoint_mode(tracker, host_frames, steps, warmup=3):
    """The entry north_star names: ``RcnnTracker.next_frame(frame: np.ndarray HxWx3 u8 BGR)`` from HOST memory, engine
    defaults (mask windows copied out, scipy Hungarian, track store): per frame a 24.9 MB staging copy + H2D over PCIe,
    the GPU path, one apse_copy_mask_window per detection.  Two figures: the plain reference loop
    (visualize_uav.py:186-221), and the same loop passing the next frame as ``upcoming`` so its upload overlaps
    (TrackPredictor.prefetch).  Informational: the headline `value` is measured on frames resident in HBM."""
    from apse_uav_amd.utils import csv_log  # noqa: F401  (log_line)
    n = len(host_frames)
    out = {}
    for name, ahead in (("next_frame", False), ("next_frame_upcoming", True)):
        tracker.reset_tracker()
        lat = []
        t0 = 0.0
        for i in range(warmup + steps):
            if i == warmup:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            ts = time.perf_counter()
            objs = tracker.next_frame(host_frames[i % n], upcoming=host_frames[(i + 1) % n] if ahead else None)
            tracker.log_line(objs, 1, i)
            if i >= warmup:
                lat.append(time.perf_counter() - ts)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[name] = {"value": round(steps / dt, 3), "unit": "frames/s", "p50_ms_per_frame": round(1000.0 * float(np.median(lat)), 3),
                     "steps": steps}
    out["what"] = ("RcnnTracker.next_frame(np.ndarray) from pageable host memory incl. staging copy, H2D, mask-window copies, "
                   "scipy association and the CSV line")
    return out


def cpu_baseline(sd, asd, host_frames, blocks, H, W, nframes, max_threads):
    """The CPU oracle (a port: the reference's own CPU path cannot run, detectron2 is absent) on the same
    frames and weights: PIL resize + detector + roi_pool/embedding + tracker association + CSV line.  Timed with all
    host cores torch will use (capped by --cpu-threads) and with 8 threads (comparable with an 8-core host)."""
    from PIL import Image
    from oracle import tracker as otr
    from oracle.detector import DetectorOracle, resize_shape
    from apse_uav_amd.utils.hostinfo import usable_cpus
    avail = usable_cpus()                                   # cgroup quota / affinity, not os.cpu_count()
    oracle = DetectorOracle(sd, dict(depth_blocks=blocks))
    ih, iw = resize_shape(H, W)

    def timed(ncores, nfr):
        torch.set_num_threads(ncores)
        otk = otr.TrackerOracle()

        def one(fr, t):
            img = np.asarray(Image.fromarray(fr).resize((iw, ih), Image.BILINEAR))
            post = oracle.inference(torch.as_tensor(img.astype("float32").transpose(2, 0, 1)), H, W)
            rois = otr.features_rois(post["features"]["p2"], post["boxes"], W)
            emb = otr.association_head(rois, asd["fc.weight"], asd["fc.bias"])
            rec = otk.next_frame(dict(boxes=post["boxes"], scores=post["scores"], classes=post["classes"],
                                      masks=list(zip(post["mask_windows"], post["mask_rects"])), emb=emb))
            otr.log_oneline(rec, 1, t)
        with torch.no_grad():
            one(host_frames[0], 0)
            t0 = time.perf_counter()
            for t in range(nfr):
                one(host_frames[(t + 1) % len(host_frames)], t + 1)
            return nfr / (time.perf_counter() - t0)
    ncores = min(avail, max_threads)
    fps = timed(ncores, nframes)
    res = {"value": round(fps, 4), "unit": "frames/s", "cores": ncores, "kind": "port",
           "sample": "%d frames of the same synthetic 3840x2160 sequence after 1 warm-up, PyTorch-CPU f32 oracle "
                     "(torch threads = %d)" % (nframes, ncores)}
    if ncores > 8:
        n8 = max(2, nframes - 1)
        res["threads_8"] = {"value": round(timed(8, n8), 4), "unit": "frames/s", "cores": 8,
                            "sample": "%d frames after 1 warm-up, torch threads = 8" % n8}
    torch.set_num_threads(avail)
    res["host_cpus_usable"] = avail
    return res


if __name__ == "__main__":
    main()
