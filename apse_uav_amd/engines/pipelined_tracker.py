"""PipelinedRcnnTracker -- a sequence driver on top of RcnnTracker for recorded video (build extension).

The reference processes a recorded sequence one frame at a time
(/root/reference/dcnn/scripts/tests/visualize_uav.py:186-221: read, ``tracker.next_frame``, log).  Per frame the
detector is stateless; only the association (ids) is sequential.  This driver keeps ``depth`` frames in flight,
each on its own HIP stream and detector context (weights replicated, ~250 MB each), and runs the association on
the host strictly in frame order from each frame's results block -- the same split a frame-sharded multi-GPU run
uses, inside one GPU.  Small-grid layers of one frame (res4 / res5 at batch 1 have one tile per CU)
then overlap with other frames' work: at 3840x2160 on one MI355X (round 2, tools/pipeline_probe.py) depth 2 / 3 / 4 / 6 / 8 =
179 / 177 / 195 / 188 / 203 frames/s against ~155 for the plain loop, at ``depth`` x the per-frame latency.  (Round 1 reported
a collapse to 45 frames/s at depth 6 and fenced the depth at 4: that was the measurement, not the device -- a slot's context
is built on its first forward (~0.2 s), and with fewer warm-up frames than slots those builds landed inside the timed
region.  Every slot now gets a priming forward first; hardware-queue count (GPU_MAX_HW_QUEUES 4 vs 8) makes no difference.)
Results are identical to ``RcnnTracker.next_frame`` frame by frame (tests/test_gpu_detector.py, tests/test_gpu_fullsize.py).

    drv = PipelinedRcnnTracker(config, image_size, weights, depth=3, detector_state=sd)
    for frame_idx, objects in drv.run(frames):              # frames: iterable of HxWx3 uint8 BGR arrays
        line, _ = drv.tracker.log_line(objects, host_id, frame_idx)
"""
import collections

import numpy as np
import torch

from ..networks.track_rcnn import TrackRCNN
from .rcnn_tracker import RcnnTracker, instances_from_record
from .track_predictor import FrameUploader


MAX_DEPTH = 8           # ~250 MB of weights + 1.6 GB of activations per slot; beyond ~4 the gain is within the noise


class PipelinedRcnnTracker:
    def __init__(self, config, image_size, weights, depth=3, want_masks=False, detector_state=None, **tracker_kwargs):
        assert 1 <= depth <= MAX_DEPTH, "depth 1..%d" % MAX_DEPTH
        self.tracker = RcnnTracker(config, image_size, weights, detector_state=detector_state, **tracker_kwargs)
        self.depth = depth
        self.want_masks = want_masks
        self.device = self.tracker.device
        first = self.tracker.predictor.model
        self.models = [first]
        for _ in range(depth - 1):
            m = TrackRCNN(self.tracker.predictor.cfg)
            m.to(self.device)
            m.load_state_dict(first._state)
            m.attach_association_head(self.tracker.association_head)
            m.set_camera(first._camera)
            self.models.append(m)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(depth)]
        self._uploader = FrameUploader(self.device, self.tracker.predictor.input_format, nslots=depth + 1)
        self._inflight = collections.deque()
        self._submitted = 0

    def set_camera(self, cam_params, gamma=2.0, fused=None):
        """TrackPredictor.set_camera for every slot (undistort + gamma in front of the resize)."""
        self.tracker.predictor.set_camera(cam_params, gamma, fused)
        for m in self.models[1:]:
            m.set_camera(self.models[0]._camera)

    # ------------------------------------------------------------------ one frame in, zero or one out
    def submit(self, frame):
        """Enqueues a frame (HxWx3 uint8 ndarray in cfg.INPUT.FORMAT order, or a uint8 BGR CUDA tensor [H, W, 3]) on
        the next slot.  Call ``collect`` first when ``depth`` frames are already in flight."""
        assert len(self._inflight) < self.depth, "collect() before submitting more than depth frames"
        k = self._submitted % self.depth
        st = self.streams[k]
        slot = None
        if torch.is_tensor(frame):
            dev = frame.reshape((1,) + tuple(frame.shape[-3:]))
            st.wait_stream(torch.cuda.current_stream(self.device))     # whatever produced the tensor
            dev.record_stream(st)                                      # the allocator must not recycle it under us
        else:
            slot = self._uploader.begin([frame], st)                   # same staging / channel-order logic as TrackPredictor
            dev = slot.dev
        with torch.cuda.stream(st):
            if self.tracker.predictor.frame_preprocessor is not None:
                dev = self.tracker.predictor.frame_preprocessor(dev)
            self.models[k].preprocess_frames(dev)
            if slot is not None:
                FrameUploader.release(slot, st)
            self.models[k].run(1)
        self._inflight.append((self._submitted, k))
        self._submitted += 1

    def collect(self):
        """Waits for the oldest frame in flight and runs its association.  Returns (frame_index, ObjectInstances)."""
        idx, k = self._inflight.popleft()
        m = self.models[k]
        with torch.cuda.stream(self.streams[k]):
            res = m.read(1)                                     # D2H of the results block + stream sync
            if self.want_masks:
                det = m.instances_from(res, 0, want_masks=True)  # mask windows copied before the slot is reused
                torch.cuda.current_stream().synchronize()
            else:
                det = instances_from_record(res.record(0), self.tracker.image_size, self.device)
        t = self.tracker
        t.frame_count += 1
        return idx, t._finish_frame(det, None, host_replay=True)

    # ------------------------------------------------------------------ whole sequence
    def run(self, frames):
        for f in frames:
            if len(self._inflight) == self.depth:
                yield self.collect()
            self.submit(f)
        while self._inflight:
            yield self.collect()
