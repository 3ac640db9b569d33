"""PipelinedRcnnTracker -- a sequence driver on top of RcnnTracker for recorded video (build extension).

The reference processes a recorded sequence one frame at a time
(/root/reference/dcnn/scripts/tests/visualize_uav.py:186-221: read, ``tracker.next_frame``, log).  Per frame the
detector is stateless; only the association (ids) is sequential.  This driver keeps ``depth`` frames in flight,
each on its own HIP stream and detector context (weights replicated, ~250 MB each), and runs the association on
the host strictly in frame order from each frame's results block -- the same split a frame-sharded multi-GPU run
uses, inside one GPU.  Small-grid layers of one frame (res4 / res5 at batch 1 have one tile per CU)
then overlap with other frames' work: ~175 vs ~155 frames/s at 3840x2160 on one MI355X, at ``depth`` x the
per-frame latency (measured: depth 2 / 3 / 4 = 176 / 180 / 183 frames/s; depth 6 collapses to 45 -- more streams than
the device serves concurrently only add queue switching, so keep depth <= 4).  Results are identical to
``RcnnTracker.next_frame`` frame by frame (tests/test_gpu_detector.py; a 48-frame 4K sequence gives the same CSV).

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


class PipelinedRcnnTracker:
    def __init__(self, config, image_size, weights, depth=3, want_masks=False, detector_state=None, **tracker_kwargs):
        assert 1 <= depth <= 4, "depth 2..4 (more frames in flight than the device runs concurrently is slower, see the module text)"
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
            self.models.append(m)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(depth)]
        self._uploader = FrameUploader(self.device, self.tracker.predictor.input_format, nslots=depth + 1)
        self._inflight = collections.deque()
        self._submitted = 0

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
