"""TrackPredictor -- counterpart of /root/reference/dcnn/engines/track_predictor.py:11-52.

Same constructor (``cfg``) and call contract: ``predictor(original_image: HxWx3 uint8 BGR)`` ->
``(predictions, backbone_features)`` with ``predictions["instances"]`` and the FPN dict.  The
reference resizes on the CPU with PIL and uploads a 12 MB f32 image; here the u8 frame is uploaded
once (24.9 MB at 4K) and the PIL-exact resize, normalisation and padding run as HIP kernels.
Weights: ``cfg.MODEL.WEIGHTS`` (a ``.pth`` with key "model" or a bare state_dict) or
``load_state_dict``.

Ingest (the reference loop is dcnn/scripts/tests/visualize_uav.py:172-190): frames go
through ``FrameUploader`` -- two pinned staging buffers, two device buffers and a copy stream.  ``prefetch(frame)``
starts the host copy + H2D of the NEXT frame while the current one computes; ``predictor(frame)`` then finds it
resident.  Without a prefetch the same buffers are used on the calling stream.
"""
import os

import numpy as np
import torch

from ..networks.track_rcnn import TrackRCNN
from ..weights import load_detector_file


_COPY_THREADS = max(1, int(os.environ.get("APSE_STAGE_THREADS", "8")))       # host threads of the staging copy (tools/entry_probe.py sweeps it)


def _host_copy(dst, src):
    """dst[...] = src for HxWx3 uint8 arrays (25 MB at 4K).  Contiguous rows go through ``apse_host_copy`` (csrc/host_stage.hip: a
    persistent pool of plain threads inside the library; ctypes releases the GIL for the call); a strided source (the RGB flip)
    is left to numpy.  Deliberately NOT a torch op: torch's intra-op pool (OpenMP) spins after a parallel region, and under a
    container CPU quota those spinning workers get the whole process throttled for tens of milliseconds every few frames."""
    if src.flags["C_CONTIGUOUS"] and dst.flags["C_CONTIGUOUS"] and src.dtype == dst.dtype and src.shape == dst.shape:
        from .. import _lib
        _lib.check(_lib.load().apse_host_copy(dst.ctypes.data, src.ctypes.data, dst.nbytes, _COPY_THREADS), None, "apse_host_copy")
        return
    np.copyto(dst, src)


class _Slot:
    __slots__ = ("pinned", "pinned_np", "dev", "h2d_done", "consumed", "consumed_pending", "key")

    def __init__(self, shape, device):
        self.pinned = torch.empty(shape, dtype=torch.uint8).pin_memory()
        self.pinned_np = self.pinned.numpy()
        self.dev = torch.empty(shape, dtype=torch.uint8, device=device)
        self.h2d_done = torch.cuda.Event()
        self.consumed = torch.cuda.Event()
        self.consumed_pending = False        # True: handed out, the consumer's event has not been recorded yet
        self.key = None


class FrameUploader:
    """Host frames -> device, double-buffered.  ``begin(frames, stream)`` stages ``frames`` (list of HxWx3 uint8
    arrays) into the next slot's pinned buffer and enqueues the H2D on ``stream``; the slot's device buffer is
    overwritten only after its previous consumer has run (``consumed`` event, recorded by ``release``)."""

    BANDS = max(1, int(os.environ.get("APSE_STAGE_BANDS", "4")))

    def __init__(self, device, input_format="BGR", nslots=2):
        self.device = torch.device(device)
        self.input_format = input_format
        self.nslots = max(2, int(nslots))
        self._slots = [None] * self.nslots
        self._k = -1

    def begin(self, frames, stream):
        B = len(frames)
        H, W = frames[0].shape[:2]
        shape = (B, H, W, 3)
        self._k = (self._k + 1) % self.nslots
        sl = self._slots[self._k]
        if sl is None or tuple(sl.pinned.shape) != shape:
            sl = self._slots[self._k] = _Slot(shape, self.device)
        else:
            sl.h2d_done.synchronize()                 # the pinned buffer's previous H2D has left the host
        for f in frames:
            if f.shape[:2] != (H, W):
                raise ValueError("frames of one batch must have the same size")
        if sl.consumed_pending:                       # a caller that never released the slot: be conservative
            sl.consumed.record(torch.cuda.current_stream(self.device))
            sl.consumed_pending = False
        with torch.cuda.stream(stream):
            stream.wait_event(sl.consumed)            # device buffer: its last reader (the resize kernels) is done
            # staged and sent in row bands: while band k crosses PCIe, the host threads copy band k + 1 into the pinned buffer
            # (a 4K frame: ~1 ms of host copy + ~0.5 ms of DMA, overlapped instead of back to back)
            nb = self.BANDS if H >= 8 * self.BANDS else 1
            step = (H + nb - 1) // nb
            for i, f in enumerate(frames):
                if self.input_format == "RGB":        # track_predictor.py:43-45: the model wants BGR
                    f = f[:, :, ::-1]
                for lo in range(0, H, step):
                    hi = min(lo + step, H)
                    _host_copy(sl.pinned_np[i, lo:hi], f[lo:hi])      # strided source (the RGB flip) is handled by numpy
                    sl.dev[i, lo:hi].copy_(sl.pinned[i, lo:hi], non_blocking=True)
            sl.h2d_done.record(stream)
        sl.key = list(frames)                        # references, so identity stays meaningful until the slot is reused
        sl.consumed_pending = True
        return sl

    @staticmethod
    def release(sl, stream=None):
        """Call after the last kernel reading ``sl.dev`` has been enqueued on ``stream`` (default: current)."""
        sl.consumed.record(stream if stream is not None else torch.cuda.current_stream(sl.dev.device))
        sl.consumed_pending = False


class TrackPredictor:
    model_class = TrackRCNN

    def __init__(self, cfg, state_dict=None):
        self.cfg = cfg.clone()
        self.model = self.model_class(self.cfg)
        self.model.to(torch.device(cfg.MODEL.DEVICE))
        self.model.eval()
        if state_dict is not None:
            self.model.load_state_dict(state_dict)
        elif cfg.MODEL.WEIGHTS:
            self.model.load_state_dict(load_detector_file(cfg.MODEL.WEIGHTS))
        self.input_format = cfg.INPUT.FORMAT
        assert self.input_format in ["RGB", "BGR"], self.input_format
        self.frame_preprocessor = None        # optional FramePreprocessor (cfg.APSE.FUSED_PREPROC / set_camera)
        self._uploader = None
        self._copy_stream = None
        self._prefetched = None
        self._last_slot = None

    def set_camera(self, cam_params, gamma=2.0, fused=None):
        """Enables undistort + Lab-gamma in front of the resize (preprocess_img, visualize_uav.py:56-71).  ``fused`` (default
        cfg.APSE.FUSED_PREPROC): computed inside the resize kernel's row staging (no intermediate 4K frame in HBM); otherwise as
        a separate kernel on the uploaded frame.  Both give the same bytes (tests/test_gpu_detector.py)."""
        from ..utils.preprocess import FramePreprocessor
        pp = FramePreprocessor(cam_params, gamma)
        if fused is None:
            fused = bool(self.cfg.APSE.get("FUSED_PREPROC", True))
        if fused:
            self.frame_preprocessor = None
            self.model.set_camera(pp)
        else:
            self.model.set_camera(None)
            self.frame_preprocessor = pp

    # ------------------------------------------------------------------ ingest
    def _up(self):
        if self._uploader is None:
            if self.model.device.type != "cuda" or not torch.cuda.is_available():
                from .._lib import ApseError
                raise ApseError("the apse_uav hot path needs a ROCm GPU (cfg.MODEL.DEVICE=%s): no CPU fallback" % self.model.device)
            self._uploader = FrameUploader(self.model.device, self.input_format)
            self._copy_stream = torch.cuda.Stream(device=self.model.device)
        return self._uploader

    def prefetch(self, frames):
        """Starts the upload of the frame(s) the NEXT call will be given (an HxWx3 array or a list of them) on the
        copy stream, so the 24.9 MB H2D overlaps the GPU work already enqueued.  The next call must pass the SAME array
        object(s), unmodified since this call (the match is by identity: the bytes were staged here); anything else is uploaded
        afresh."""
        if frames is None:
            return
        frames = list(frames) if isinstance(frames, (list, tuple)) else [frames]
        self._prefetched = self._up().begin(frames, self._copy_stream)

    def _upload(self, frames):
        """list of HxWx3 uint8 arrays -> CUDA tensor [B, H, W, 3] (BGR), resident or in flight on the current stream."""
        up = self._up()
        cur = torch.cuda.current_stream(self.model.device)
        sl, self._prefetched = self._prefetched, None
        if sl is not None and len(sl.key) == len(frames) and all(a is b for a, b in zip(sl.key, frames)):
            cur.wait_event(sl.h2d_done)
        else:
            if sl is not None:
                FrameUploader.release(sl, self._copy_stream)          # a prefetch that was never used
            sl = up.begin(frames, cur)
        self._last_slot = sl
        dev = sl.dev
        if self.frame_preprocessor is not None:
            dev = self.frame_preprocessor(dev)
            FrameUploader.release(sl)
        return dev

    def _frames_consumed(self):
        """The resize kernels reading the uploaded frame are enqueued: its device buffer may be refilled after them."""
        if self._last_slot is not None and self._last_slot.consumed_pending:
            FrameUploader.release(self._last_slot)

    # ------------------------------------------------------------------ inference
    def _prestage(self):
        """The announced next frame(s) are on their way to the device: enqueue their resize + normalise BEHIND the current forward
        (and behind its results copy), so that the next call starts with the network itself.  The input tensor is free by then
        (the stem read it at the start of this forward); nothing else of this forward is touched."""
        sl = self._prefetched
        if sl is None or self.frame_preprocessor is not None:
            return
        key = self.model._ctx_key
        if key is None or tuple(sl.dev.shape[1:3]) != tuple(key[0]) or sl.dev.shape[0] > int(self.cfg.APSE.MAX_BATCH):
            # an announced frame of ANOTHER size: staging it would rebuild the context while this frame's results copy is still
            # pending (read_begin .. read_end).  Leave the slot to the next call's normal upload path (it still finds the frame
            # resident); no pre-stage, hence no run-ahead.
            return
        self._prefetched = None
        torch.cuda.current_stream(self.model.device).wait_event(sl.h2d_done)
        self.model.preprocess_frames(sl.dev, tag=sl.key)
        FrameUploader.release(sl)

    def _predict(self, frames, given=None, want_masks=True, upcoming=None, rpn_levels=31, run_ahead=False):
        """``run_ahead`` (callers that do not read the returned feature dict after this call returns -- RcnnTracker.next_frame):
        when the announced next frame has been staged, its NETWORK is enqueued too, right behind this frame's results copy, so the
        card goes from one frame to the next without waiting for the host and the host-side association of this frame overlaps the
        GPU work of the next.  The next call then only reads.  Stream order keeps this frame's results copy in front of everything
        the next forward overwrites; the mask windows copied out below come from this frame's set of bit planes (two sets alternate)."""
        model = self.model
        rtag = model._running_tag
        if (rtag is not None and given is None and rtag[1] == rpn_levels and len(rtag[0]) == len(frames)
                and all(a is b for a, b in zip(rtag[0], frames))):
            B = len(frames)                   # this frame's forward was enqueued at the end of the previous call
            model._running_tag = None
        else:
            tag = model._input_tag
            if tag is not None and len(tag) == len(frames) and all(a is b for a, b in zip(tag, frames)):
                B = len(frames)               # pre-staged behind the previous forward: the input already holds these frames
                model._input_tag = None
            else:
                dev = self._upload(frames)
                B = model.preprocess_frames(dev)
                self._frames_consumed()
            model.run(B, given, rpn_levels)
        self.prefetch(upcoming)               # host copy + H2D of the next frame while this one computes
        model.read_begin(B)
        self._prestage()                      # ... and its resize, behind this frame's results copy
        ahead = None
        if run_ahead and given is None and model._input_tag is not None:
            # ... and its network: nothing in it depends on this frame; the stream keeps this frame's results copy in front of what
            # the forward overwrites and the library alternates the mask bit planes, so the card never waits for the host to wake up
            nxt = model._input_tag
            model._input_tag = None
            model.run(len(nxt), None, rpn_levels)
            ahead = (nxt, rpn_levels)
        res = model.read_end(B)
        from ..networks.track_rcnn import LazyFeatures
        insts = [model.instances_from(res, b, want_masks) for b in range(B)]       # mask windows (of THIS frame) are copied out here
        model._running_tag = ahead
        return insts, LazyFeatures(model, B)

    def __call__(self, original_image, upcoming=None, run_ahead=False):
        with torch.no_grad():
            insts, feats = self._predict([original_image], upcoming=upcoming, run_ahead=run_ahead)
            return {"instances": insts[0]}, feats

    def predict_batch(self, frames, given=None, want_masks=True, upcoming=None):
        """Build extension (reference is batch 1): several frames in one forward."""
        with torch.no_grad():
            insts, feats = self._predict(list(frames), given=given, want_masks=want_masks, upcoming=upcoming)
            return [{"instances": i} for i in insts], feats
