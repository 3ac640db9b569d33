"""TrackRCNN -- counterpart of /root/reference/dcnn/networks/track_rcnn.py:5-58.

The reference subclasses detectron2's GeneralizedRCNN and returns ``(postprocessed results,
FPN feature dict)`` from ``inference``.  Here the whole per-frame computation is one enqueue
sequence of hand-written HIP kernels inside ``libapse_hip.so`` (include/apse_hip.h): this class
owns the library context, feeds it weights (detectron2 state_dict key names) and turns the
results block into the reference's return types.  There is no CPU fallback.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..structures.instances import Boxes, Instances
from ..structures.window_mask import MaskList, WindowMask
from ..utils import resample
from ..weights import blocks_from_state


class _Shape:
    def __init__(self, channels, stride):
        self.channels, self.stride = channels, stride


class _Backbone:
    """``model.backbone.output_shape()`` as read by rcnn_tracker.py:53."""
    size_divisibility = 32

    def output_shape(self):
        return {"p%d" % l: _Shape(256, 2 ** l) for l in range(2, 7)}


class LazyFeatures(dict):
    """{"p2".."p6": NCHW f32 CUDA tensor}, exported from the context on first access."""

    def __init__(self, model, batch):
        super().__init__()
        self._model, self._batch = model, batch

    def __missing__(self, key):
        t = self._model.export_feature(key, self._batch)
        self[key] = t
        return t

    def keys(self):
        return ["p2", "p3", "p4", "p5", "p6"]


class FrameResults:
    """Host view of one forward's results block (apse_results_layout)."""

    def __init__(self, raw, lay, batch):
        self.raw, self.lay, self.batch = raw, lay, batch
        n, kd, e = lay.n_max, lay.dets_per_image, lay.embed_dim

        def arr(off, dtype, shape):
            cnt = int(np.prod(shape))
            return np.frombuffer(raw, dtype=dtype, count=cnt, offset=off).reshape(shape)

        self.total = int(arr(lay.total, np.int32, (1,))[0])
        self.offset = arr(lay.offset, np.int32, (lay.max_batch + 1,))
        self.prop_count = arr(lay.prop_count, np.int32, (lay.max_batch,))
        self.img = arr(lay.img, np.int32, (n,))
        self.cls = arr(lay.cls, np.int32, (n,))
        self.roi = arr(lay.roi, np.int32, (n,))
        self.score = arr(lay.score, np.float32, (n,))
        self.box_resized = arr(lay.box_resized, np.float32, (n, 4))
        self.box = arr(lay.box, np.float32, (n, 4))
        self.valid = arr(lay.valid, np.int32, (n,))
        self.rect = arr(lay.rect, np.int32, (n, 4))
        self.mass = arr(lay.mass, np.int32, (n,))
        self.centroid = arr(lay.centroid, np.int32, (n, 2))
        self.closest = arr(lay.closest, np.int32, (n, kd, 2))
        self.embedding = arr(lay.embedding, np.float32, (n, e))

    def image_slice(self, b):
        return int(self.offset[b]), int(self.offset[b + 1])

    def record(self, b):
        """Self-contained per-frame record (what a rank ships to rank 0 in sharded mode):
        only detections whose scaled box is non-empty (detector_postprocess drops the others)."""
        lo, hi = self.image_slice(b)
        keep = np.nonzero(self.valid[lo:hi])[0]
        idx = lo + keep
        return dict(
            boxes=self.box[idx].copy(), scores=self.score[idx].copy(), classes=self.cls[idx].astype(np.int64),
            centroids=self.centroid[idx].copy(), mass=self.mass[idx].copy(), rects=self.rect[idx].copy(),
            closest=self.closest[idx][:, keep, :].copy(), embeddings=self.embedding[idx].copy(), packed_index=idx.copy())


class TrackRCNN:
    def __init__(self, cfg):
        self.cfg = cfg
        self.device = torch.device(cfg.MODEL.DEVICE)
        self.backbone = _Backbone()
        self.training = False
        self._state = None
        self._assoc = None
        self._ctx = None
        self._ctx_key = None
        self._lay = None
        self._host = None
        self._camera = None                   # FramePreprocessor: undistort + gamma fused into preprocess_frames
        self._input_tag = None                # frames (objects) the network input was pre-staged with, or None
        self._running_tag = None              # (frames, rpn_levels) whose forward is already enqueued and not yet read, or None
        self.last_results = None

    # ---- nn.Module-like surface the reference touches
    def to(self, device):
        self.device = torch.device(device)
        return self

    def eval(self):
        self.training = False
        return self

    def load_state_dict(self, sd):
        self._state = {k: v.detach().to(torch.float32).cpu().contiguous() for k, v in sd.items()}
        self._drop_ctx()

    def attach_association_head(self, head):
        """Fuses the tracker's AssociationHead (rcnn_tracker.py:55-57) into the per-frame launch sequence."""
        self._assoc = head
        self._drop_ctx()

    def set_camera(self, preproc):
        """``preproc``: utils.preprocess.FramePreprocessor (camera matrix, distortion, gamma LUT) or None.  The context then runs
        undistort + gamma inside ``apse_preprocess_frames`` (include/apse_hip.h: apse_set_camera)."""
        self._camera = preproc
        if self._ctx is not None:
            self._push_camera()

    def _push_camera(self):
        lib = _lib.load()
        pc = self._camera
        if pc is None:
            _lib.check(lib.apse_set_camera(self._ctx, None, None, 0, None, 0, 0), self._ctx, "apse_set_camera")
            return
        _lib.check(lib.apse_set_camera(self._ctx, pc.mtx.ctypes.data_as(C.POINTER(C.c_double)), pc.dist.ctypes.data_as(C.POINTER(C.c_double)),
                                       int(pc.dist.size), _lib.ptr(pc.lut_host), int(pc.undistort), int(pc.gamma_correct)),
                   self._ctx, "apse_set_camera")

    def _drop_ctx(self):
        self._input_tag = None
        self._running_tag = None
        if self._ctx is not None:
            _lib.load().apse_destroy(self._ctx)
            self._ctx = None
            self._ctx_key = None

    def __del__(self):
        try:
            self._drop_ctx()
        except Exception:
            pass

    # ---- context
    def _ensure_ctx(self, frame_hw, image_hw):
        key = (tuple(frame_hw), tuple(image_hw))
        if self._ctx is not None and self._ctx_key == key:
            return
        self._drop_ctx()
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise _lib.ApseError("the apse_uav hot path needs a ROCm GPU (cfg.MODEL.DEVICE=%s): no CPU fallback" % self.device)
        if self._state is None:
            raise _lib.ApseError("no detector weights loaded")
        lib = _lib.load()
        cfg = self.cfg
        c = _lib.Config()
        c.struct_size = C.sizeof(_lib.Config)
        c.device = self.device.index if self.device.index is not None else torch.cuda.current_device()
        c.max_batch = int(cfg.APSE.MAX_BATCH)
        c.frame_h, c.frame_w = int(frame_hw[0]), int(frame_hw[1])
        c.image_h, c.image_w = int(image_hw[0]), int(image_hw[1])
        blocks = blocks_from_state(self._state)
        for i in range(4):
            c.blocks[i] = blocks[i]
        c.num_classes = int(cfg.MODEL.ROI_HEADS.NUM_CLASSES)
        c.score_thresh = float(cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST)
        c.box_nms = float(cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST)
        c.rpn_nms = float(cfg.MODEL.RPN.NMS_THRESH)
        c.mask_thresh = 0.5
        c.rpn_pre_topk = int(cfg.MODEL.RPN.PRE_NMS_TOPK_TEST)
        c.rpn_post_topk = int(cfg.MODEL.RPN.POST_NMS_TOPK_TEST)
        c.dets_per_image = int(cfg.TEST.DETECTIONS_PER_IMAGE)
        for i in range(3):
            c.pixel_mean[i] = float(cfg.MODEL.PIXEL_MEAN[i])
        roi = self._assoc.roi_size if self._assoc is not None else 10
        c.assoc_roi = roi
        c.embed_dim = self._assoc.embedding_dim if self._assoc is not None else 128
        ph = (c.image_h + 31) // 32 * 32
        pw = (c.image_w + 31) // 32 * 32
        c.assoc_scale = (pw // 4) / float(c.frame_w)          # features.size()[3] / image_size[1]  (rcnn_tracker.py:165)
        c.compute_dtype = {"f32": 0, "bf16": 1, "f16": 2, "fp16": 2}[str(cfg.APSE.DTYPE)]
        c.storage16 = int(bool(cfg.APSE.get("STORAGE16", True))) if c.compute_dtype else 0
        ctx = C.c_void_p()
        _lib.check(lib.apse_create(C.byref(c), C.byref(ctx)), None, "apse_create: " + lib.apse_last_error(None).decode())
        try:
            sd = dict(self._state)
            if self._assoc is not None:
                sd["association.fc.weight"] = self._assoc.fc.weight
                sd["association.fc.bias"] = self._assoc.fc.bias
            else:
                sd["association.fc.weight"] = torch.zeros(c.embed_dim, 256 * roi * roi)
                sd["association.fc.bias"] = torch.zeros(c.embed_dim)
            for name, t in sd.items():
                if name.startswith("proposal_generator.anchor_generator") or name in ("pixel_mean", "pixel_std"):
                    continue
                a = np.ascontiguousarray(t.numpy(), dtype=np.float32)
                shp = (C.c_int64 * a.ndim)(*a.shape)
                _lib.check(lib.apse_set_weight(ctx, name.encode(), _lib.ptr(a), shp, a.ndim), ctx, "apse_set_weight")
            _lib.check(lib.apse_finalize_weights(ctx), ctx, "apse_finalize_weights")
            hb, hc, hk = resample.precompute_coeffs(c.frame_w, c.image_w)
            vb, vc, vk = resample.precompute_coeffs(c.frame_h, c.image_h)
            _lib.check(lib.apse_set_resize_tables(ctx, _lib.ptr(hb), _lib.ptr(hc), hk, _lib.ptr(vb), _lib.ptr(vc), vk), ctx,
                       "apse_set_resize_tables")
            lay = _lib.ResultsLayout()
            _lib.check(lib.apse_results_describe(ctx, C.byref(lay)), ctx, "apse_results_describe")
        except Exception:
            lib.apse_destroy(ctx)
            raise
        self._ctx, self._ctx_key, self._lay = ctx, key, lay
        if self._camera is not None:
            self._push_camera()
        self._host = torch.empty(lay.bytes, dtype=torch.uint8).pin_memory()
        self._cfg_c = c

    # ---- stage calls (each enqueues on the current stream)
    def _call(self, fn, *args):
        lib = _lib.load()
        _lib.check(getattr(lib, fn)(self._ctx, *args), self._ctx, fn)

    def preprocess_frames(self, frames, tag=None):
        """frames: uint8 CUDA tensor [B, H, W, 3] (BGR).  Fused PIL-exact resize + normalise + pad.  ``tag``: what the network
        input now holds, for a caller that stages the NEXT frame's input behind the current forward (TrackPredictor._prestage);
        any other write of the input clears it."""
        B, H, W, _ = frames.shape
        ih, iw = resample.resize_shortest_edge(H, W, self.cfg.INPUT.MIN_SIZE_TEST, self.cfg.INPUT.MAX_SIZE_TEST)
        self._ensure_ctx((H, W), (ih, iw))
        self._input_tag = None
        self._running_tag = None
        self._call("apse_preprocess_frames", _lib.ptr(frames.contiguous()), B, _lib.stream_ptr())
        self._input_tag = tag
        return B

    def preprocess_images(self, images, frame_hw):
        """images: f32 CUDA tensor [B, 3, h, w] already resized (the reference's model input)."""
        B, _, h, w = images.shape
        self._ensure_ctx(frame_hw, (h, w))
        self._input_tag = None
        self._running_tag = None
        self._call("apse_preprocess_images", _lib.ptr(images.contiguous()), B, _lib.stream_ptr())
        return B

    def run(self, batch, given=None, rpn_levels=31):
        self._running_tag = None              # any forward invalidates a speculatively enqueued one (TrackPredictor run-ahead)
        s = _lib.stream_ptr()
        self._call("apse_backbone", batch, s)
        if given is None:
            self._call("apse_rpn_levels", batch, int(rpn_levels), s)
            self._call("apse_box_head", batch, s)
        else:
            boxes, classes, counts = given
            boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 4)
            classes = np.ascontiguousarray(classes, np.int32)
            counts = np.ascontiguousarray(counts, np.int32)
            self._call("apse_set_detections", _lib.ptr(boxes), _lib.ptr(classes), None, _lib.ptr(counts), batch, s)
        self._call("apse_mask_tail", batch, s)
        self._call("apse_embed", batch, s)

    def read(self, batch):
        self.read_begin(batch)
        return self.read_end(batch)

    def read_begin(self, batch):
        """Enqueues the D2H of the results block; ``read_end`` waits for THAT copy only, so kernels enqueued in between (the next
        frame's ``preprocess_frames``) run behind the current forward without delaying its results."""
        self._call("apse_read_results_begin", C.c_void_p(self._host.data_ptr()), self._lay.bytes, _lib.stream_ptr())

    def read_end(self, batch):
        self._call("apse_read_results_end", C.c_void_p(self._host.data_ptr()))
        raw = self._host.numpy().tobytes()
        self.last_results = FrameResults(raw, self._lay, batch)
        return self.last_results

    def export_feature(self, name, batch):
        shp = (C.c_int * 3)()
        self._call("apse_feature_shape", name.encode(), C.byref(shp))
        out = torch.empty((batch, shp[0], shp[1], shp[2]), dtype=torch.float32, device=self.device)
        self._call("apse_export_feature", name.encode(), _lib.ptr(out), batch, _lib.stream_ptr())
        return out

    def debug_tensor(self, name, dtype=torch.float32):
        n = C.c_size_t()
        self._call("apse_debug_tensor", name.encode(), None, 0, C.byref(n), _lib.stream_ptr())
        out = torch.empty(n.value // torch.empty((), dtype=dtype).element_size(), dtype=dtype, device=self.device)
        self._call("apse_debug_tensor", name.encode(), _lib.ptr(out), n.value, C.byref(n), _lib.stream_ptr())
        return out

    def flops(self, batch, proposals, detections):
        return float(_lib.load().apse_flops(self._ctx, batch, proposals, detections))

    # ---- results -> reference types
    def instances_from(self, res, b, want_masks=True):
        frame_hw = self._ctx_key[0]
        rec = res.record(b)
        n = len(rec["scores"])
        inst = Instances(frame_hw)
        inst.pred_boxes = Boxes(torch.from_numpy(rec["boxes"]))
        inst.scores = torch.from_numpy(rec["scores"])
        inst.pred_classes = torch.from_numpy(rec["classes"])
        masks = MaskList()
        lib = _lib.load()
        rects = np.ascontiguousarray(rec["rects"], np.int32).reshape(n, 4)
        # every window of the frame out of the bit planes with ONE launch into one pooled buffer (views per detection);
        # more than 100 detections cannot occur per image (TEST.DETECTIONS_PER_IMAGE <= 100)
        live = [k for k in range(n) if want_masks and rects[k, 2] > rects[k, 0] and rects[k, 3] > rects[k, 1]]
        pool, offs = None, {}
        if live:
            nw = ((rects[live, 2] + 63) >> 6) - (rects[live, 0] >> 6)
            words = nw.astype(np.int64) * (rects[live, 3] - rects[live, 1])
            off = np.zeros(len(live) + 1, np.int64)
            np.cumsum(words, out=off[1:])
            pool = torch.empty((int(off[-1]),), dtype=torch.int64, device=self.device)
            dets = np.ascontiguousarray(rec["packed_index"][live], np.int32)
            lrects = np.ascontiguousarray(rects[live])
            _lib.check(lib.apse_copy_mask_windows(self._ctx, len(live), _lib.ptr(dets), _lib.ptr(lrects), _lib.ptr(pool),
                                                  _lib.ptr(np.ascontiguousarray(off[:-1])), _lib.stream_ptr()), self._ctx, "apse_copy_mask_windows")
            offs = {k: (int(off[j]), int(off[j + 1]), int(nw[j])) for j, k in enumerate(live)}
        for k in range(n):
            x0, y0, x1, y1 = [int(v) for v in rects[k]]
            bits = None
            if k in offs:
                lo, hi, w = offs[k]
                bits = pool[lo:hi].view(y1 - y0, w)
            masks.append(WindowMask(bits, (x0, y0, x1, y1), frame_hw, rec["centroids"][k], rec["mass"][k]))
        inst.pred_masks = masks
        inst._record = rec
        return inst

    def _inference_images(self, batched_inputs, detected_instances, do_postprocess, rpn_levels):
        """Shared body of ``inference`` (track_rcnn.py:16-58) and ``SelectiveMaskRCNN.scan`` (selective_rcnn.py:27-84):
        the reference's model input -- resized f32 CHW images -- through normalise + pad and the whole launch sequence."""
        assert not self.training
        if not do_postprocess:
            raise NotImplementedError("do_postprocess=False is not on the CSV path and is not provided")
        B = len(batched_inputs)
        imgs = torch.stack([bi["image"].to(torch.float32) for bi in batched_inputs]).to(self.device)
        frame_hw = (int(batched_inputs[0]["height"]), int(batched_inputs[0]["width"]))
        self.preprocess_images(imgs, frame_hw)
        given = None
        if detected_instances is not None:
            assert len(detected_instances) == B
            boxes = np.concatenate([d.pred_boxes.tensor.cpu().numpy().reshape(-1, 4) for d in detected_instances])
            classes = np.concatenate([np.asarray(d.pred_classes.cpu()).reshape(-1) for d in detected_instances])
            counts = np.asarray([len(d) for d in detected_instances], np.int32)
            given = (boxes, classes, counts)
        self.run(B, given, rpn_levels)
        res = self.read(B)
        return [{"instances": self.instances_from(res, b)} for b in range(B)], B

    def inference(self, batched_inputs, detected_instances=None, do_postprocess=True):
        """Same contract as track_rcnn.py:16-58: ``batched_inputs`` = list of
        {"image": f32 CHW tensor (resized, BGR), "height", "width"}; returns
        (list of {"instances": Instances}, feature dict).  ``detected_instances``: list of Instances
        with ``pred_boxes`` (resized-image pixels) and ``pred_classes`` -> box branch skipped."""
        out, B = self._inference_images(batched_inputs, detected_instances, do_postprocess, 31)
        return out, LazyFeatures(self, B)

    def inference_frames(self, frames, given=None, want_masks=True, rpn_levels=31):
        """Fused path: uint8 CUDA frames [B, H, W, 3] -> (list of Instances, feature dict)."""
        B = self.preprocess_frames(frames)
        self.run(B, given, rpn_levels)
        res = self.read(B)
        return [self.instances_from(res, b, want_masks) for b in range(B)], LazyFeatures(self, B)
