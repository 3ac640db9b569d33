"""TrackPredictor -- counterpart of /root/reference/dcnn/engines/track_predictor.py:11-52.

Same constructor (``cfg``) and call contract: ``predictor(original_image: HxWx3 uint8 BGR)`` ->
``(predictions, backbone_features)`` with ``predictions["instances"]`` and the FPN dict.  The
reference resizes on the CPU with PIL and uploads a 12 MB f32 image; here the u8 frame is uploaded
once (24.9 MB at 4K) and the PIL-exact resize, normalisation and padding run as HIP kernels.
Weights: ``cfg.MODEL.WEIGHTS`` (a ``.pth`` with key "model" or a bare state_dict) or
``load_state_dict``.
"""
import numpy as np
import torch

from ..networks.track_rcnn import TrackRCNN
from ..weights import load_detector_file


class TrackPredictor:
    def __init__(self, cfg, state_dict=None):
        self.cfg = cfg.clone()
        self.model = TrackRCNN(self.cfg)
        self.model.to(torch.device(cfg.MODEL.DEVICE))
        self.model.eval()
        if state_dict is not None:
            self.model.load_state_dict(state_dict)
        elif cfg.MODEL.WEIGHTS:
            self.model.load_state_dict(load_detector_file(cfg.MODEL.WEIGHTS))
        self.input_format = cfg.INPUT.FORMAT
        assert self.input_format in ["RGB", "BGR"], self.input_format
        self._staging = None
        self.frame_preprocessor = None        # optional FramePreprocessor (cfg.APSE.FUSED_PREPROC / set_camera)

    def set_camera(self, cam_params, gamma=2.0):
        """Enables undistort + Lab-gamma in front of the resize (preprocess_img, visualize_uav.py:56-71)."""
        from ..utils.preprocess import FramePreprocessor
        self.frame_preprocessor = FramePreprocessor(cam_params, gamma)

    def _upload(self, frames):
        """list of HxWx3 uint8 arrays -> CUDA tensor [B, H, W, 3] through a pinned staging buffer."""
        B = len(frames)
        H, W = frames[0].shape[:2]
        if self._staging is None or self._staging.shape != (B, H, W, 3):
            self._staging = torch.empty((B, H, W, 3), dtype=torch.uint8).pin_memory()
        for i, f in enumerate(frames):
            if self.input_format == "RGB":
                f = f[:, :, ::-1]
            self._staging[i].copy_(torch.from_numpy(np.ascontiguousarray(f)))
        dev = self._staging.to(self.model.device, non_blocking=True)
        if self.frame_preprocessor is not None:
            dev = self.frame_preprocessor(dev)
        return dev

    def __call__(self, original_image):
        with torch.no_grad():
            dev = self._upload([original_image])
            insts, feats = self.model.inference_frames(dev)
            return {"instances": insts[0]}, feats

    def predict_batch(self, frames, given=None, want_masks=True):
        """Build extension (reference is batch 1): several frames in one forward."""
        with torch.no_grad():
            dev = self._upload(frames)
            insts, feats = self.model.inference_frames(dev, given=given, want_masks=want_masks)
            return [{"instances": i} for i in insts], feats
