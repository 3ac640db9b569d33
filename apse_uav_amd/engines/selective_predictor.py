"""SelectivePredictor -- counterpart of /root/reference/dcnn/engines/selective_predictor.py:11-51:
``predictor(bgr_u8) -> predictions`` (no feature dict) from ``SelectiveMaskRCNN.scan``."""
import torch

from ..networks.selective_rcnn import LAST_LEVEL_ONLY, SelectiveMaskRCNN
from ..weights import load_detector_file
from .track_predictor import TrackPredictor


class SelectivePredictor(TrackPredictor):
    def __init__(self, cfg, state_dict=None):
        self.cfg = cfg.clone()
        self.model = SelectiveMaskRCNN(self.cfg)
        self.model.to(torch.device(cfg.MODEL.DEVICE))
        self.model.eval()
        if state_dict is not None:
            self.model.load_state_dict(state_dict)
        elif cfg.MODEL.WEIGHTS:
            self.model.load_state_dict(load_detector_file(cfg.MODEL.WEIGHTS))
        self.input_format = cfg.INPUT.FORMAT
        assert self.input_format in ["RGB", "BGR"], self.input_format
        self._staging = None
        self.frame_preprocessor = None

    def __call__(self, original_image):
        with torch.no_grad():
            dev = self._upload([original_image])
            insts, _ = self.model.inference_frames(dev, rpn_levels=LAST_LEVEL_ONLY)
            return {"instances": insts[0]}
