"""SelectivePredictor -- counterpart of /root/reference/dcnn/engines/selective_predictor.py:11-51:
``predictor(bgr_u8) -> predictions`` (no feature dict) from ``SelectiveMaskRCNN.scan``."""
import torch

from ..networks.selective_rcnn import LAST_LEVEL_ONLY, SelectiveMaskRCNN
from .track_predictor import TrackPredictor


class SelectivePredictor(TrackPredictor):
    model_class = SelectiveMaskRCNN

    def __call__(self, original_image, upcoming=None):
        with torch.no_grad():
            insts, _ = self._predict([original_image], upcoming=upcoming, rpn_levels=LAST_LEVEL_ONLY)
            return {"instances": insts[0]}
