"""SelectiveMaskRCNN / SelectiveRPN -- counterparts of /root/reference/dcnn/networks/selective_rcnn.py:15-84
and dcnn/networks/selective_rpn.py:7-86.

The reference's experimental variant runs the normal backbone and RPN head but hands only the LAST
pyramid level (p6) to ``find_top_rpn_proposals`` (selective_rpn.py:47-48), then the standard ROI heads.
Here that is the same launch sequence with ``apse_rpn_levels(level_mask=16)``; ``scan`` keeps the
reference's signature and returns the post-processed results only (no feature dict), like the original.
The reference's timing prints are not reproduced.
"""
from .track_rcnn import TrackRCNN

LAST_LEVEL_ONLY = 1 << 4          # bit 0 = p2 ... bit 4 = p6


class SelectiveMaskRCNN(TrackRCNN):
    def scan(self, batched_inputs, detected_instances=None, do_postprocess=True):
        import numpy as np
        import torch
        assert not self.training
        if not do_postprocess:
            raise NotImplementedError("do_postprocess=False is not provided")
        B = len(batched_inputs)
        imgs = torch.stack([bi["image"].to(torch.float32) for bi in batched_inputs]).to(self.device)
        frame_hw = (int(batched_inputs[0]["height"]), int(batched_inputs[0]["width"]))
        self.preprocess_images(imgs, frame_hw)
        given = None
        if detected_instances is not None:
            boxes = np.concatenate([d.pred_boxes.tensor.cpu().numpy().reshape(-1, 4) for d in detected_instances])
            classes = np.concatenate([np.asarray(d.pred_classes.cpu()).reshape(-1) for d in detected_instances])
            given = (boxes, classes, np.asarray([len(d) for d in detected_instances], np.int32))
        self.run(B, given, rpn_levels=LAST_LEVEL_ONLY)
        res = self.read(B)
        return [{"instances": self.instances_from(res, b)} for b in range(B)]
