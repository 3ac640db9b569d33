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
        return self._inference_images(batched_inputs, detected_instances, do_postprocess, LAST_LEVEL_ONLY)[0]
