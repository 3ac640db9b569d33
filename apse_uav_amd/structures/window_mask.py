"""Box-local instance masks.

The reference keeps every instance mask as a dense ``2160 x 3840`` bool tensor
(detectron2 ``paste_masks_in_image`` output, consumed by
/root/reference/dcnn/utils/mask_utils.py and stacked in
dcnn/structures/object_instances.py:85-97).  The HIP path produces the identical pixel
set bit-packed inside the paste window only; ``WindowMask`` carries that window
(64 pixels per int64 word, rows ``y0..y1``, word columns ``x0>>6 ..``) together with the
mask's mass and 1-based floor centroid computed on the GPU.  ``dense()`` materialises the
reference's representation on demand (API compatibility, not on the measured path).
"""
import torch


class WindowMask:
    __slots__ = ("bits", "rect", "frame_size", "centroid", "mass")

    def __init__(self, bits, rect, frame_size, centroid, mass):
        self.bits = bits                  # int64 [rows, words] on the device (or None for an empty window)
        self.rect = tuple(int(v) for v in rect)          # x0, y0, x1, y1
        self.frame_size = tuple(frame_size)              # (H, W)
        self.centroid = (float(centroid[0]), float(centroid[1])) if centroid[0] >= 0 else (float("nan"), float("nan"))
        self.mass = int(mass)

    def size(self):
        return torch.Size(self.frame_size)

    @property
    def shape(self):
        return torch.Size(self.frame_size)

    @property
    def device(self):
        return self.bits.device if self.bits is not None else torch.device("cpu")

    def window(self):
        """Bool tensor of the paste window [y1-y0, x1-x0] (device of ``bits``)."""
        x0, y0, x1, y1 = self.rect
        if self.bits is None or x1 <= x0 or y1 <= y0:
            return torch.zeros((max(y1 - y0, 0), max(x1 - x0, 0)), dtype=torch.bool)
        sh = torch.arange(64, device=self.bits.device, dtype=torch.int64)
        px = ((self.bits.unsqueeze(-1) >> sh) & 1).to(torch.bool).reshape(self.bits.shape[0], -1)
        off = x0 - ((x0 >> 6) << 6)
        return px[:, off:off + (x1 - x0)]

    def dense(self):
        H, W = self.frame_size
        x0, y0, x1, y1 = self.rect
        win = self.window()
        out = torch.zeros((H, W), dtype=torch.bool, device=win.device)
        if win.numel():
            out[y0:y1, x0:x1] = win
        return out

    def cpu(self):
        return self.dense().cpu()

    def to(self, device):
        return WindowMask(None if self.bits is None else self.bits.to(device), self.rect, self.frame_size,
                          (self.centroid[0], self.centroid[1]) if self.mass else (-1, -1), self.mass)

    def sum(self):
        return self.mass


class MaskList(list):
    """List of WindowMask with the tensor-like helpers the reference scripts use on ``pred_masks``."""

    def dense(self):
        if len(self) == 0:
            return torch.zeros((0, 0, 0), dtype=torch.bool)
        return torch.stack([m.dense() for m in self])

    def size(self):
        if len(self) == 0:
            return torch.Size((0, 0, 0))
        return torch.Size((len(self),) + tuple(self[0].frame_size))

    def to(self, device):
        return MaskList(m.to(device) for m in self)
