"""Minimal stand-ins for detectron2's ``Boxes`` / ``Instances`` containers.

The reference's engines exchange detections through these two detectron2 classes
(/root/reference/dcnn/engines/rcnn_tracker.py:15-16,
dcnn/structures/object_instances.py:7, dcnn/structures/set_boxes.py:7).  detectron2 is
not a dependency of this build, so the subset of their surface the path touches is
provided here with the same names and semantics.
"""
import torch


class Boxes:
    def __init__(self, tensor):
        tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape(0, 4)
        if tensor.dim() == 1:
            tensor = tensor.reshape(1, 4)
        self.tensor = tensor

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, item):
        if isinstance(item, int):
            return type(self)(self.tensor[item].view(1, -1))
        return type(self)(self.tensor[item])

    def __setitem__(self, item, value):                    # SetBoxes.__setitem__ (set_boxes.py:14-16)
        self.tensor[item] = value.tensor

    def __iter__(self):
        yield from self.tensor

    def __repr__(self):
        return "Boxes(" + str(self.tensor) + ")"

    def to(self, device):
        return type(self)(self.tensor.to(device))

    def clone(self):
        return type(self)(self.tensor.clone())

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def get_centers(self):
        return (self.tensor[:, :2] + self.tensor[:, 2:]) / 2

    def clip(self, box_size):
        h, w = box_size
        self.tensor[:, 0].clamp_(min=0, max=w)
        self.tensor[:, 1].clamp_(min=0, max=h)
        self.tensor[:, 2].clamp_(min=0, max=w)
        self.tensor[:, 3].clamp_(min=0, max=h)

    def scale(self, scale_x, scale_y):
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    def nonempty(self, threshold=0.0):
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    @classmethod
    def cat(cls, boxes_list):
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))


SetBoxes = Boxes          # the reference's SetBoxes only adds __setitem__/cat, both provided above


class Instances:
    def __init__(self, image_size, **kwargs):
        object.__setattr__(self, "_image_size", tuple(image_size))
        object.__setattr__(self, "_fields", {})
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            object.__setattr__(self, name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError("Cannot find field '{}' in the given Instances!".format(name))
        return self._fields[name]

    def set(self, name, value):
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def remove(self, name):
        del self._fields[name]

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def to(self, device):
        ret = type(self)(self._image_size)
        for k, v in self._fields.items():
            if hasattr(v, "to"):
                v = v.to(device)
            ret.set(k, v)
        return ret

    def __getitem__(self, item):
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        return 0

    def __str__(self):
        s = self.__class__.__name__ + "(num_instances={}, image_height={}, image_width={}, fields=[{}])".format(
            len(self), self._image_size[0], self._image_size[1], ", ".join(self._fields.keys()))
        return s

    __repr__ = __str__
