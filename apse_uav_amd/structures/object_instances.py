"""Track store -- counterpart of /root/reference/dcnn/structures/object_instances.py:11-176.

Same fields and semantics (lists per object: ``detected_this_frame, ids,
frames_since_detected, pred_boxes, scores, pred_classes, pred_masks, embeddings``);
``pred_masks`` is a list of :class:`WindowMask` instead of a stacked dense tensor.
Ids start at 1, are never reused (:48-52); ``associate_detection`` overwrites box,
class, mask and embedding but not the score (:146-152); objects undetected for more
than the threshold are deleted (:105-125); ``finish_association`` ages (:155-162).
"""
from .instances import Instances, SetBoxes
from .window_mask import MaskList

_FIELDS = ("detected_this_frame", "ids", "frames_since_detected", "pred_boxes", "scores", "pred_classes", "pred_masks",
           "embeddings")


class ObjectInstances(Instances):
    def __init__(self, image_size, display_info=(), metadata=None, **kwargs):
        super().__init__(image_size=image_size, **kwargs)
        self._assigned_ids = []
        self._display_info = list(display_info)
        self._metadata = metadata

    def __len__(self):
        if not self._fields:
            return 0
        for v in self._fields.values():
            return len(v)
        return 0

    def __str__(self):
        s = "objects: " + str(len(self)) + "\n"
        for k in range(len(self)):
            cls = int(self.pred_classes[k])
            name = self._metadata.get("thing_classes", None)[cls] if self._metadata is not None else cls
            s += "\tid: {}\tclass: {}\tundetected for: {}\n".format(self.ids[k], name, self.frames_since_detected[k])
        return s

    def get_new_id(self):
        if len(self._assigned_ids) == 0:
            return 1
        return self._assigned_ids[-1] + 1

    def to(self, device):
        ret = ObjectInstances(self._image_size)
        for k, v in self._fields.items():
            if hasattr(v, "to"):
                v = v.to(device)
            ret.set(k, v)
        return ret

    # ---- table-driven field handling: which per-object lists exist, and where each value comes from
    #  * bookkeeping fields get a fixed value whenever an object is (re)detected,
    #  * detection fields are copied from the detection that was associated,
    #  * ``scores`` is copied only when the object is created (frozen afterwards, reference :146-152),
    #  * ``embeddings`` exist only when the tracker supplies them.
    _ON_DETECTION = (("detected_this_frame", True), ("frames_since_detected", 0))
    _COPIED_ALWAYS = ("pred_boxes", "pred_classes", "pred_masks")
    _COPIED_AT_BIRTH = ("scores",)

    def _ensure_lists(self, with_embeddings):
        if len(self) != 0:
            return
        for name in _FIELDS:
            if name == "embeddings" and not with_embeddings:
                continue
            self._fields[name] = MaskList() if name == "pred_masks" else []

    @staticmethod
    def _detection_value(name, det_fields, detection_id, fresh_box):
        v = det_fields[name][detection_id]
        if name == "pred_boxes" and fresh_box:
            v = SetBoxes(v.tensor)            # an object owns its box (a detection's Boxes row is a view)
        return v

    def add_new_object(self, detection_id, detections, detection_embeddings=None, verbose=True):
        det_fields = detections.get_fields()
        new_id = self.get_new_id()
        if verbose and "new_objects" in self._display_info:
            print("adding detection_id: {} as new object with id: {}".format(detection_id, new_id))
        self._ensure_lists(detection_embeddings is not None)
        store = self._fields
        for name, value in self._ON_DETECTION:
            store[name].append(value)
        store["ids"].append(new_id)
        for name in self._COPIED_ALWAYS + self._COPIED_AT_BIRTH:
            store[name].append(self._detection_value(name, det_fields, detection_id, fresh_box=True))
        if detection_embeddings is not None and "embeddings" in store:
            store["embeddings"].append(detection_embeddings[detection_id])
        self._assigned_ids.append(new_id)

    def delete_undetected_objects(self, frames_threshold):
        stale = [k for k in range(len(self)) if self.frames_since_detected[k] > frames_threshold]
        for k in reversed(stale):
            for column in self._fields.values():
                del column[k]

    def associate_detection(self, detection_id, object_index, detections, detections_embeddings=None):
        if "associations" in self._display_info:
            print("associating detection {} to object id: {}".format(detection_id, self.ids[object_index]))
        det_fields = detections.get_fields()
        store = self._fields
        for name, value in self._ON_DETECTION:
            store[name][object_index] = value
        for name in self._COPIED_ALWAYS:
            store[name][object_index] = self._detection_value(name, det_fields, detection_id, fresh_box=False)
        if "embeddings" in store and detections_embeddings is not None:
            store["embeddings"][object_index] = detections_embeddings[detection_id]

    def finish_association(self):
        if len(self) == 0:
            return
        self._fields["frames_since_detected"] = [0 if self.detected_this_frame[k] else self.frames_since_detected[k] + 1
                                                 for k in range(len(self))]
        self._fields["detected_this_frame"] = [False] * len(self)

    def get_recent_objects(self):
        new = ObjectInstances(image_size=self._image_size, display_info=self._display_info, metadata=self._metadata)
        for k in range(len(self)):
            if self.detected_this_frame[k]:
                for name in self._fields.keys():
                    if name not in new._fields:
                        new._fields[name] = MaskList() if name == "pred_masks" else []
                    new._fields[name].append(self._fields[name][k])
        return new
