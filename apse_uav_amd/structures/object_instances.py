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

    def add_new_object(self, detection_id, detections, detection_embeddings=None, verbose=True):
        d = detections.get_fields()
        o = self._fields
        new_id = self.get_new_id()
        if verbose and "new_objects" in self._display_info:
            print("adding detection_id: {} as new object with id: {}".format(detection_id, new_id))
        if len(self) == 0:
            o["detected_this_frame"] = []
            o["ids"] = []
            o["frames_since_detected"] = []
            o["pred_boxes"] = []
            o["scores"] = []
            o["pred_classes"] = []
            o["pred_masks"] = MaskList()
            if detection_embeddings is not None:
                o["embeddings"] = []
        o["detected_this_frame"].append(True)
        o["ids"].append(new_id)
        o["frames_since_detected"].append(0)
        o["pred_boxes"].append(SetBoxes(d["pred_boxes"][detection_id].tensor))
        o["scores"].append(d["scores"][detection_id])
        o["pred_classes"].append(d["pred_classes"][detection_id])
        o["pred_masks"].append(d["pred_masks"][detection_id])
        if detection_embeddings is not None and "embeddings" in o:
            o["embeddings"].append(detection_embeddings[detection_id])
        self._assigned_ids.append(new_id)

    def delete_undetected_objects(self, frames_threshold):
        if len(self) == 0:
            return
        drop = [k for k in range(len(self)) if self.frames_since_detected[k] > frames_threshold]
        for k in sorted(drop, reverse=True):
            for name in _FIELDS:
                if name in self._fields:
                    del self._fields[name][k]

    def associate_detection(self, detection_id, object_index, detections, detections_embeddings=None):
        if "associations" in self._display_info:
            print("associating detection {} to object id: {}".format(detection_id, self.ids[object_index]))
        d = detections.get_fields()
        o = self._fields
        o["detected_this_frame"][object_index] = True
        o["frames_since_detected"][object_index] = 0
        o["pred_boxes"][object_index] = d["pred_boxes"][detection_id]
        o["pred_classes"][object_index] = d["pred_classes"][detection_id]
        o["pred_masks"][object_index] = d["pred_masks"][detection_id]
        if "embeddings" in o and detections_embeddings is not None:
            o["embeddings"][object_index] = detections_embeddings[detection_id]

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
