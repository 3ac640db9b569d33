"""Lean replay of the sequential association from per-frame records.

In frame-sharded runs rank 0 receives every rank's records with one gather at sequence end and must then
run the id assignment for N x K frames on the host.  ``RcnnTracker.next_record`` + ``log_line`` do that
through the reference-shaped containers (``Instances`` / ``ObjectInstances`` / ``WindowMask``), ~200 us per
frame; ``FastReplay`` applies exactly the same rules on plain lists / numpy arrays (~10x cheaper), so the
replay does not eat the scaling of an 8-GPU run.  Rules (all from the reference):
  * first detections become objects 1..N in detection order      (dcnn/engines/rcnn_tracker.py:126-128)
  * squared-L2 distance matrix, scipy Hungarian, ``dist < 0.6``   (:130-143)
  * unmatched detections -> new ids, ascending detection index    (:145-147, object_instances.py:48-52)
  * objects unseen for more than 100 frames are dropped           (:70, object_instances.py:105-125)
  * CSV line: ids 1..max present this frame, blanks otherwise     (scripts/tests/visualize_uav.py:117-141)
tests/test_replay.py checks it line-for-line against the RcnnTracker path.
"""
import numpy as np
from scipy.optimize import linear_sum_assignment

EMBEDDING_THRESHOLD = 0.6
UNDETECTED_FRAMES_TH = 100


class FastReplay:
    def __init__(self, host_id):
        self.host_id = host_id
        self.ids, self.since, self.emb = [], [], []
        self.next_id = 1
        self.frame_count = 0
        self.max_id = 0

    def step(self, rec, frame_idx):
        """Consumes one record, returns (csv_line, ids_seen_this_frame)."""
        self.frame_count += 1
        n = len(rec["scores"])
        ids, since, emb = self.ids, self.since, self.emb
        det_of = {}                                    # object slot -> detection index (this frame)
        if n > 0:
            E = np.asarray(rec["embeddings"], np.float32)
            if not ids:
                for d in range(n):
                    ids.append(self.next_id); since.append(0); emb.append(E[d]); det_of[len(ids) - 1] = d
                    self.next_id += 1
            else:
                O = np.stack(emb)
                diff = O[:, None, :] - E[None, :, :]
                D = (diff * diff).sum(axis=2, dtype=np.float32)
                oi, di = linear_sum_assignment(D)
                matched = set()
                for o, d in zip(oi.tolist(), di.tolist()):
                    if D[o, d] < EMBEDDING_THRESHOLD:
                        emb[o] = E[d]; since[o] = 0; det_of[o] = d
                        matched.add(d)
                for d in range(n):
                    if d not in matched:
                        ids.append(self.next_id); since.append(0); emb.append(E[d]); det_of[len(ids) - 1] = d
                        self.next_id += 1
        # delete_undetected_objects(100): frames_since_detected is still last frame's value for unseen objects
        if any(s > UNDETECTED_FRAMES_TH for s in since):
            keep = [k for k in range(len(ids)) if not since[k] > UNDETECTED_FRAMES_TH]
            remap = {k: j for j, k in enumerate(keep)}
            det_of = {remap[k]: d for k, d in det_of.items() if k in remap}
            self.ids = ids = [ids[k] for k in keep]
            self.since = since = [since[k] for k in keep]
            self.emb = emb = [emb[k] for k in keep]
        seen = sorted(det_of)                          # store order == get_recent_objects order
        # finish_association: ageing
        for k in range(len(ids)):
            since[k] = 0 if k in det_of else since[k] + 1
        if not seen:
            return "", []
        seen_ids = [ids[k] for k in seen]
        cent, clos = rec["centroids"], rec["closest"]
        hi = max(seen_ids)
        self.max_id = max(self.max_id, hi)
        by_id = {ids[k]: det_of[k] for k in seen}
        hdet = by_id.get(self.host_id)
        cells = [str(frame_idx)]
        for oid in range(1, hi + 1):
            d = by_id.get(oid)
            if d is None:
                cells += ["", "", "", ""]
                continue
            cx, cy = int(cent[d][0]), int(cent[d][1])
            cells.append("%d.0" % cx if cx >= 0 else "nan")
            cells.append("%d.0" % cy if cx >= 0 else "nan")
            if hdet is None:
                cells += ["nan", "nan"]
            else:
                c = clos[d][hdet]
                cells.append("%d.0" % int(c[0]))
                cells.append("%d.0" % int(c[1]))
        return ",".join(cells), seen_ids


class NativeReplay:
    """The same replay in C++ (``apse_replay_*`` in libapse_hip.so, host-only): a few microseconds per frame."""

    def __init__(self, host_id, embed_dim=128):
        import ctypes as C
        from .. import _lib
        self._C, self._lib = C, _lib.load()
        self._h = self._lib.apse_replay_create(int(host_id), int(embed_dim), float(EMBEDDING_THRESHOLD), int(UNDETECTED_FRAMES_TH))
        if not self._h:
            raise _lib.ApseError("apse_replay_create failed")
        self._buf = C.create_string_buffer(1 << 16)
        self.edim = embed_dim

    def __del__(self):
        try:
            self._lib.apse_replay_destroy(self._h)
        except Exception:
            pass

    @property
    def max_id(self):
        return int(self._lib.apse_replay_max_id(self._h))

    @property
    def next_id(self):
        return int(self._lib.apse_replay_next_id(self._h))

    def step(self, rec, frame_idx):
        n = len(rec["scores"])
        emb = np.ascontiguousarray(rec["embeddings"], np.float32).reshape(n, self.edim)
        cent = np.ascontiguousarray(rec["centroids"], np.int32).reshape(n, 2)
        clos = np.ascontiguousarray(rec["closest"], np.int32).reshape(n, n, 2)
        ids = np.full((max(n, 1),), -1, np.int32)
        rc = self._lib.apse_replay_step(self._h, int(frame_idx), n, emb.ctypes.data, cent.ctypes.data, clos.ctypes.data,
                                        self._buf, len(self._buf), ids.ctypes.data)
        if rc < 0:
            raise RuntimeError("apse_replay_step failed (%d)" % rc)
        return self._buf.value.decode(), sorted(int(v) for v in ids[:n] if v > 0)

    def run_packed(self, packed, nrec=None, kd=100, first_frame=0, chunk=512):
        """packed: the gather's wire format -- a flat float32 array of ``nrec`` count-prefixed records laid end to end, or the
        ``(flat, nrec)`` pair ``gather_records(unpack=False)`` returns -> list of CSV lines.  Walked in chunks of records so the
        line buffer can be sized from the ids issued so far (a line has a cell group for every id up to the frame's largest)."""
        if isinstance(packed, tuple):
            packed, nrec = packed
        packed = np.ascontiguousarray(packed, np.float32).reshape(-1)
        nrec = int(nrec)
        lines, o, done = [], 0, 0
        while done < nrec:
            k, e, dets = 0, o, 0
            while k < chunk and done + k < nrec:
                if e >= packed.size:
                    raise RuntimeError("wire format: %d records announced, data ends in record %d" % (nrec, done + k))
                n = int(packed[e])
                e += 1 + 13 * n + 2 * n * n + n * self.edim
                if n < 0 or n > kd or e > packed.size:
                    raise RuntimeError("wire format: record %d announces %d detections, %d floats past the end" % (done + k, n, e - packed.size))
                dets += n
                k += 1
            cap = (1 << 12) + k * (16 + 4 * (self.max_id + dets)) + 40 * dets
            buf = self._C.create_string_buffer(cap)
            w = self._lib.apse_replay_packed(self._h, packed[o:].ctypes.data, int(e - o), k, int(kd), int(first_frame + done), buf, cap)
            if w < 0:
                raise RuntimeError("apse_replay_packed failed (%d)" % w)
            lines += buf.raw[:w].decode().split("\n")[:-1]
            o, done = e, done + k
        return lines
