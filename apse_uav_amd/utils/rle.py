"""COCO run-length masks (the format of pycocotools ``encode`` / ``decode``).

The reference reads ground-truth instance masks as COCO RLE dicts
(/root/reference/dcnn/engines/roi_features_generator.py:93-97) and writes MOTS text lines with the compressed
``counts`` string (/root/reference/dcnn/utils/mots_evaluation.py:43-52).  pycocotools is not a dependency of this
build; this module restates its published format: column-major (Fortran) runs that start with a run of zeros,
each count stored as the difference to the count two positions back (from the fourth on), in 5-bit groups with a
continuation bit, offset by 48 into printable ASCII.
"""
import numpy as np


def counts_from_mask(mask):
    """mask: [H, W] bool / u8 -> list of run lengths over the column-major pixel order, first run = zeros."""
    m = np.asarray(mask)
    flat = np.asfortranarray(m != 0).reshape(-1, order="F")
    if flat.size == 0:
        return []
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    bounds = np.concatenate(([0], change, [flat.size]))
    counts = np.diff(bounds).tolist()
    if flat[0]:
        counts = [0] + counts
    return counts


def mask_from_counts(counts, h, w):
    runs = np.asarray(counts, dtype=np.int64)
    assert runs.sum() == h * w, "run lengths do not cover the image"
    vals = (np.arange(len(runs)) & 1).astype(np.uint8)
    return np.repeat(vals, runs).reshape((h, w), order="F")


def counts_to_string(counts):
    out = []
    for i, x in enumerate(counts):
        x = int(x)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            c = x & 0x1F
            x >>= 5                                   # arithmetic shift: negative differences sign-extend
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(chr(c + 48))
    return "".join(out)


def string_to_counts(s):
    if isinstance(s, bytes):
        s = s.decode("ascii")
    counts = []
    p = 0
    while p < len(s):
        x = 0
        k = 0
        more = True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def encode(mask):
    """pycocotools.mask.encode for one [H, W] mask -> {"size": [h, w], "counts": bytes}."""
    h, w = np.asarray(mask).shape
    return {"size": [int(h), int(w)], "counts": counts_to_string(counts_from_mask(mask)).encode("ascii")}


def decode(rle):
    """pycocotools.mask.decode for one RLE dict (compressed string or plain list of counts) -> u8 [H, W]."""
    h, w = rle["size"]
    counts = rle["counts"]
    if not isinstance(counts, (list, tuple)):
        counts = string_to_counts(counts)
    return mask_from_counts(counts, int(h), int(w))
