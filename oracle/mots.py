"""TEST INFRASTRUCTURE (see oracle/__init__.py) -- dense restatement of /root/reference/dcnn/utils/mots_evaluation.py.

Objects are plain dicts {"classes": [n], "ids": [n], "scores": [n], "masks": bool ndarray [n, H, W]}.
The RLE string of ``file_lines`` follows pycocotools' published format (not installed: parity of the codec is
pinned by the example line the reference quotes at mots_evaluation.py:13, see tests/test_host_logic.py).
"""
import numpy as np


def _cls(c):
    return {0: 2, 2: 1}.get(int(c))           # :31-37


def rle_string(mask):
    """maskApi.c rleEncode + rleToString on one [H, W] mask."""
    flat = np.asarray(mask, dtype=np.uint8).T.reshape(-1)        # column-major
    counts, prev, run = [], 0, 0
    for v in flat.tolist():
        if v != prev:
            counts.append(run)
            run, prev = 0, v
        run += 1
    counts.append(run)
    s = ""
    for i, x in enumerate(counts):
        if i > 2:
            x -= counts[i - 2]
        more = True
        while more:
            c = x & 0x1F
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            s += chr(c + 48)
    return s


def file_lines(objs, frame_num, image_size):          # :25-55
    out = ""
    for k in range(len(objs["ids"])):
        c = _cls(objs["classes"][k])
        if c is None:
            continue
        out += "%d %d %d %d %d %s\n" % (frame_num, c * 1000 + objs["ids"][k], c, image_size[0], image_size[1],
                                        rle_string(objs["masks"][k]))
    return out


def result_image(objs, image_size):                   # :58-77
    img = np.zeros(image_size, dtype=np.uint16)
    for k in range(len(objs["ids"])):
        c = _cls(objs["classes"][k])
        if c is None:
            continue
        img[objs["masks"][k]] = c * 1000 + objs["ids"][k]
    return img


def crop_overlapping(objs):                           # :97-117 (in place; later pairs see earlier crops)
    m = objs["masks"]
    n = len(m)
    for i in range(n):
        for j in range(i + 1, n):
            inter = m[i] & m[j]
            if inter.any():
                if objs["scores"][i] > objs["scores"][j]:
                    m[j] = np.logical_xor(m[j], inter)
                else:
                    m[i] = np.logical_xor(m[i], inter)
