#!/usr/bin/env python3
"""Generate golden vectors from the reference's own Python (build container only).

Run from the repo root:  python tests/golden/make_golden.py
Needs /root/reference (read-only).  The reference never travels to the GPU box:
only the small data files this script writes are committed.

What is imported from the reference:
  dcnn/networks/association_head.py  AssociationHead          (pure torch)
  dcnn/utils/mask_utils.py           get_mask_centroid, compute_closest_point
      mask_utils does `import cv2` at module level for its debug `show_mask`; cv2 is
      not installed, so an *empty* placeholder module is registered under that name
      for the import only.  Neither function under test touches cv2.
Data fixtures copied from the reference's data/ directory (data, not source):
  header lines + sample rows of static_dcnn_data.csv / dynamic_dcnn_data.csv, cam_params.json
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def formula_tensor(shape, a, b, mod, scale):
    """Closed-form integer pattern -> exactly reproducible f32 values anywhere."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.int64)
    v = ((i * a + (i // 7) * b) % mod).astype(np.float32) - np.float32(mod // 2)
    return torch.from_numpy((v / np.float32(scale)).reshape(shape))


def make_mask(h, w, spec):
    m = np.zeros((h, w), bool)
    for s in spec:
        if s["kind"] == "rect":
            m[s["y0"]:s["y1"], s["x0"]:s["x1"]] = True
        elif s["kind"] == "ellipse":
            yy, xx = np.ogrid[:h, :w]
            m |= (((xx - s["cx"]) / s["rx"]) ** 2 + ((yy - s["cy"]) / s["ry"]) ** 2) <= 1.0
        elif s["kind"] == "pixel":
            m[s["y"], s["x"]] = True
    return m


MASK_CASES = [
    dict(name="rect_mid", spec=[dict(kind="rect", x0=1800, x1=2050, y0=900, y1=1010)], point=(1911.0, 966.0)),
    dict(name="rect_far", spec=[dict(kind="rect", x0=100, x1=350, y0=1300, y1=1410)], point=(1911.0, 966.0)),
    dict(name="ellipse", spec=[dict(kind="ellipse", cx=3388, cy=1020, rx=125, ry=55)], point=(1911.0, 966.0)),
    dict(name="two_blob", spec=[dict(kind="rect", x0=200, x1=330, y0=450, y1=540),
                                dict(kind="ellipse", cx=420, cy=560, rx=40, ry=25)], point=(1911.0, 966.0)),
    dict(name="single_px", spec=[dict(kind="pixel", x=3839, y=2159)], point=(1.0, 1.0)),
    dict(name="border_tl", spec=[dict(kind="rect", x0=0, x1=97, y0=0, y1=61)], point=(3840.0, 2160.0)),
    dict(name="border_br", spec=[dict(kind="rect", x0=3700, x1=3840, y0=2100, y1=2160)], point=(1.0, 1.0)),
    dict(name="tie_sym", spec=[dict(kind="rect", x0=1000, x1=1201, y0=500, y1=601)], point=(1101.0, 300.0)),
    dict(name="tie_two", spec=[dict(kind="pixel", x=1500, y=700), dict(kind="pixel", x=1700, y=700),
                               dict(kind="pixel", x=1600, y=800)], point=(1601.0, 701.0)),
    dict(name="far_f32", spec=[dict(kind="ellipse", cx=3700, cy=2000, rx=111, ry=77)], point=(3.0, 5.0)),
    dict(name="inside", spec=[dict(kind="ellipse", cx=1911, cy=966, rx=130, ry=60)], point=(1911.0, 966.0)),
]


def main():
    sys.path.insert(0, os.path.join(REF, "dcnn"))
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")      # empty placeholder, see module docstring
    from networks.association_head import AssociationHead
    from utils.mask_utils import get_mask_centroid, compute_closest_point

    # ---- AssociationHead: full-size head with closed-form weights, plus a small seeded case
    head = AssociationHead(roi_size=10, input_depth=256)
    w = formula_tensor((128, 25600), 131, 71, 257, 8192.0)
    b = formula_tensor((128,), 17, 5, 61, 64.0)
    head.load_state_dict({"fc.weight": w, "fc.bias": b})
    x = formula_tensor((3, 256, 10, 10), 37, 11, 509, 97.0)
    x[1] = torch.relu(x[1])
    x[2] = 0.0                                             # zero row: exercises the eps clamp of F.normalize
    with torch.no_grad():
        y = head(x)
    torch.manual_seed(1234)
    small = AssociationHead(roi_size=10, input_depth=8)
    xs = torch.randn(5, 8, 10, 10)
    with torch.no_grad():
        ys = small(xs)
    np.savez_compressed(os.path.join(HERE, "association_head_golden.npz"),
                        full_out=y.numpy(), small_w=small.fc.weight.detach().numpy(),
                        small_b=small.fc.bias.detach().numpy(), small_x=xs.numpy(), small_out=ys.numpy())

    # ---- mask utils on full 2160x3840 frames
    H, W = 2160, 3840
    res = []
    for c in MASK_CASES:
        m = torch.from_numpy(make_mask(H, W, c["spec"]))
        cen = get_mask_centroid(m)
        clo = compute_closest_point(m, c["point"])
        clo_self = compute_closest_point(m, cen)
        res.append(dict(name=c["name"], spec=c["spec"], point=list(c["point"]), centroid=list(cen),
                        closest=list(clo), closest_to_own_centroid=list(clo_self), mass=int(m.sum())))
    with open(os.path.join(HERE, "mask_utils_golden.json"), "w") as f:
        json.dump(dict(height=H, width=W, cases=res), f, indent=1)

    # ---- data fixtures (schema): header + sample rows, incl. rows with blank cells
    for name in ("static", "dynamic"):
        with open(os.path.join(REF, "data", name + "_dcnn_data.csv")) as f:
            lines = f.read().split("\n")
        keep = lines[:7]
        if name == "dynamic":
            keep += lines[2 + 865:2 + 870] + lines[2 + 1129:2 + 1133]
        with open(os.path.join(HERE, name + "_dcnn_data_head.csv"), "w") as f:
            f.write("\n".join(keep) + "\n")
        meta = dict(n_lines=len(lines), trailing_newline=(lines[-1] == ""),
                    n_rows=len([l for l in lines[2:] if l != ""]))
        with open(os.path.join(HERE, name + "_dcnn_data_meta.json"), "w") as f:
            json.dump(meta, f)
    with open(os.path.join(REF, "data", "cam_params.json")) as f:
        cam = json.load(f)
    with open(os.path.join(HERE, "cam_params.json"), "w") as f:
        json.dump(cam, f)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
