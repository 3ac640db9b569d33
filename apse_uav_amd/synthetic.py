"""Synthetic 3840x2160 UAV-like frames (no video ships with the reference).

SURVEY.md 8d: smooth background + K filled rotated rectangles (~250x110 px,
distinct colours).  "static": no motion; "dynamic": linear motion, one vehicle
leaves the frame for a span so blank CSV cells occur (as in
/root/reference/data/dynamic_dcnn_data.csv, blank id_1 cells at frames 867-1212).
Frames are u8 HxWx3 BGR like ``cv2.VideoCapture.read`` returns
(/root/reference/dcnn/scripts/tests/visualize_uav.py:188).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

COLORS = [(40, 40, 200), (200, 60, 40), (60, 190, 60), (220, 220, 230), (30, 30, 30), (200, 180, 40)]


class SyntheticSequence:
    def __init__(self, kind="static", height=2160, width=3840, n_vehicles=4, seed=1234):
        self.kind, self.h, self.w, self.k = kind, height, width, n_vehicles
        g = torch.Generator().manual_seed(seed)
        sy, sx = height / 2160.0, width / 3840.0
        coarse = torch.rand(1, 3, 34, 60, generator=g) * 120.0 + 60.0
        bg = F.interpolate(coarse, size=(height, width), mode="bilinear", align_corners=False)[0]
        self.bg = bg.permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).numpy()
        self.veh = []
        for i in range(n_vehicles):
            cx = (0.12 + 0.76 * (i + 0.5) / n_vehicles) * width + float(torch.rand(1, generator=g)) * 40 * sx
            cy = (0.25 + 0.5 * float(torch.rand(1, generator=g))) * height
            ang = (float(torch.rand(1, generator=g)) - 0.5) * 0.6
            vx = (1.0 + float(torch.rand(1, generator=g))) * (1 if i % 2 == 0 else -1) * sx
            vy = (float(torch.rand(1, generator=g)) - 0.5) * sy
            self.veh.append(dict(cx=cx, cy=cy, ang=ang, vx=vx, vy=vy, L=250.0 * sx, Wd=110.0 * sy,
                                 color=COLORS[i % len(COLORS)]))

    def pose(self, v, t, idx):
        if self.kind == "static":
            return v["cx"], v["cy"], True
        cx, cy = v["cx"] + v["vx"] * t, v["cy"] + v["vy"] * t
        visible = not (idx == 1 and 20 <= t < 40)       # vehicle 1 disappears for a span
        return cx, cy, visible

    def boxes(self, t):
        """Axis-aligned boxes (x1, y1, x2, y2) of the visible vehicles in frame t, f32."""
        out = []
        for i, v in enumerate(self.veh):
            cx, cy, vis = self.pose(v, t, i)
            if not vis:
                continue
            c, s = abs(math.cos(v["ang"])), abs(math.sin(v["ang"]))
            hw = 0.5 * (v["L"] * c + v["Wd"] * s)
            hh = 0.5 * (v["L"] * s + v["Wd"] * c)
            out.append([max(cx - hw, 0.0), max(cy - hh, 0.0), min(cx + hw, float(self.w)), min(cy + hh, float(self.h))])
        return np.asarray(out, np.float32).reshape(-1, 4)

    def frame(self, t):
        img = self.bg.copy()
        for i, v in enumerate(self.veh):
            cx, cy, vis = self.pose(v, t, i)
            if not vis:
                continue
            r = 0.5 * math.hypot(v["L"], v["Wd"]) + 2
            x0, x1 = max(int(cx - r), 0), min(int(cx + r) + 1, self.w)
            y0, y1 = max(int(cy - r), 0), min(int(cy + r) + 1, self.h)
            if x1 <= x0 or y1 <= y0:
                continue
            yy, xx = np.mgrid[y0:y1, x0:x1]
            dx, dy = xx + 0.5 - cx, yy + 0.5 - cy
            ca, sa = math.cos(v["ang"]), math.sin(v["ang"])
            u = dx * ca + dy * sa
            w_ = -dx * sa + dy * ca
            inside = (np.abs(u) <= v["L"] / 2) & (np.abs(w_) <= v["Wd"] / 2)
            img[y0:y1, x0:x1][inside] = np.asarray(v["color"], np.uint8)
        return img
