"""Oracle for the optional frame pre-processing (TEST INFRASTRUCTURE): undistort + Lab-L gamma.

Follows /root/reference/dcnn/scripts/tests/visualize_uav.py:56-71 ``preprocess_img`` (same maths in
/root/reference/aruco_detect.py:250-259,537-540,568):
    frame = cv2.undistort(frame, mtx, dist)            # rational model, 14 coefficients (data/cam_params.json)
    lab = cv2.cvtColor(frame, cv2.COLOR_RGB2LAB)       # applied to a BGR frame: channel 0 plays "R"
    lab[..., 0] = LUT[lab[..., 0]]                     # LUT[i] = uint8(clip((i/255)^2 * 255))
    frame = cv2.cvtColor(lab, cv2.COLOR_LAB2RGB)
OpenCV is not installed and its source is not under /root/reference: **parity unpinned**.
 * undistort: restated from OpenCV 4.2's initUndistortRectifyMap (double arithmetic, R = I, new camera
   matrix = mtx, tilt terms zero) + remap INTER_LINEAR on CV_16SC2 maps: coordinates rounded to 1/32 px
   (round-half-even), 15-bit integer bilinear weights (exact for 5-bit fractions), (sum + 2^14) >> 15,
   constant border 0.  Integer arithmetic after the map -> the HIP kernel matches this bit for bit.
 * Lab (round 3): integer and table-driven like the 8-bit path it stands in for (OpenCV 4.2 RGB2Lab_b /
   Lab2RGBinteger): sRGB -> linear through a 256-entry table in 1/2040 units (gamma shift 3), a Q12 XYZ matrix
   with rows divided by the D65 white point, a tabulated f() (cube root / linear toe) over that quantised value
   in Q15, L / a / b descaled from Q15 with L scale (116*255+50)/100 and shift (16*255*2^15+50)/100 -- that much
   is OpenCV's published structure -- then the way back: tabulated fy and Y per L byte, Q15 a / b offsets,
   integer f^-1, a Q12 inverse matrix with the white point folded in, and a tabulated inverse gamma over a
   12-bit linear value.  The published CIE constants (0.008856, 7.787, 903.3, sRGB 0.04045 / 12.92 / 2.4) are
   the ones OpenCV documents; OpenCV's exact table contents and the shifts of its inverse path are NOT
   reproducible here (cv2 absent, no fixture in the reference): parity with OpenCV itself stays unpinned.
   What this buys: the HIP kernel (csrc/preproc_pixel.h) performs the same integer steps on tables built by
   the same expressions, so HIP bytes == oracle bytes (tests/test_gpu_ops.py::test_undistort_gamma: 0 differ).
   The f32-formula form of rounds 1-2 is kept as ``lab_gamma_f32`` (a cross-check of the integer design:
   tests/test_oracle_golden.py bounds their difference).
"""
import numpy as np

INTER_BITS = 5
INTER_TAB = 1 << INTER_BITS
COEF_BITS = 15


def gamma_lut(gamma=2.0):
    lut = np.empty(256, np.uint8)
    for i in range(256):
        lut[i] = np.uint8(np.clip(pow(i / 255.0, gamma) * 255.0, 0, 255))       # truncation, as the uint8 store does
    return lut


def undistort_map(mtx, dist, height, width):
    """(ix, iy, fx, fy): integer source coordinates and 5-bit fractions per destination pixel."""
    A = np.asarray(mtx, np.float64)
    d = np.zeros(14, np.float64)
    dd = np.asarray(dist, np.float64).reshape(-1)
    d[: dd.size] = dd
    k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4 = d[:12]
    fx, fy, u0, v0 = A[0, 0], A[1, 1], A[0, 2], A[1, 2]
    ir = np.linalg.inv(A)
    j = np.arange(width, dtype=np.float64)[None, :]
    i = np.arange(height, dtype=np.float64)[:, None]
    _x = j * ir[0, 0] + (i * ir[0, 1] + ir[0, 2])
    _y = j * ir[1, 0] + (i * ir[1, 1] + ir[1, 2])
    _w = j * ir[2, 0] + (i * ir[2, 1] + ir[2, 2])
    w = 1.0 / _w
    x = _x * w
    y = _y * w
    x2, y2 = x * x, y * y
    r2 = x2 + y2
    _2xy = 2 * x * y
    kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
    xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2
    yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2
    u = fx * xd + u0
    v = fy * yd + v0
    iu = np.rint(u * INTER_TAB).astype(np.int64)
    iv = np.rint(v * INTER_TAB).astype(np.int64)
    return (iu >> INTER_BITS), (iv >> INTER_BITS), (iu & (INTER_TAB - 1)), (iv & (INTER_TAB - 1))


def remap_bilinear_u8(img, ix, iy, fx, fy):
    h, w, c = img.shape
    src = img.astype(np.int64)

    def px(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        v = src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
        return np.where(ok[..., None], v, 0)

    w00 = ((INTER_TAB - fx) * (INTER_TAB - fy) * 32)[..., None]
    w01 = (fx * (INTER_TAB - fy) * 32)[..., None]
    w10 = ((INTER_TAB - fx) * fy * 32)[..., None]
    w11 = (fx * fy * 32)[..., None]
    acc = px(iy, ix) * w00 + px(iy, ix + 1) * w01 + px(iy + 1, ix) * w10 + px(iy + 1, ix + 1) * w11
    return np.clip((acc + (1 << (COEF_BITS - 1))) >> COEF_BITS, 0, 255).astype(np.uint8)


def undistort(img, mtx, dist):
    ix, iy, fx, fy = undistort_map(mtx, dist, img.shape[0], img.shape[1])
    return remap_bilinear_u8(img, ix, iy, fx, fy)


_M = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]], np.float32)
_MI = np.array([[3.240479, -1.53715, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]], np.float32)
_XN, _ZN = np.float32(0.950456), np.float32(1.088754)


LAB_LIN_MAX, LAB_CBRT_N, LAB_INV_N = 2040, 3072, 4096
_M64 = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]], np.float64)
_MI64 = np.array([[3.240479, -1.53715, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]], np.float64)
_WH64 = np.array([0.950456, 1.0, 1.088754], np.float64)


def lab_tables(lut):
    """The integer tables of the Lab step (float64 expressions, rounded half-to-even like C rint)."""
    i = np.arange(256, dtype=np.float64)
    x = i / 255.0
    lin = np.where(x <= 0.04045, x / 12.92, np.power((x + 0.055) / 1.055, 2.4))
    L = i * 100.0 / 255.0
    fy = (L + 16.0) / 116.0
    j = np.arange(LAB_CBRT_N, dtype=np.float64) / 2040.0
    k = np.arange(LAB_INV_N + 1, dtype=np.float64) / 4096.0
    inv = np.rint(255.0 * np.where(k <= 0.0031308, 12.92 * k, 1.055 * np.power(k, 1.0 / 2.4) - 0.055))
    return dict(lin=np.rint(2040.0 * lin).astype(np.int64),
                cbrt=np.rint(32768.0 * np.where(j > 0.008856, np.cbrt(j), 7.787 * j + 16.0 / 116.0)).astype(np.int64),
                c=np.rint(4096.0 * _M64 / _WH64[:, None]).astype(np.int64),
                lut=np.asarray(lut, np.int64),
                fy=np.rint(32768.0 * fy).astype(np.int64),
                y=np.rint(32768.0 * np.where(L > 7.9996248, fy * fy * fy, L / 903.3)).astype(np.int64),
                at=np.rint(32768.0 * (i - 128.0) / 500.0).astype(np.int64),
                bt=np.rint(32768.0 * (i - 128.0) / 200.0).astype(np.int64),
                ci=np.rint(4096.0 * _MI64 * _WH64[None, :]).astype(np.int64),
                inv=np.clip(inv, 0, 255).astype(np.int64))


def _finv_q15(t):
    """f^-1 on Q15 integers (may be negative): t > 0.206893 ? t^3 : (t - 16/116) / 7.787; >> is floor division."""
    return np.where(t >= 6780, (t * t * t + (1 << 29)) >> 30, ((t - 4520) * 269314 + (1 << 20)) >> 21)


def lab_gamma(img_u8, lut):
    """RGB2LAB (8-bit) -> LUT on L -> LAB2RGB in integer arithmetic on ``lab_tables``; channel 0 treated as R
    (the reference converts a BGR frame with COLOR_RGB2LAB, visualize_uav.py:63)."""
    T = lab_tables(lut)
    src = img_u8.astype(np.int64)
    R, G, B = T["lin"][src[..., 0]], T["lin"][src[..., 1]], T["lin"][src[..., 2]]
    c = T["c"]
    fX = T["cbrt"][(R * c[0, 0] + G * c[0, 1] + B * c[0, 2] + 2048) >> 12]
    fY = T["cbrt"][(R * c[1, 0] + G * c[1, 1] + B * c[1, 2] + 2048) >> 12]
    fZ = T["cbrt"][(R * c[2, 0] + G * c[2, 1] + B * c[2, 2] + 2048) >> 12]
    L8 = np.clip((296 * fY - 1336934 + 16384) >> 15, 0, 255)
    a8 = np.clip((500 * (fX - fY) + (128 << 15) + 16384) >> 15, 0, 255)
    b8 = np.clip((200 * (fY - fZ) + (128 << 15) + 16384) >> 15, 0, 255)
    L2 = T["lut"][L8]
    fy = T["fy"][L2]
    X = _finv_q15(fy + T["at"][a8])
    Y = T["y"][L2]
    Z = _finv_q15(fy - T["bt"][b8])
    ci = T["ci"]
    out = np.empty(img_u8.shape, np.uint8)
    for ch in range(3):
        v = (X * ci[ch, 0] + Y * ci[ch, 1] + Z * ci[ch, 2] + 16384) >> 15
        out[..., ch] = T["inv"][np.clip(v, 0, LAB_INV_N)]
    return out


def lab_gamma_f32(img_u8, lut):
    """The f32 CIE-formula form (rounds 1-2), kept as a cross-check of the integer design above."""
    f = np.float32
    c = img_u8.astype(np.float32) / f(255)
    lin = np.where(c <= f(0.04045), c / f(12.92), np.power((c + f(0.055)) / f(1.055), f(2.4))).astype(np.float32)
    X = (lin @ _M[0]) / _XN
    Y = lin @ _M[1]
    Z = (lin @ _M[2]) / _ZN

    def fn(t):
        return np.where(t > f(0.008856), np.cbrt(t), f(7.787) * t + f(16.0 / 116.0)).astype(np.float32)

    fX, fY, fZ = fn(X), fn(Y), fn(Z)
    L = np.where(Y > f(0.008856), f(116) * fY - f(16), f(903.3) * Y).astype(np.float32)
    a = f(500) * (fX - fY)
    b = f(200) * (fY - fZ)
    L8 = np.clip(np.rint(L * f(255.0 / 100.0)), 0, 255).astype(np.uint8)
    a8 = np.clip(np.rint(a + f(128)), 0, 255).astype(np.uint8)
    b8 = np.clip(np.rint(b + f(128)), 0, 255).astype(np.uint8)
    L8 = lut[L8]
    L = L8.astype(np.float32) * f(100.0 / 255.0)
    a = a8.astype(np.float32) - f(128)
    b = b8.astype(np.float32) - f(128)
    fy_ = (L + f(16)) / f(116)
    fx_ = fy_ + a / f(500)
    fz_ = fy_ - b / f(200)
    Y = np.where(L > f(7.9996248), fy_ * fy_ * fy_, L / f(903.3)).astype(np.float32)

    def inv(t):
        return np.where(t > f(0.206893), t * t * t, (t - f(16.0 / 116.0)) / f(7.787)).astype(np.float32)

    X = inv(fx_) * _XN
    Z = inv(fz_) * _ZN
    xyz = np.stack([X, Y, Z], axis=-1)
    lin = np.stack([xyz @ _MI[0], xyz @ _MI[1], xyz @ _MI[2]], axis=-1).astype(np.float32)
    lin = np.clip(lin, 0, 1)
    srgb = np.where(lin <= f(0.0031308), f(12.92) * lin, f(1.055) * np.power(lin, f(1.0 / 2.4)) - f(0.055))
    return np.clip(np.rint(srgb * f(255)), 0, 255).astype(np.uint8)


def preprocess_img(frame, mtx, dist, gamma=2.0):
    return lab_gamma(undistort(frame, mtx, dist), gamma_lut(gamma))
