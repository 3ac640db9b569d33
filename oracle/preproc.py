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
 * Lab: the published CIE formulas OpenCV documents for 8-bit images (sRGB gamma, D65 white point,
   L*255/100, a+128, b+128), evaluated in f32; OpenCV's own 8-bit path uses fixed-point tables whose
   results can differ from this by +-1 level.
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


def lab_gamma(img_u8, lut):
    """RGB2LAB (8-bit convention) -> LUT on L -> LAB2RGB, f32 maths, channel 0 treated as R."""
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
