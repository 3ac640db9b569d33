// Per-pixel undistort + Lab-L gamma (device), shared by the stand-alone kernel (preproc.hip) and the fused staging of the
// horizontal resize pass (elementwise.hip).  Restates preprocess_img of /root/reference/dcnn/scripts/tests/visualize_uav.py:56-71
// (cv2.undistort, cvtColor RGB2LAB on a BGR frame, LUT on L, cvtColor LAB2RGB; same maths in aruco_detect.py:250-259):
//  * map: OpenCV initUndistortRectifyMap in f64 per destination pixel (R = I, new camera matrix = camera matrix, rational model
//    k1..k6, p1, p2, s1..s4), rounded half-to-even to 1/32 px;
//  * remap: INTER_LINEAR with the 15-bit integer weights of OpenCV's fixed-point path, border 0 -- integer arithmetic, bit-exact
//    with oracle/preproc.py;
//  * Lab: published CIE formulas in f32 (OpenCV's 8-bit tables are not reproducible here: parity unpinned).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct UndistortParams {
    double ir[9];           // inverse camera matrix
    double k[12];           // k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4
    double fx, fy, u0, v0;
    int H, W;
    int do_undistort, do_gamma;
};

#ifdef __HIPCC__
__device__ __forceinline__ float pp_srgb_to_lin(float c) { return c <= 0.04045f ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f); }
__device__ __forceinline__ float pp_lin_to_srgb(float c) { return c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f; }
__device__ __forceinline__ float pp_lab_f(float t) { return t > 0.008856f ? cbrtf(t) : 7.787f * t + (float)(16.0 / 116.0); }
__device__ __forceinline__ float pp_lab_finv(float t) { return t > 0.206893f ? t * t * t : (t - (float)(16.0 / 116.0)) / 7.787f; }
__device__ __forceinline__ int pp_sat8(float v) {
    const float r = rintf(v);
    return r < 0.f ? 0 : (r > 255.f ? 255 : (int)r);
}

// Source position of destination pixel (x, y) in 1/32 px: (sx, sy) integer part, (fx, fy) 5-bit fraction.
__device__ __forceinline__ void undistort_map_pixel(const UndistortParams& p, int x, int y, int& sx, int& sy, int& fx, int& fy) {
    const double j = (double)x, i = (double)y;
    const double _x = j * p.ir[0] + (i * p.ir[1] + p.ir[2]);
    const double _y = j * p.ir[3] + (i * p.ir[4] + p.ir[5]);
    const double _w = j * p.ir[6] + (i * p.ir[7] + p.ir[8]);
    const double w = 1.0 / _w;
    const double xx = _x * w, yy = _y * w;
    const double x2 = xx * xx, y2 = yy * yy;
    const double r2 = x2 + y2, _2xy = 2 * xx * yy;
    const double kr = (1 + ((p.k[4] * r2 + p.k[1]) * r2 + p.k[0]) * r2) / (1 + ((p.k[7] * r2 + p.k[6]) * r2 + p.k[5]) * r2);
    const double xd = xx * kr + p.k[2] * _2xy + p.k[3] * (r2 + 2 * x2) + p.k[8] * r2 + p.k[9] * r2 * r2;
    const double yd = yy * kr + p.k[2] * (r2 + 2 * y2) + p.k[3] * _2xy + p.k[10] * r2 + p.k[11] * r2 * r2;
    const double u = p.fx * xd + p.u0, v = p.fy * yd + p.v0;
    const long long iu = (long long)rint(u * 32.0), iv = (long long)rint(v * 32.0);
    // far-away sources (nothing of the frame under the 2x2 footprint) are parked at -2: they read zeros like any border tap
    const long long qx = iu >> 5, qy = iv >> 5;
    sx = (qx < -2 || qx > 100000) ? -2 : (int)qx;
    sy = (qy < -2 || qy > 100000) ? -2 : (int)qy;
    fx = (int)(iu & 31); fy = (int)(iv & 31);
}

// s: the frame [H][W][3] u8 (BGR); (x, y): destination pixel; out: its three channels.  map (optional): the camera's remap table
// [H][W] of (sx, sy << 0 | fractions) built once by apse_set_camera with undistort_map_pixel -- the map depends on the camera
// only, the f64 rational model per pixel was most of this function's time; lin (optional): srgb_to_lin of the 256 byte values,
// built on the device with the same powf.
__device__ __forceinline__ void undistort_gamma_pixel(const UndistortParams& p, const uint8_t* __restrict__ s,
                                                      const uint8_t* __restrict__ lut, int x, int y, int& c0, int& c1, int& c2,
                                                      const int2* __restrict__ map = nullptr, const float* __restrict__ lin = nullptr) {
    if (p.do_undistort) {
        int sx, sy, fx, fy;
        if (map) {
            const int2 mv = map[(size_t)y * p.W + x];
            sx = mv.x; sy = mv.y >> 10; fx = (mv.y >> 5) & 31; fy = mv.y & 31;
        } else {
            undistort_map_pixel(p, x, y, sx, sy, fx, fy);
        }
        const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
        int acc[3] = {1 << 14, 1 << 14, 1 << 14};
        const bool x0 = (unsigned)sx < (unsigned)p.W, x1 = (unsigned)(sx + 1) < (unsigned)p.W;
        const bool y0 = (unsigned)sy < (unsigned)p.H, y1 = (unsigned)(sy + 1) < (unsigned)p.H;
        if (y0 && x0) { const uint8_t* q = s + ((size_t)sy * p.W + sx) * 3; acc[0] += q[0] * w00; acc[1] += q[1] * w00; acc[2] += q[2] * w00; }
        if (y0 && x1) { const uint8_t* q = s + ((size_t)sy * p.W + sx + 1) * 3; acc[0] += q[0] * w01; acc[1] += q[1] * w01; acc[2] += q[2] * w01; }
        if (y1 && x0) { const uint8_t* q = s + ((size_t)(sy + 1) * p.W + sx) * 3; acc[0] += q[0] * w10; acc[1] += q[1] * w10; acc[2] += q[2] * w10; }
        if (y1 && x1) { const uint8_t* q = s + ((size_t)(sy + 1) * p.W + sx + 1) * 3; acc[0] += q[0] * w11; acc[1] += q[1] * w11; acc[2] += q[2] * w11; }
        c0 = acc[0] >> 15; c1 = acc[1] >> 15; c2 = acc[2] >> 15;
        c0 = c0 > 255 ? 255 : c0; c1 = c1 > 255 ? 255 : c1; c2 = c2 > 255 ? 255 : c2;
    } else {
        const uint8_t* q = s + ((size_t)y * p.W + x) * 3;
        c0 = q[0]; c1 = q[1]; c2 = q[2];
    }
    if (p.do_gamma) {
        // channel 0 plays "R" (the reference converts a BGR frame with COLOR_RGB2LAB)
        const float R = lin ? lin[c0] : pp_srgb_to_lin((float)c0 / 255.f), G = lin ? lin[c1] : pp_srgb_to_lin((float)c1 / 255.f),
                    B = lin ? lin[c2] : pp_srgb_to_lin((float)c2 / 255.f);
        const float X = (R * 0.412453f + G * 0.357580f + B * 0.180423f) / 0.950456f;
        const float Y = R * 0.212671f + G * 0.715160f + B * 0.072169f;
        const float Z = (R * 0.019334f + G * 0.119193f + B * 0.950227f) / 1.088754f;
        const float fX = pp_lab_f(X), fY = pp_lab_f(Y), fZ = pp_lab_f(Z);
        const float L = Y > 0.008856f ? 116.f * fY - 16.f : 903.3f * Y;
        int L8 = pp_sat8(L * (float)(255.0 / 100.0));
        const int a8 = pp_sat8(500.f * (fX - fY) + 128.f), b8 = pp_sat8(200.f * (fY - fZ) + 128.f);
        L8 = lut[L8];
        const float L2 = (float)L8 * (float)(100.0 / 255.0), a = (float)a8 - 128.f, bb = (float)b8 - 128.f;
        const float fy_ = (L2 + 16.f) / 116.f, fx_ = fy_ + a / 500.f, fz_ = fy_ - bb / 200.f;
        const float Y2 = L2 > 7.9996248f ? fy_ * fy_ * fy_ : L2 / 903.3f;
        const float X2 = pp_lab_finv(fx_) * 0.950456f, Z2 = pp_lab_finv(fz_) * 1.088754f;
        float r = X2 * 3.240479f + Y2 * -1.53715f + Z2 * -0.498535f;
        float g = X2 * -0.969256f + Y2 * 1.875991f + Z2 * 0.041556f;
        float bl = X2 * 0.055648f + Y2 * -0.204043f + Z2 * 1.057311f;
        r = fminf(fmaxf(r, 0.f), 1.f); g = fminf(fmaxf(g, 0.f), 1.f); bl = fminf(fmaxf(bl, 0.f), 1.f);
        c0 = pp_sat8(pp_lin_to_srgb(r) * 255.f); c1 = pp_sat8(pp_lin_to_srgb(g) * 255.f); c2 = pp_sat8(pp_lin_to_srgb(bl) * 255.f);
    }
}
#endif
