// Per-pixel undistort + Lab-L gamma (device), shared by the stand-alone kernel (preproc.hip) and the fused staging of the
// horizontal resize pass (elementwise.hip).  Restates preprocess_img of /root/reference/dcnn/scripts/tests/visualize_uav.py:56-71
// (cv2.undistort, cvtColor RGB2LAB on a BGR frame, LUT on L, cvtColor LAB2RGB; same maths in aruco_detect.py:250-259):
//  * map: OpenCV initUndistortRectifyMap in f64 per destination pixel (R = I, new camera matrix = camera matrix, rational model
//    k1..k6, p1, p2, s1..s4), rounded half-to-even to 1/32 px;
//  * remap: INTER_LINEAR with the 15-bit integer weights of OpenCV's fixed-point path, border 0 -- integer arithmetic, bit-exact
//    with oracle/preproc.py;
//  * Lab (round 3): integer and table-driven, in the manner of OpenCV's own 8-bit path (RGB2Lab_b / Lab2RGBinteger: a 256-entry
//    sRGB -> linear table, a Q12 XYZ matrix, a tabulated cube root, Q15 L / a / b, and on the way back a tabulated inverse gamma
//    over a 12-bit linear value).  Every step is integer arithmetic on tables built once on the host (lab_tables_build below),
//    so the bytes equal oracle/preproc.py's, which builds the same tables with numpy (tests/test_host_logic.py compares the
//    tables entry by entry; tests/test_gpu_ops.py asserts 0 differing bytes at 4K).  OpenCV's exact table constants are not
//    reproducible here (cv2 absent): parity with OpenCV itself stays unpinned.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define LAB_GS 3                          // gamma shift: linear values carry 255 << 3 = 2040 levels
#define LAB_LIN_MAX 2040
#define LAB_CBRT_N 3072                   // 256 * 3 / 2 << LAB_GS entries, index = white-normalised X / Y / Z in 1 / 2040
#define LAB_INV_N 4096                    // inverse gamma over a 12-bit linear value (+ 1 entry for 1.0)
struct LabTables {
    uint16_t lin[256];                    // rint(2040 * sRGB EOTF(i / 255))
    uint16_t cbrt[LAB_CBRT_N];            // rint(32768 * f(i / 2040)), f(t) = t > 0.008856 ? cbrt(t) : 7.787 t + 16 / 116
    int32_t c[9];                         // rint(4096 * M[r][c] / white[r]): RGB -> XYZ, rows normalised by the D65 white point
    uint8_t lut[256];                     // the reference's LUT on L: uint8(clip((i / 255)^gamma * 255))
    uint16_t fy[256];                     // rint(32768 * (L + 16) / 116), L = i * 100 / 255
    int32_t y[256];                       // rint(32768 * (L > 7.9996248 ? fy^3 : L / 903.3))
    int32_t at[256], bt[256];             // rint(32768 * (i - 128) / 500), rint(32768 * (i - 128) / 200)
    int32_t ci[9];                        // rint(4096 * MI[r][c] * white[c]): white-normalised XYZ -> linear RGB
    uint8_t inv[LAB_INV_N + 1];           // sat8(rint(255 * sRGB OETF(i / 4096)))
    uint8_t pad_[3];
};

// Host builder (double precision, libm).  Same expressions, evaluated in the same order, as oracle/preproc.py lab_tables().
static inline void lab_tables_build(LabTables* t, const uint8_t* lut256) {
    static const double M[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
    static const double MI[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311};
    static const double WH[3] = {0.950456, 1.0, 1.088754};
    for (int i = 0; i < 256; ++i) {
        const double x = (double)i / 255.0;
        const double l = x <= 0.04045 ? x / 12.92 : pow((x + 0.055) / 1.055, 2.4);
        t->lin[i] = (uint16_t)rint(2040.0 * l);
        const double L = (double)i * 100.0 / 255.0;
        const double fy = (L + 16.0) / 116.0;
        t->fy[i] = (uint16_t)rint(32768.0 * fy);
        t->y[i] = (int32_t)rint(32768.0 * (L > 7.9996248 ? fy * fy * fy : L / 903.3));
        t->at[i] = (int32_t)rint(32768.0 * (double)(i - 128) / 500.0);
        t->bt[i] = (int32_t)rint(32768.0 * (double)(i - 128) / 200.0);
        t->lut[i] = lut256[i];
    }
    for (int i = 0; i < LAB_CBRT_N; ++i) {
        const double x = (double)i / 2040.0;
        t->cbrt[i] = (uint16_t)rint(32768.0 * (x > 0.008856 ? cbrt(x) : 7.787 * x + 16.0 / 116.0));
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            t->c[r * 3 + c] = (int32_t)rint(4096.0 * M[r * 3 + c] / WH[r]);
            t->ci[r * 3 + c] = (int32_t)rint(4096.0 * MI[r * 3 + c] * WH[c]);
        }
    for (int i = 0; i <= LAB_INV_N; ++i) {
        const double x = (double)i / 4096.0;
        const double v = rint(255.0 * (x <= 0.0031308 ? 12.92 * x : 1.055 * pow(x, 1.0 / 2.4) - 0.055));
        t->inv[i] = (uint8_t)(v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v));
    }
    t->pad_[0] = t->pad_[1] = t->pad_[2] = 0;
}

struct UndistortParams {
    double ir[9];           // inverse camera matrix
    double k[12];           // k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4
    double fx, fy, u0, v0;
    int H, W;
    int do_undistort, do_gamma;
};

#ifdef __HIPCC__
__device__ __forceinline__ int pp_sat8i(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
// inverse of f() on a Q15 argument (may be negative): t > 0.206893 ? t^3 : (t - 16 / 116) / 7.787, Q15 result, floor rounding
// of the shifted products like numpy's >> on int64.  Ranges (from the tables: fy <= 32768, |at| <= 8323, |bt| <= 20972): t in
// [-20972, 53740], so t^2 < 2^32 and the 32 x 32 -> 64-bit multiply-adds below are exact; the result is in [-3274, 144500].
// (Round 4: the same integers as the long long form of round 3 -- which made every product a 64 x 64-bit multiply, ~270 VALU
// instructions per pixel and the fused resize pass VALU-bound -- checked on all 2^24 colours by tests/test_gpu_ops.py.)
__device__ __forceinline__ int pp_lab_finv_q15(int t) {
    if (t >= 6780) {
        const unsigned t2 = (unsigned)t * (unsigned)t;
        return (int)(((unsigned long long)t2 * (unsigned)t + (1ull << 29)) >> 30);
    }
    return (int)(((long long)(t - 4520) * 269314 + (1ll << 20)) >> 21);          // 2^21 / 7.787 = 269 314.5; 16 / 116 in Q15 = 4519.7
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

// Lab-L gamma of one pixel (channel 0 plays "R": the reference converts a BGR frame with COLOR_RGB2LAB), in place.
__device__ __forceinline__ void lab_gamma_pixel(const LabTables* __restrict__ lab, int& c0, int& c1, int& c2) {
    const int R = lab->lin[c0], G = lab->lin[c1], B = lab->lin[c2];
    const int fX = lab->cbrt[(R * lab->c[0] + G * lab->c[1] + B * lab->c[2] + 2048) >> 12];
    const int fY = lab->cbrt[(R * lab->c[3] + G * lab->c[4] + B * lab->c[5] + 2048) >> 12];
    const int fZ = lab->cbrt[(R * lab->c[6] + G * lab->c[7] + B * lab->c[8] + 2048) >> 12];
    const int L8 = pp_sat8i((296 * fY - 1336934 + 16384) >> 15);           // (116 fY - 16) * 255 / 100 in Q15: 296 = (116 * 255 + 50) / 100, 1336934 = (16 * 255 * 2^15 + 50) / 100
    const int a8 = pp_sat8i((500 * (fX - fY) + (128 << 15) + 16384) >> 15);
    const int b8 = pp_sat8i((200 * (fY - fZ) + (128 << 15) + 16384) >> 15);
    const int L2 = lab->lut[L8];
    const int fy = lab->fy[L2];
    const int X = pp_lab_finv_q15(fy + lab->at[a8]), Y = lab->y[L2], Z = pp_lab_finv_q15(fy - lab->bt[b8]);
    // |X|, |Z| <= 144500, Y <= 32768, |ci| <= 12615: each product fits 31 bits, the sum of three needs 64 (one v_mad_i64_i32 each)
    int r = (int)(((long long)X * lab->ci[0] + (long long)Y * lab->ci[1] + (long long)Z * lab->ci[2] + 16384) >> 15);
    int g = (int)(((long long)X * lab->ci[3] + (long long)Y * lab->ci[4] + (long long)Z * lab->ci[5] + 16384) >> 15);
    int bl = (int)(((long long)X * lab->ci[6] + (long long)Y * lab->ci[7] + (long long)Z * lab->ci[8] + 16384) >> 15);
    r = r < 0 ? 0 : (r > LAB_INV_N ? LAB_INV_N : r);
    g = g < 0 ? 0 : (g > LAB_INV_N ? LAB_INV_N : g);
    bl = bl < 0 ? 0 : (bl > LAB_INV_N ? LAB_INV_N : bl);
    c0 = lab->inv[r]; c1 = lab->inv[g]; c2 = lab->inv[bl];
}

// s: the frame [H][W][3] u8 (BGR); (x, y): destination pixel; out: its three channels; lab: the Lab tables (device copy of
// LabTables).  The f64 rational model is evaluated per pixel: the form of the stand-alone operator, and of a context whose camera
// does not fit the compact remap table below.
__device__ __forceinline__ void undistort_gamma_pixel(const UndistortParams& p, const uint8_t* __restrict__ s,
                                                      const LabTables* __restrict__ lab, int x, int y, int& c0, int& c1, int& c2) {
    if (p.do_undistort) {
        int sx, sy, fx, fy;
        undistort_map_pixel(p, x, y, sx, sy, fx, fy);
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
    if (p.do_gamma) lab_gamma_pixel(lab, c0, c1, c2);
}

// ---- the fused path of the horizontal resize pass (round 4): compact remap table + wide gathers, four pixels in flight per thread.
// Compact table entry (4 bytes instead of 8: the table was 66 of the 117 MB the fused pass fetched per 4K frame): the source
// position RELATIVE to the destination pixel in 1/32 px, dx = iu - 32 x in the low half, dy = iv - 32 y in the high half, int16
// each (|displacement| < 1024 px); PP_MAP_FAR marks a pixel whose 2x2 footprint lies outside the frame (reads zeros like any
// border tap) -- and any pixel whose displacement does not fit, which the builder counts: a camera with such pixels INSIDE the
// frame keeps the per-pixel f64 model (no table), never a wrong tap.
#define PP_MAP_FAR 0x80008000u
__device__ __forceinline__ uint32_t undistort_map_compact(const UndistortParams& p, int x, int y, int* overflow) {
    int sx, sy, fx, fy;
    undistort_map_pixel(p, x, y, sx, sy, fx, fy);
    const bool touches = sx >= -1 && sx < p.W && sy >= -1 && sy < p.H;          // some tap of the 2x2 footprint is a frame pixel
    if (!touches) return PP_MAP_FAR;
    const int dx = ((sx - x) << 5) | fx, dy = ((sy - y) << 5) | fy;
    if (dx <= -32768 || dx > 32767 || dy <= -32768 || dy > 32767) { if (overflow) atomicAdd(overflow, 1); return PP_MAP_FAR; }
    return (uint32_t)(uint16_t)(int16_t)dx | ((uint32_t)(uint16_t)(int16_t)dy << 16);
}
// Six consecutive bytes (two BGR pixels) at byte offset `off` of the frames buffer through a range-checked descriptor: three
// aligned dwords, funnel-shifted (a byte gather is 12 load instructions per pixel; this is 2).  Bytes outside the buffer read 0.
typedef unsigned int pp_u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ pp_u32x3 pp_load12(__amdgpu_buffer_rsrc_t rs, int off) {
    return __builtin_amdgcn_raw_buffer_load_b96(rs, off & ~3, 0, 0);
}
__device__ __forceinline__ unsigned long long pp_six(pp_u32x3 d, int off) {
    const int sh = (off & 3) * 8;
    const unsigned long long lo = ((unsigned long long)d.y << 32) | d.x;
    return sh ? ((lo >> sh) | ((unsigned long long)d.z << (64 - sh))) : lo;
}
// One pixel from its table entry and its two gathered 6-byte rows (row sy: q0, row sy + 1: q1); same integer arithmetic and tap
// order as undistort_gamma_pixel.
__device__ __forceinline__ void undistort_from_taps(const UndistortParams& p, uint32_t m, int x, int y, unsigned long long q0,
                                                    unsigned long long q1, int& c0, int& c1, int& c2) {
    int sx = -2, sy = -2, fx = 0, fy = 0;
    if (m != PP_MAP_FAR) {
        const int dx = (int)(int16_t)(m & 0xffffu), dy = (int)(int16_t)(m >> 16);
        sx = x + (dx >> 5); sy = y + (dy >> 5); fx = dx & 31; fy = dy & 31;
    }
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    int acc[3] = {1 << 14, 1 << 14, 1 << 14};
    const bool x0 = (unsigned)sx < (unsigned)p.W, x1 = (unsigned)(sx + 1) < (unsigned)p.W;
    const bool y0 = (unsigned)sy < (unsigned)p.H, y1 = (unsigned)(sy + 1) < (unsigned)p.H;
#define PP_B(q, k) ((int)(((q) >> (8 * (k))) & 0xffull))
    if (y0 && x0) { acc[0] += PP_B(q0, 0) * w00; acc[1] += PP_B(q0, 1) * w00; acc[2] += PP_B(q0, 2) * w00; }
    if (y0 && x1) { acc[0] += PP_B(q0, 3) * w01; acc[1] += PP_B(q0, 4) * w01; acc[2] += PP_B(q0, 5) * w01; }
    if (y1 && x0) { acc[0] += PP_B(q1, 0) * w10; acc[1] += PP_B(q1, 1) * w10; acc[2] += PP_B(q1, 2) * w10; }
    if (y1 && x1) { acc[0] += PP_B(q1, 3) * w11; acc[1] += PP_B(q1, 4) * w11; acc[2] += PP_B(q1, 5) * w11; }
#undef PP_B
    c0 = acc[0] >> 15; c1 = acc[1] >> 15; c2 = acc[2] >> 15;
    c0 = c0 > 255 ? 255 : c0; c1 = c1 > 255 ? 255 : c1; c2 = c2 > 255 ? 255 : c2;
}
#endif
