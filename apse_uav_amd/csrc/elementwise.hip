// HBM-bound elementwise / resampling kernels of the dcnn hot path (gfx950).
//  * PIL-style antialiased bilinear resize (two integer passes, bit-exact with Pillow's
//    Resample.c 8bpc path) fused with mean subtraction and /32 zero padding:
//    replaces ResizeShortestEdge.apply_image + GeneralizedRCNN.preprocess_image
//    (/root/reference/dcnn/engines/track_predictor.py:48-49, dcnn/networks/track_rcnn.py:35).
//  * stem max-pool 3x3/2 and FPN p6 subsample (detectron2 ResNet stem / LastLevelMaxPool,
//    reached from track_rcnn.py:42).
#include "apse_common.h"
#include "preproc_pixel.h"
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#define PIL_PRECISION_BITS 22

__device__ __forceinline__ int clip8(int v) {
    v >>= PIL_PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// Horizontal pass: one block per input row.  The row (W*3 bytes, BGR interleaved) is staged in
// LDS with coalesced 4-byte loads; each thread then produces output samples (ox, c).
// bounds: [OW][2] = (xmin, count); coef: [OW][ksize] int32 (Pillow normalize_coeffs_8bpc).
// FUSED: the row is not copied but COMPUTED while staging -- undistort gather + Lab gamma of the raw frame (preproc_pixel.h,
// the reference's preprocess_img) -- so a pre-processed 4K frame is never written to / re-read from HBM (2 x 24.9 MB per frame
// and one launch less than undistort_gamma -> pil_resize_h).  Same per-pixel function, same bytes as the two-kernel form.
template <bool FUSED>
__device__ __forceinline__ void pil_stage_row(uint8_t* row, const uint8_t* __restrict__ src, int y, int b, int W, size_t src_img_stride,
                                              const UndistortParams& cam, const LabTables* __restrict__ lut,
                                              const uint32_t* __restrict__ cam_map, unsigned src_bytes) {
    const uint8_t* srow = src + (size_t)b * src_img_stride + (size_t)y * W * 3;
    const int nbytes = W * 3;
    if constexpr (FUSED) {
        // the Lab step is ~15 table look-ups per pixel (14.7 KB of tables): gathers from LDS, not from global memory -- the block
        // copies the tables once (round 3: the fused horizontal pass was 117 us per 4K frame, most of it those gathers)
        __shared__ LabTables lab_s;
        if (cam.do_gamma) {
            const uint32_t* g = reinterpret_cast<const uint32_t*>(lut);
            uint32_t* l = reinterpret_cast<uint32_t*>(&lab_s);
            for (int i = threadIdx.x; i < (int)(sizeof(LabTables) / 4); i += blockDim.x) l[i] = g[i];
            __syncthreads();
        }
        const uint8_t* frame = src + (size_t)b * src_img_stride;
        if (cam.do_undistort && cam_map) {
            // Round 4.  The pass was a chain of dependent round trips per pixel (table entry -> four byte gathers of three loads each
            // -> LDS look-ups), 15 pixels per thread one after the other: ~60 us per block, latency- not bandwidth-bound, and of the
            // 117 MB it fetched per 4K frame 66 MB were the 8-byte table entries.  Now: 4-byte entries (preproc_pixel.h), the two
            // 6-byte rows of a pixel's 2x2 footprint as two range-checked 12-byte loads, and FOUR pixels per thread in flight --
            // 4 table loads, then 8 gathers, then the arithmetic.  Same integers as undistort_gamma_pixel (tests: 0 differing bytes).
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, (int)src_bytes, 0x00020000);
            const int fo = (int)((size_t)b * src_img_stride);
            const uint32_t* mrow = cam_map + (size_t)y * W;
            for (int x0 = threadIdx.x; x0 < W; x0 += 4 * blockDim.x) {
                uint32_t m[4];
                int off0[4], off1[4], shl[4];
                pp_u32x3 d0[4], d1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int x = x0 + u * blockDim.x; m[u] = x < W ? mrow[x] : PP_MAP_FAR; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int x = x0 + u * blockDim.x;
                    off0[u] = off1[u] = 0; shl[u] = 0;
                    if (m[u] != PP_MAP_FAR) {
                        const int dx = (int)(int16_t)(m[u] & 0xffffu), dy = (int)(int16_t)(m[u] >> 16);
                        const int sx = x + (dx >> 5), sy = y + (dy >> 5);
                        const int sxc = sx < 0 ? 0 : sx;          // sx = -1: the left tap is outside; load from pixel 0 and shift it into the right tap's place
                        shl[u] = sx < 0 ? 24 : 0;
                        off0[u] = fo + (sy * W + sxc) * 3;
                        off1[u] = off0[u] + W * 3;
                    }
                    d0[u] = pp_load12(rs, off0[u]);
                    d1[u] = pp_load12(rs, off1[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int x = x0 + u * blockDim.x;
                    if (x >= W) break;
                    int c0, c1, c2;
                    undistort_from_taps(cam, m[u], x, y, pp_six(d0[u], off0[u]) << shl[u], pp_six(d1[u], off1[u]) << shl[u], c0, c1, c2);
                    if (cam.do_gamma) lab_gamma_pixel(&lab_s, c0, c1, c2);
                    row[x * 3 + 0] = (uint8_t)c0; row[x * 3 + 1] = (uint8_t)c1; row[x * 3 + 2] = (uint8_t)c2;
                }
            }
        } else {
            for (int x = threadIdx.x; x < W; x += blockDim.x) {
                int c0, c1, c2;
                undistort_gamma_pixel(cam, frame, &lab_s, x, y, c0, c1, c2);
                row[x * 3 + 0] = (uint8_t)c0; row[x * 3 + 1] = (uint8_t)c1; row[x * 3 + 2] = (uint8_t)c2;
            }
        }
    } else
    // W*3 is a multiple of 4 for every supported width (W % 4 == 0); rows start 4-byte aligned.
    if ((nbytes & 15) == 0) {              // 16-byte rows (W % 16 == 0, e.g. 3840): three 16-byte loads per thread, all in flight
        const uint4* s16 = reinterpret_cast<const uint4*>(srow);
        uint4* r16 = reinterpret_cast<uint4*>(row);
        for (int i = threadIdx.x; i < (nbytes >> 4); i += blockDim.x) r16[i] = s16[i];
    } else {
        const uint32_t* s4 = reinterpret_cast<const uint32_t*>(srow);
        uint32_t* r4 = reinterpret_cast<uint32_t*>(row);
        for (int i = threadIdx.x; i < (nbytes >> 2); i += blockDim.x) r4[i] = s4[i];
    }
}

// Any filter length: each thread produces output samples (ox, c) one after the other.
template <bool FUSED>
__global__ __launch_bounds__(256) void pil_resize_h(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp,
                                                    const int* __restrict__ bounds, const int* __restrict__ coef,
                                                    int H, int W, int OW, int ksize, size_t src_img_stride,
                                                    size_t tmp_img_stride, const UndistortParams cam, const LabTables* __restrict__ lut,
                                                    const uint32_t* __restrict__ cam_map, int tp, unsigned src_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint8_t* row = reinterpret_cast<uint8_t*>(smem);
    const int y = blockIdx.x, b = blockIdx.y;
    pil_stage_row<FUSED>(row, src, y, b, W, src_img_stride, cam, lut, cam_map, src_bytes);
    __syncthreads();
    uint8_t* orow = tmp + (size_t)b * tmp_img_stride + (size_t)y * tp;
    for (int o = threadIdx.x; o < OW * 3; o += blockDim.x) {
        const int ox = o / 3, c = o - ox * 3;
        const int xmin = bounds[2 * ox], cnt = bounds[2 * ox + 1];
        const int* k = coef + ox * ksize;
        int ss = 1 << (PIL_PRECISION_BITS - 1);
        for (int j = 0; j < cnt; ++j) ss += (int)row[(xmin + j) * 3 + c] * k[j];
        orow[o] = (uint8_t)clip8(ss);
    }
}

// Filters of at most 8 taps (every downscale up to 3.5x, e.g. 3840 -> 1344): the form above is bound by its chain of dependent
// loads (bounds -> filter taps -> LDS, once per output sample: ~16 samples x 2 round trips per thread, 43 us per 4K frame
// for 25 MB).  Here a thread owns output pixel ox = tid + 256 i with its three channels, and the bounds and the 8 tap
// slots of TWO pixels are requested together (their addresses depend on ox only; slots past the pixel's count read as 0),
// so a block makes ~3 round trips instead of ~32.  The output row is collected in LDS and leaves in 16-byte stores.
// Same integer arithmetic in the same order: bit-identical output.
template <bool FUSED>
__global__ __launch_bounds__(256) void pil_resize_h8(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp,
                                                     const int2* __restrict__ bounds, const int* __restrict__ coef,
                                                     int H, int W, int OW, int ksize, size_t src_img_stride,
                                                     size_t tmp_img_stride, const UndistortParams cam, const LabTables* __restrict__ lut,
                                                     const uint32_t* __restrict__ cam_map, const int* __restrict__ coefT, int tp,
                                                     unsigned src_bytes, int nb) {
    // coefT (optional): the taps tap-major, [8][OW], zero past a pixel's count.  A wave's load of tap j is then 256 consecutive
    // bytes; pixel-major (coef[ox * ksize + j]) it touches 28 cache lines, and with every block (= source row) re-reading the
    // whole table that was what paced this pass (TA cycles: 144 us per 8 frames at 1.4 TB/s).
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint8_t* row = reinterpret_cast<uint8_t*>(smem);                      // W*3 bytes (+ 32: tap slots past the row end)
    int y = blockIdx.x, b = blockIdx.y;
    if (nb > 0) {
        // fused form, 1-D grid: the nb frames of a batch share the camera's remap table.  Blocks are dealt round-robin over the 8
        // XCDs, so the blocks of one source row are given ids that are equal mod 8 and adjacent in time: a row of the table
        // (15 KB) is fetched into ONE XCD's L2 once per batch instead of once per frame.
        const int id = blockIdx.x, grp = id / (8 * nb), r = id - grp * (8 * nb);
        y = grp * 8 + (r & 7); b = r >> 3;
        if (y >= H) return;
    }
    uint8_t* gdst = tmp + (size_t)b * tmp_img_stride + (size_t)y * tp;          // tp: row pitch of the intermediate image (>= OW * 3)
    // the output row sits in LDS at the same offset mod 16 as its destination, so the 16-byte body of the copy-out is
    // aligned on both sides whatever OW is (1333 * 3 bytes per row: rows start at every alignment)
    const int mis = (int)(reinterpret_cast<uintptr_t>(gdst) & 15);
    uint8_t* orow_l = row + (((size_t)W * 3 + 32 + 15) & ~(size_t)15) + mis;    // OW*3 bytes
    pil_stage_row<FUSED>(row, src, y, b, W, src_img_stride, cam, lut, cam_map, src_bytes);
    if (threadIdx.x < 32) row[W * 3 + threadIdx.x] = 0;
    __syncthreads();
    // (requesting the taps of all of a thread's pixels before the row is staged -- one round trip instead of four -- was tried:
    // 133 against 123 us per 8 frames, the 60 extra registers cost more occupancy than the round trips cost time)
    for (int ox0 = threadIdx.x; ox0 < OW; ox0 += 512) {
        int2 bd[2];
        int k[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ox = ox0 + 256 * u;
            const bool in = ox < OW;
            bd[u] = in ? bounds[ox] : int2{0, 0};
#pragma unroll
            for (int j = 0; j < 8; ++j) k[u][j] = !in ? 0 : (coefT ? coefT[j * OW + ox] : (j < ksize ? coef[ox * ksize + j] : 0));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ox = ox0 + 256 * u;
            if (ox >= OW) break;
            const uint8_t* px = row + bd[u].x * 3;
            int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = j < bd[u].y ? k[u][j] : 0;
                s0 += (int)px[j * 3 + 0] * kk;
                s1 += (int)px[j * 3 + 1] * kk;
                s2 += (int)px[j * 3 + 2] * kk;
            }
            orow_l[ox * 3 + 0] = (uint8_t)clip8(s0);
            orow_l[ox * 3 + 1] = (uint8_t)clip8(s1);
            orow_l[ox * 3 + 2] = (uint8_t)clip8(s2);
        }
    }
    __syncthreads();
    const int nbytes = OW * 3;
    int head = (16 - mis) & 15;
    head = head < nbytes ? head : nbytes;
    const int n16 = (nbytes - head) >> 4;
    if ((int)threadIdx.x < head) gdst[threadIdx.x] = orow_l[threadIdx.x];
    uint4* g16 = reinterpret_cast<uint4*>(gdst + head);
    const uint4* l16 = reinterpret_cast<const uint4*>(orow_l + head);
    for (int i = threadIdx.x; i < n16; i += blockDim.x) g16[i] = l16[i];
    for (int i = head + n16 * 16 + threadIdx.x; i < nbytes; i += blockDim.x) gdst[i] = orow_l[i];
}

// Network input layouts (both zeroed once at allocation; only the OH x OW interior and the 3 real channels are ever written):
//   out_st = 0: NHWC4 f32 [B][PH][PW][4] (BGR0);
//   out_st = 1 / 2 (16-bit storage modes): space-to-depth(2) NHWC16 bf16 / f16 [B][PH/2][PW/2][16], channel = (2 (y & 1) + (x & 1)) 3 + c:
//     the 7x7 / stride-2 stem becomes a 4x4 / stride-1 convolution over 12 (+4 zero) channels whose filter-row run is exactly one
//     64-element k-step of the 16-bit MFMA kernels (detector.hip, stem).  The value is rounded once, here, to the operand type
//     (the f32-input stem kernel rounded the same value while staging).
__device__ __forceinline__ void input_store(void* out, int out_st, int b, int oy, int ox, int c, int PH, int PW, float v) {
    if (out_st == 0) reinterpret_cast<float*>(out)[(((size_t)b * PH + oy) * PW + ox) * 4 + c] = v;
    else apse_st1(out, (((size_t)b * (PH >> 1) + (oy >> 1)) * (PW >> 1) + (ox >> 1)) * 16 + ((oy & 1) * 2 + (ox & 1)) * 3 + c, v, out_st);
}

// Vertical pass + normalise + pad.
#ifndef PIL_VROWS
#define PIL_VROWS 4
#endif
__global__ __launch_bounds__(256) void pil_resize_v_norm(const uint8_t* __restrict__ tmp, void* __restrict__ out, int out_st,
                                                         const int* __restrict__ bounds, const int* __restrict__ coef,
                                                         int OH, int OW, int ksize, int PH, int PW,
                                                         float m0, float m1, float m2, size_t tmp_img_stride, int tp,
                                                         uint8_t* __restrict__ resized_u8) {
    // a thread produces its sample column for PIL_VROWS consecutive output rows: their taps (<= 8 each, independent byte loads) are
    // all requested together -- one block per output row made 12 000 blocks per 4K frame of two dependent round trips each
    // (90 us per 8 frames, launch- and latency-bound)
    const int b = blockIdx.z;
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= OW * 3) return;
    const int ox = o / 3, c = o - ox * 3;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
    const uint8_t* tb = tmp + (size_t)b * tmp_img_stride + o;
    int vv[PIL_VROWS];
#pragma unroll
    for (int r = 0; r < PIL_VROWS; ++r) {
        const int oy = blockIdx.y * PIL_VROWS + r;
        int ss = 1 << (PIL_PRECISION_BITS - 1);
        if (oy < OH) {
            const int ymin = bounds[2 * oy], cnt = bounds[2 * oy + 1];
            const int* k = coef + oy * ksize;
            const uint8_t* t = tb + (size_t)ymin * tp;
            if (ksize <= 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const bool on = j < cnt; ss += (int)t[on ? (size_t)j * tp : 0] * (on ? k[on ? j : 0] : 0); }
            } else {
                for (int j = 0; j < cnt; ++j) ss += (int)t[(size_t)j * tp] * k[j];
            }
        }
        vv[r] = clip8(ss);
    }
#pragma unroll
    for (int r = 0; r < PIL_VROWS; ++r) {
        const int oy = blockIdx.y * PIL_VROWS + r;
        if (oy >= OH) break;
        input_store(out, out_st, b, oy, ox, c, PH, PW, (float)vv[r] - mean);
        if (resized_u8) resized_u8[((size_t)b * OH + oy) * OW * 3 + o] = (uint8_t)vv[r];
    }
}

// The same with a 4-byte-aligned row pitch of the intermediate image (the context's own buffer; <= 8 taps): a thread takes FOUR
// consecutive samples with one dword load per tap.  The byte form issues one 64-byte wave load per tap and sample column and is
// paced by the texture addresser (70 us per 8 frames at 1.7 TB/s); this one moves the same bytes in a quarter of the loads.
__global__ __launch_bounds__(256) void pil_resize_v_norm_dw(const uint8_t* __restrict__ tmp, void* __restrict__ out, int out_st,
                                                            const int* __restrict__ bounds, const int* __restrict__ coef,
                                                            int OH, int OW, int ksize, int PH, int PW,
                                                            float m0, float m1, float m2, size_t tmp_img_stride, int tp,
                                                            uint8_t* __restrict__ resized_u8) {
    const int b = blockIdx.z;
    const int o0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (o0 >= OW * 3) return;
    const uint32_t* tb = reinterpret_cast<const uint32_t*>(tmp + (size_t)b * tmp_img_stride + o0);
    const int tpw = tp >> 2;
    int vv[PIL_VROWS][4];
#pragma unroll
    for (int r = 0; r < PIL_VROWS; ++r) {
        const int oy = blockIdx.y * PIL_VROWS + r;
        int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0, s3 = s0;
        if (oy < OH) {
            const int ymin = bounds[2 * oy], cnt = bounds[2 * oy + 1];
            const int* k = coef + oy * ksize;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool on = j < cnt;
                const uint32_t w = tb[(size_t)(ymin + (on ? j : 0)) * tpw];
                const int kk = on ? k[on ? j : 0] : 0;
                s0 += (int)(w & 0xffu) * kk; s1 += (int)((w >> 8) & 0xffu) * kk; s2 += (int)((w >> 16) & 0xffu) * kk; s3 += (int)(w >> 24) * kk;
            }
        }
        vv[r][0] = clip8(s0); vv[r][1] = clip8(s1); vv[r][2] = clip8(s2); vv[r][3] = clip8(s3);
    }
#pragma unroll
    for (int r = 0; r < PIL_VROWS; ++r) {
        const int oy = blockIdx.y * PIL_VROWS + r;
        if (oy >= OH) break;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = o0 + i;
            if (o >= OW * 3) break;
            const int ox = o / 3, c = o - ox * 3;
            const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
            input_store(out, out_st, b, oy, ox, c, PH, PW, (float)vv[r][i] - mean);
            if (resized_u8) resized_u8[((size_t)b * OH + oy) * OW * 3 + o] = (uint8_t)vv[r][i];
        }
    }
}

// f32 CHW (the reference's model input, track_predictor.py:49) -> normalised padded NHWC4.
__global__ __launch_bounds__(256) void chw_to_nhwc4_norm(const float* __restrict__ img, void* __restrict__ out, int out_st, int OH,
                                                         int OW, int PH, int PW, float m0, float m1, float m2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= OH * OW) return;
    const int oy = i / OW, ox = i - oy * OW;
    const float* p = img + (size_t)b * 3 * OH * OW;
    if (out_st == 0) {
        f32x4 v = {p[i] - m0, p[(size_t)OH * OW + i] - m1, p[(size_t)2 * OH * OW + i] - m2, 0.f};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + (((size_t)b * PH + oy) * PW + ox) * 4) = v;
    } else {
        input_store(out, out_st, b, oy, ox, 0, PH, PW, p[i] - m0);
        input_store(out, out_st, b, oy, ox, 1, PH, PW, p[(size_t)OH * OW + i] - m1);
        input_store(out, out_st, b, oy, ox, 2, PH, PW, p[(size_t)2 * OH * OW + i] - m2);
    }
}

// max_pool2d(k=3, s=2, p=1) on NHWC (f32 or 16-bit storage), C % 4 == 0.  One thread per (pixel, 4 channels).
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc(const void* __restrict__ x, void* __restrict__ y, int B, int H,
                                                         int W, int C, int OH, int OW, int st) {
    const int c4 = C >> 2;
    const size_t total = (size_t)B * OH * OW * c4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % c4);
        size_t pix = e / c4;
        const int ox = (int)(pix % OW);
        pix /= OW;
        const int oy = (int)(pix % OH);
        const int b = (int)(pix / OH);
        f32x4 m = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = apse_ld4(x, (((size_t)b * H + iy) * W + ix) * C + c * 4, st);
                m[0] = v[0] > m[0] ? v[0] : m[0];
                m[1] = v[1] > m[1] ? v[1] : m[1];
                m[2] = v[2] > m[2] ? v[2] : m[2];
                m[3] = v[3] > m[3] ? v[3] : m[3];
            }
        }
        apse_st4(y, (((size_t)b * OH + oy) * OW + ox) * C + c * 4, m, st);
    }
}

// the same on 16-bit storage: one thread per (pixel, 8 channels) -- 16-byte loads and stores, half the load instructions of the
// 4-channel form (the stem output is the largest activation of the network: 8 x 33 MB at batch 8); max is exact in any order
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc16(const uint16_t* __restrict__ x, void* __restrict__ y, int B, int H,
                                                           int W, int C, int OH, int OW, int st) {
    const int c8 = C >> 3;
    const size_t total = (size_t)B * OH * OW * c8;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % c8);
        size_t pix = e / c8;
        const int ox = (int)(pix % OW);
        pix /= OW;
        const int oy = (int)(pix % OH);
        const int b = (int)(pix / OH);
        uint4 raw[9];
        bool in[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = oy * 2 - 1 + t / 3, ix = ox * 2 - 1 + t % 3;
            in[t] = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const int cy = in[t] ? iy : oy * 2, cx = in[t] ? ix : ox * 2;         // the window centre is always inside
            raw[t] = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + cy) * W + cx) * C + c * 8);
        }
        f32x8 m = apse_cvt8(raw[4], st);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t == 4) continue;
            const f32x8 v = apse_cvt8(raw[t], st);       // a tap outside the map re-reads the centre: max unchanged
#pragma unroll
            for (int k = 0; k < 8; ++k) m[k] = v[k] > m[k] ? v[k] : m[k];
        }
        apse_st8(y, (((size_t)b * OH + oy) * OW + ox) * C + c * 8, m, st);
    }
}

// max_pool2d(k=1, s=2): p6 = p5[:, ::2, ::2]
__global__ __launch_bounds__(256) void subsample2_nhwc(const void* __restrict__ x, void* __restrict__ y, int B, int H,
                                                       int W, int C, int OH, int OW, int st) {
    const int c4 = C >> 2;
    const size_t total = (size_t)B * OH * OW * c4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % c4);
        size_t pix = e / c4;
        const int ox = (int)(pix % OW);
        pix /= OW;
        const int oy = (int)(pix % OH);
        const int b = (int)(pix / OH);
        apse_st4(y, (((size_t)b * OH + oy) * OW + ox) * C + c * 4,
                 apse_ld4(x, (((size_t)b * H + 2 * oy) * W + 2 * ox) * C + c * 4, st), st);
    }
}

// NHWC -> NCHW copy (exposes p2..p6 in the layout the reference API returns).
__global__ __launch_bounds__(256) void nhwc_to_nchw(const void* __restrict__ x, float* __restrict__ y, int B, int HW, int C, int st) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        float v = 0.f;
        if (p < HW && c < C) {
            const size_t idx = ((size_t)b * HW + p) * C + c;
            if (st == 0) v = reinterpret_cast<const float*>(x)[idx];
            else if (st == 1) v = __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(x)[idx] << 16);
            else v = (float)reinterpret_cast<const _Float16*>(x)[idx];
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        if (p < HW && c < C) y[((size_t)b * C + c) * HW + p] = tile[tx][i];
    }
}

// f32 -> bf16 (dtype 1) / f16 (dtype 2), round-to-nearest-even (v_cvt_pk_bf16_f32 / v_cvt_f16_f32), NaN stays NaN
__global__ __launch_bounds__(256) void round16_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, size_t n, int dtype) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) apse_st1(y, i, x[i], dtype);
}

extern "C" {
int apse_k_round16(const float* x, uint16_t* y, size_t n, int dtype, hipStream_t s) {
    if (n == 0) return APSE_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(round16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, y, n, dtype);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
int apse_k_pil_resize(const uint8_t* src, uint8_t* tmp, void* out, int out_st, uint8_t* resized_u8, const int* hb, const int* hc,
                      int hk, const int* vb, const int* vc, int vk, int B, int H, int W, int OH, int OW, int PH, int PW,
                      const float* mean, const UndistortParams* cam, const LabTables* lut, const void* cam_map, const int* hcT, int tmp_pitch,
                      hipStream_t s) {
    // tmp_pitch: row pitch of tmp in bytes (0: OW * 3); a multiple of 4 selects the dword vertical pass
    const int tp = tmp_pitch > 0 ? tmp_pitch : OW * 3;
    if (tp < OW * 3) return APSE_E_INVALID;
    // hcT (optional, device): the horizontal taps tap-major [8][OW], zero-filled (apse_set_resize_tables builds it)
    if ((W & 3) != 0 || (size_t)W * 3 > 150000) return APSE_E_INVALID;
    UndistortParams none;
    memset(&none, 0, sizeof none);
    const bool fused = cam && (cam->do_undistort || cam->do_gamma);
    if (fused && (cam->H != H || cam->W != W)) return APSE_E_INVALID;
    // at most 8 taps: the batched form
    const bool h8 = hk <= 8 && (reinterpret_cast<uintptr_t>(hb) & 7) == 0;
    const size_t lds8 = (((size_t)W * 3 + 32 + 15) & ~(size_t)15) + (size_t)OW * 3 + 32;
    const int2* hb2 = reinterpret_cast<const int2*>(hb);
    // the compact remap table and the 12-byte gathers need 32-bit byte offsets into the batch of frames
    const size_t src_total = (size_t)B * H * W * 3;
    const uint32_t* map2 = src_total < 0x7ffffff0ull ? reinterpret_cast<const uint32_t*>(cam_map) : nullptr;
    const unsigned sb = (unsigned)(src_total < 0x7ffffff0ull ? src_total : 0);
    if (h8 && fused)
        hipLaunchKernelGGL(pil_resize_h8<true>, dim3(((H + 7) / 8) * 8 * B), dim3(256), lds8, s, src, tmp, hb2, hc, H, W, OW, hk, (size_t)H * W * 3,
                           (size_t)H * tp, *cam, lut, map2, hcT, tp, sb, B);
    else if (h8)
        hipLaunchKernelGGL(pil_resize_h8<false>, dim3(H, B), dim3(256), lds8, s, src, tmp, hb2, hc, H, W, OW, hk, (size_t)H * W * 3,
                           (size_t)H * tp, none, (const LabTables*)nullptr, (const uint32_t*)nullptr, hcT, tp, 0u, 0);
    else if (fused)
        hipLaunchKernelGGL(pil_resize_h<true>, dim3(H, B), dim3(256), (size_t)W * 3, s, src, tmp, hb, hc, H, W, OW, hk,
                           (size_t)H * W * 3, (size_t)H * tp, *cam, lut, map2, tp, sb);
    else
        hipLaunchKernelGGL(pil_resize_h<false>, dim3(H, B), dim3(256), (size_t)W * 3, s, src, tmp, hb, hc, H, W, OW, hk,
                           (size_t)H * W * 3, (size_t)H * tp, none, (const LabTables*)nullptr, (const uint32_t*)nullptr, tp, 0u);
    if ((tp & 3) == 0 && vk <= 8 && (reinterpret_cast<uintptr_t>(tmp) & 3) == 0)
        hipLaunchKernelGGL(pil_resize_v_norm_dw, dim3(((OW * 3 + 3) / 4 + 255) / 256, (OH + PIL_VROWS - 1) / PIL_VROWS, B), dim3(256), 0, s, tmp, out,
                           out_st, vb, vc, OH, OW, vk, PH, PW, mean[0], mean[1], mean[2], (size_t)H * tp, tp, resized_u8);
    else
        hipLaunchKernelGGL(pil_resize_v_norm, dim3((OW * 3 + 255) / 256, (OH + PIL_VROWS - 1) / PIL_VROWS, B), dim3(256), 0, s, tmp, out, out_st, vb, vc,
                           OH, OW, vk, PH, PW, mean[0], mean[1], mean[2], (size_t)H * tp, tp, resized_u8);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
int apse_k_chw_norm(const float* img, void* out, int out_st, int B, int OH, int OW, int PH, int PW, const float* mean, hipStream_t s) {
    hipLaunchKernelGGL(chw_to_nhwc4_norm, dim3((OH * OW + 255) / 256, B), dim3(256), 0, s, img, out, out_st, OH, OW, PH, PW, mean[0],
                       mean[1], mean[2]);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
int apse_k_maxpool3x3s2(const void* x, void* y, int B, int H, int W, int C, int st, hipStream_t s) {
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    size_t total = (size_t)B * OH * OW * (C / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (st != 0 && (C & 7) == 0) {
        total = (size_t)B * OH * OW * (C / 8);
        blocks = (int)((total + 255) / 256);
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(maxpool3x3s2_nhwc16, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const uint16_t*>(x), y, B, H, W, C, OH, OW, st);
        return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
    }
    hipLaunchKernelGGL(maxpool3x3s2_nhwc, dim3(blocks), dim3(256), 0, s, x, y, B, H, W, C, OH, OW, st);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
int apse_k_subsample2(const void* x, void* y, int B, int H, int W, int C, int st, hipStream_t s) {
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    size_t total = (size_t)B * OH * OW * (C / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(subsample2_nhwc, dim3(blocks), dim3(256), 0, s, x, y, B, H, W, C, OH, OW, st);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
int apse_k_nhwc_to_nchw(const void* x, float* y, int B, int HW, int C, int st, hipStream_t s) {
    hipLaunchKernelGGL(nhwc_to_nchw, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0, s, x, y, B, HW, C, st);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
}
