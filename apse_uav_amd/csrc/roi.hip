// ROI gather kernels (gfx950), NHWC f32 feature maps with C = 256 (one wave = 64 lanes x float4).
//  * roi_align_nhwc : detectron2 ROIPooler + ROIAlign(aligned=True, sampling_ratio=0) over p2..p5
//                     (box head 7x7, mask head 14x14; reached from dcnn/networks/track_rcnn.py:51)
//  * roi_pool_nhwc  : torchvision.ops.roi_pool on p2 (dcnn/engines/rcnn_tracker.py:182)
//  * l2_normalize, sqdist_matrix : AssociationHead's F.normalize (dcnn/networks/association_head.py:25)
//                     and RcnnTracker.calculate_distance_matrix (dcnn/engines/rcnn_tracker.py:192-221)
// Output layout is [roi][ph][pw][C]; the consuming FC weights are permuted (c,h,w)->(h,w,c) at load.
#include "apse_common.h"
#include <float.h>

struct FpnMaps {
    const void* p[4];      // p2..p5, each [B][H][W][256], f32 or 16-bit (st)
    int H[4], W[4];
    float scale[4];        // 1/4 .. 1/32
    int st;                // storage type of the maps: 0 f32, 1 bf16, 2 f16
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// ---- ROIAlign ----------------------------------------------------------------------------------------------------------
// One block per roi, one wave per output bin.  rois: [n_max][4]; image of roi r: roi_img ? roi_img[r] : r / per_img.
// live rois: (roi_img ? r < *total : (r % per_img) < cnt[r / per_img]); dead rois write zeros.
//
// A bin averages gh x gw bilinear samples (adaptive grid: g = ceil(roi extent / R)), 4 taps each.  The sample grid is a
// product grid and the clamp / skip rules act per axis, so the sum is separable:
//     sum_iy sum_ix bilinear(y_iy, x_ix) = sum_Y sum_X wy[Y] wx[X] F[Y][X],
// wy[Y] = sum of the row weights (hy or ly) the samples put on map row Y, wx likewise.  Neighbouring samples are at most
// one cell apart, so the taps cover a contiguous (<= gh + 1) x (<= gw + 1) window: a 4 x 4 grid reads 25 cells instead
// of 64 taps.  Windows of at most RA_CAP cells per axis (larger ones -- boxes that are long and thin on a fine level --
// take the per-sample loop).  16-bit maps: a cell's 256 channels are 512 B, so the
// two half-waves take alternate cells of the window with 16 B per lane and are added at the end.
// f32 sums in a fixed order: (Y, X) row-major per half-wave.
#define RA_CAP 16
#ifndef RA_FLIGHT
#define RA_FLIGHT 4        // cell loads requested together per lane.  Round 3 (rocprofv3, fp16 batch 8 / f32 batch 1, us per launch): 8 loads:
#endif                     // 195.9 / 34.8 (124 / 83 VGPRs: 4-5 waves per SIMD); 4 loads: 169.0 / 33.6 (88 / 72 VGPRs); 16 loads: 360 / 42.6;
                           // a forced 6 blocks per CU (launch bounds) spills: 603 / 38.1

struct RaAxis { int base, n; float w; bool ok; };

// taps of one axis of one bin: s0 = roi start (map cells, -0.5 shifted), pi = bin index, bsz = bin size, g = samples, L = map extent
__device__ __forceinline__ RaAxis ra_axis(float s0, int pi, float bsz, int g, int L, int lane) {
    RaAxis a;
    int lo = 0x7fffffff, hi = -1;
    for (int i = 0; i < g; ++i) {
        float v = s0 + (float)pi * bsz + ((float)i + 0.5f) * bsz / (float)g;
        if (v < -1.0f || v > (float)L) continue;
        if (v <= 0.f) v = 0.f;
        int l = (int)v, h;
        if (l >= L - 1) { h = l = L - 1; } else { h = l + 1; }
        lo = l < lo ? l : lo; hi = h > hi ? h : hi;
    }
    a.base = lo; a.n = hi < 0 ? 0 : hi - lo + 1; a.w = 0.f;
    a.ok = a.n <= RA_CAP;
    if (!a.ok || a.n == 0) return a;
    for (int i = 0; i < g; ++i) {
        float v = s0 + (float)pi * bsz + ((float)i + 0.5f) * bsz / (float)g;
        if (v < -1.0f || v > (float)L) continue;
        if (v <= 0.f) v = 0.f;
        int l = (int)v, h;
        if (l >= L - 1) { h = l = L - 1; v = (float)l; } else { h = l + 1; }
        const float lv = v - (float)l, hv = 1.f - lv;
        if (l - lo == lane) a.w += hv;
        if (h - lo == lane) a.w += lv;
    }
    return a;
}

// the per-sample form (any window size); also the statement the separable form is derived from
__device__ __forceinline__ f32x4 ra_bin_direct(const void* fmap, int st, size_t f, int H, int W, float sh, float sw, float bh,
                                                float bw, int gh, int gw, int ph, int pw) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int iy = 0; iy < gh; ++iy) {
        float y = sh + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
        if (y < -1.0f || y > (float)H) continue;
        if (y <= 0.f) y = 0.f;
        int yl = (int)y, yh;
        if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else { yh = yl + 1; }
        const float ly = y - (float)yl, hy = 1.f - ly;
        for (int ix = 0; ix < gw; ++ix) {
            float x = sw + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
            if (x < -1.0f || x > (float)W) continue;
            if (x <= 0.f) x = 0.f;
            int xl = (int)x, xh;
            if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else { xh = xl + 1; }
            const float lx = x - (float)xl, hx = 1.f - lx;
            const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
            const f32x4 v1 = apse_ld4(fmap, f + ((size_t)yl * W + xl) * 256, st);
            const f32x4 v2 = apse_ld4(fmap, f + ((size_t)yl * W + xh) * 256, st);
            const f32x4 v3 = apse_ld4(fmap, f + ((size_t)yh * W + xl) * 256, st);
            const f32x4 v4 = apse_ld4(fmap, f + ((size_t)yh * W + xh) * 256, st);
            acc += w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
        }
    }
    return acc;
}

#define RA_MAXR 14          // largest pooler resolution (mask head)

#ifndef RA_MINBLK
#define RA_MINBLK 1        // second argument of __launch_bounds__: minimum 256-thread blocks per CU the register budget must allow
#endif
// ST: storage type of the maps at compile time (0 f32, 1 bf16, 2 f16)
template <int ST>
__global__ __launch_bounds__(256, RA_MINBLK) void roi_align_nhwc(const FpnMaps F, const float* __restrict__ rois,
                                                      const int* __restrict__ roi_img, const int* __restrict__ cnt,
                                                      const int* __restrict__ total, int per_img, int n_max, int R,
                                                      void* __restrict__ out, int out_st) {
    // One block per roi: the tap tables of its R row bins and R column bins are built once, 16 lanes per (axis, bin) --
    // per bin this arithmetic is wave-uniform and would otherwise be repeated by all 64 lanes of all R*R bin waves
    // (it was 2/3 of the kernel's instructions) -- then the four waves take the R*R bins round-robin.
    __shared__ float wtab[2][RA_MAXR][RA_CAP];
    __shared__ int btab[2][RA_MAXR][2];
    __shared__ int all_ok;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rr = R * R;
    const int nlive = roi_img ? (*total < n_max ? *total : n_max) : n_max;
    const int nparts = gridDim.y, part = blockIdx.y;       // a roi's bins are dealt over gridDim.y blocks (few rois: more blocks)
    for (int r = blockIdx.x; r < nlive; r += gridDim.x) {
    int img;
    bool live;
    if (roi_img) {
        live = r < *total;
        img = live ? roi_img[r] : 0;
    } else {
        img = r / per_img;
        live = (r - img * per_img) < cnt[img];
    }
    if (!live) {
        if (!roi_img)
            for (int pb = part * 4 + wave; pb < rr; pb += 4 * nparts) apse_st4(out, ((size_t)r * rr + pb) * 256 + lane * 4, f32x4{0.f, 0.f, 0.f, 0.f}, out_st);
        continue;                         // packed list: rows past the count are never read
    }
    const float x1 = rois[r * 4 + 0], y1 = rois[r * 4 + 1], x2 = rois[r * 4 + 2], y2 = rois[r * 4 + 3];
    // assign_boxes_to_levels: floor(4 + log2(sqrt(area) / 224 + eps)), clamped to [2, 5]
    const float sz = sqrtf((x2 - x1) * (y2 - y1));
    float lvf = floorf(4.0f + log2f(sz / 224.0f + 2.220446049250313e-16f));
    lvf = fminf(fmaxf(lvf, 2.f), 5.f);
    const int lv = (int)lvf - 2;
    const int H = F.H[lv], W = F.W[lv];
    const float sc = F.scale[lv];
    const void* fmap = F.p[lv];
    const size_t fimg = (size_t)img * H * W * 256;
    const float sw = x1 * sc - 0.5f, sh = y1 * sc - 0.5f;
    const float ew = x2 * sc - 0.5f, eh = y2 * sc - 0.5f;
    const float rw = ew - sw, rh = eh - sh;
    const float bw = rw / (float)R, bh = rh / (float)R;
    const int gh = (int)ceilf(rh / (float)R), gw = (int)ceilf(rw / (float)R);
    const float cntf = (float)((gh * gw) > 1 ? gh * gw : 1);
    __syncthreads();                      // the previous roi's tables are no longer read
    if (threadIdx.x == 0) all_ok = 1;
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * R * RA_CAP; e += blockDim.x) {
        const int axis = e / (R * RA_CAP), b = (e / RA_CAP) % R, c = e & (RA_CAP - 1);
        const RaAxis a = axis == 0 ? ra_axis(sh, b, bh, gh, H, c) : ra_axis(sw, b, bw, gw, W, c);
        wtab[axis][b][c] = a.w;
        if (c == 0) { btab[axis][b][0] = a.base; btab[axis][b][1] = a.n; if (!a.ok) all_ok = 0; }
    }
    __syncthreads();
    const bool sep = all_ok != 0;
    for (int pb = part * 4 + wave; pb < rr; pb += 4 * nparts) {
    const int ph = pb / R, pw = pb - ph * R;
    const size_t o = ((size_t)r * rr + pb) * 256 + lane * 4;
    if (!sep) {
        const f32x4 acc = ra_bin_direct(fmap, F.st, fimg + lane * 4, H, W, sh, sw, bh, bw, gh, gw, ph, pw);
        apse_st4(out, o, acc / cntf, out_st);
        continue;
    }
    const int ybase = btab[0][ph][0], ny = btab[0][ph][1], xbase = btab[1][pw][0], nx = btab[1][pw][1];
    const float* wyv = wtab[0][ph];
    const float* wxv = wtab[1][pw];
    if constexpr (ST == 0) {
        // f32 maps: one cell = 64 lanes x 16 B
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int T = ny * nx;
        const size_t f0 = fimg + ((size_t)ybase * W + xbase) * 256 + lane * 4;
        int yy = 0, xx = 0;                                  // window cell of element t, advanced with t
        for (int t0 = 0; t0 < T; t0 += RA_FLIGHT) {
            f32x4 v[RA_FLIGHT];
            float w[RA_FLIGHT];
#pragma unroll
            for (int u = 0; u < RA_FLIGHT; ++u) {
                const bool in = yy < ny;
                w[u] = in ? wyv[in ? yy : 0] * wxv[xx] : 0.f;
                v[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(fmap) + f0 + ((size_t)(in ? yy : 0) * W + xx) * 256);
                if (++xx == nx) { xx = 0; ++yy; }
            }
#pragma unroll
            for (int u = 0; u < RA_FLIGHT; ++u) acc += w[u] * v[u];
        }
        apse_st4(out, o, acc / cntf, out_st);
    } else {
        // 16-bit maps: half-wave h takes the cells with (X & 1) == h, 32 lanes x 16 B each.  The cell stream (row-major over the window,
        // RA_FLIGHT loads in flight) is addressed with ONE per-lane byte offset -- the lane's chunk of window cell (0, half) -- plus a
        // wave-uniform scalar offset that walks the cells; the descriptor ends with this image's map, so whatever lies past it reads
        // as zeros, and the padding iterations of the last group get an offset past the range.  The x weight of a cell past the
        // window's last column is 0 in the tap table (no sample touches it), so such a lane adds +0 as before.  (Round 4: the loop
        // was bound by the scalar unit -- 64-bit cell indices with selects, ~31 scalar + 23 vector instructions per cell.)
        const int half = lane >> 5, cg = lane & 31;
        const int np = (nx + 1) >> 1;                       // cell pairs per window row
        const int T = ny * np;
        const size_t cell0 = fimg + ((size_t)ybase * W + xbase) * 256;                 // element index of window cell (0, 0)
        const size_t img_end = (fimg + (size_t)H * W * 256) * 2;                       // bytes up to the end of this image's map
        f32x8 acc;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        if (img_end < 0x7ffffff0ull) {
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(fmap), 0, (int)img_end, 0x00020000);
            const unsigned voff = (unsigned)((cell0 + (size_t)half * 256 + cg * 8) * 2);
            const unsigned rowb = (unsigned)W * 512u;
            const float* wxh = wxv + half;                   // this half-wave's x weights: wxh[2 p]
            int yy = 0, xp = 0;
            unsigned soff = 0;
            for (int t0 = 0; t0 < T; t0 += RA_FLIGHT) {
                uint4 raw[RA_FLIGHT];
                float w[RA_FLIGHT];
#pragma unroll
                for (int u = 0; u < RA_FLIGHT; ++u) {
                    const bool live = yy < ny;
                    w[u] = wyv[live ? yy : 0] * wxh[2 * xp];
                    raw[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(live ? voff : 0x80000000u), (int)soff, 0));
                    soff += 1024u;
                    if (++xp == np) { xp = 0; ++yy; soff = (unsigned)yy * rowb; }
                }
#pragma unroll
                for (int u = 0; u < RA_FLIGHT; ++u) {
                    const f32x8 v = apse_cvt8(raw[u], ST);
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] += w[u] * v[k];
                }
            }
        } else {
        const size_t f0 = cell0 + cg * 8;
        int yy = 0, xp = 0;
        for (int t0 = 0; t0 < T; t0 += RA_FLIGHT) {
            uint4 raw[RA_FLIGHT];                            // RA_FLIGHT cell loads in flight per lane, converted when consumed
            float w[RA_FLIGHT];
#pragma unroll
            for (int u = 0; u < RA_FLIGHT; ++u) {
                const int xx = 2 * xp + half;
                const bool in = yy < ny && xx < nx;
                w[u] = in ? wyv[in ? yy : 0] * wxv[in ? xx : 0] : 0.f;
                raw[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(fmap) + f0 +
                                                         ((size_t)(in ? yy : 0) * W + (in ? xx : 0)) * 256);
                if (++xp == np) { xp = 0; ++yy; }
            }
#pragma unroll
            for (int u = 0; u < RA_FLIGHT; ++u) {
                const f32x8 v = apse_cvt8(raw[u], ST);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += w[u] * v[k];
            }
        }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = (acc[k] + __shfl_xor(acc[k], 32)) / cntf;
        if (half == 0) apse_st8(out, ((size_t)r * rr + pb) * 256 + cg * 8, acc, out_st);
    }
    }
    }
}

// torchvision roi_pool forward on one NHWC map; rois in original-frame pixels, packed list.
// One BLOCK per output bin: the four waves take the bin's rows round-robin (a bin of a frame-sized box covers hundreds
// of cells), two cells per iteration in flight, and meet in LDS for the final max.  roi_img == nullptr: every roi
// reads image img0; total == nullptr: all n_max rois are live; nchw: write [roi][C][R][R] (the layout
// RoiFeaturesGenerator returns) instead of [roi][R][R][C].
__global__ __launch_bounds__(256) void roi_pool_nhwc(const void* __restrict__ feat, int st, int H, int W,
                                                     const float* __restrict__ rois, const int* __restrict__ roi_img,
                                                     const int* __restrict__ total, int n_max, int R, float scale,
                                                     float* __restrict__ out, int img0, int nchw) {
    __shared__ f32x4 part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rr = R * R;
    const int nl = total ? (*total < n_max ? *total : n_max) : n_max;
    const int live = nl * rr;
    for (int bin = blockIdx.x; bin < live; bin += gridDim.x) {
        const int r = bin / rr;
        const int pb = bin - r * rr;
        const int ph = pb / R, pw = pb - ph * R;
        const size_t f = (size_t)(roi_img ? roi_img[r] : img0) * H * W * 256 + lane * 4;
        const int sw = (int)roundf(rois[r * 4 + 0] * scale), sh = (int)roundf(rois[r * 4 + 1] * scale);
        const int ew = (int)roundf(rois[r * 4 + 2] * scale), eh = (int)roundf(rois[r * 4 + 3] * scale);
        const int rw = (ew - sw + 1) > 1 ? (ew - sw + 1) : 1;
        const int rh = (eh - sh + 1) > 1 ? (eh - sh + 1) : 1;
        const float bh = (float)rh / (float)R, bw = (float)rw / (float)R;
        int hs = (int)floorf((float)ph * bh), he = (int)ceilf((float)(ph + 1) * bh);
        int ws = (int)floorf((float)pw * bw), we = (int)ceilf((float)(pw + 1) * bw);
        hs = min(max(hs + sh, 0), H); he = min(max(he + sh, 0), H);
        ws = min(max(ws + sw, 0), W); we = min(max(we + sw, 0), W);
        const bool empty = (he <= hs) || (we <= ws);
        f32x4 m = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
        auto mx = [&](const f32x4 v) {
            m[0] = v[0] > m[0] ? v[0] : m[0];
            m[1] = v[1] > m[1] ? v[1] : m[1];
            m[2] = v[2] > m[2] ? v[2] : m[2];
            m[3] = v[3] > m[3] ? v[3] : m[3];
        };
        // two cell loads in flight per lane.  (Round 4: eight in flight measured SLOWER, 105 -> 118 us per 4 frames at ~38 detections
        // per frame -- the kernel waits (84 % of its wave time parked) because it moves ~1.4 GB of cell rows per launch through L2
        // (boxes hundreds of pixels wide max-pooled on the stride-4 map: the reference's choice of p2), not for want of requests.)
        for (int y = hs + wave; y < he; y += 4) {
            int x = ws;
            for (; x + 1 < we; x += 2) {
                const f32x4 v0 = apse_ld4(feat, f + ((size_t)y * W + x) * 256, st);
                const f32x4 v1 = apse_ld4(feat, f + ((size_t)y * W + x + 1) * 256, st);
                mx(v0);
                mx(v1);
            }
            if (x < we) mx(apse_ld4(feat, f + ((size_t)y * W + x) * 256, st));
        }
        part[wave][lane] = m;
        __syncthreads();
        if (wave == 0) {
            mx(part[1][lane]); mx(part[2][lane]); mx(part[3][lane]);
            if (empty) m = f32x4{0.f, 0.f, 0.f, 0.f};
            if (nchw) {
                for (int k = 0; k < 4; ++k) out[((size_t)r * 256 + lane * 4 + k) * rr + pb] = m[k];
            } else {
                *reinterpret_cast<f32x4*>(out + ((size_t)r * rr + pb) * 256 + lane * 4) = m;
            }
        }
        __syncthreads();
    }
}

// F.interpolate(mask.float(), size=(OH, OW), mode='bilinear') with align_corners=False
// (dcnn/engines/roi_features_generator.py:99-101): src = max(0, (dst + 0.5) * in/out - 0.5), clamped neighbours.
__global__ __launch_bounds__(256) void mask_resize_bilinear(const uint8_t* __restrict__ masks, int n, int H, int W, int OH, int OW,
                                                            float* __restrict__ out) {
    const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
    const size_t total = (size_t)n * OH * OW;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int ox = (int)(e % OW);
        const int oy = (int)((e / OW) % OH);
        const int i = (int)(e / ((size_t)OW * OH));
        float fy = sh * ((float)oy + 0.5f) - 0.5f;
        float fx = sw * ((float)ox + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const uint8_t* mp = masks + (size_t)i * H * W;
        const float v00 = mp[(size_t)y0 * W + x0] ? 1.f : 0.f, v01 = mp[(size_t)y0 * W + x1] ? 1.f : 0.f;
        const float v10 = mp[(size_t)y1 * W + x0] ? 1.f : 0.f, v11 = mp[(size_t)y1 * W + x1] ? 1.f : 0.f;
        out[e] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    }
}

// torchvision.ops.roi_align (aligned=False, sampling_ratio = SR) over feat * mask[roi]
// (dcnn/engines/roi_features_generator.py:102-111; the same branch is rcnn_tracker.py:166-180).  One wave per bin.
// rois: frame pixels; mask: [n][H][W] f32 at feature resolution; out: [roi][256][R][R].
__global__ __launch_bounds__(256) void roi_align_masked(const void* __restrict__ feat, int st, int H, int W, int img0,
                                                        const float* __restrict__ rois, const float* __restrict__ mask,
                                                        int n, int R, int SR, float scale, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int rr = R * R;
    for (int bin = blockIdx.x * 4 + (threadIdx.x >> 6); bin < n * rr; bin += gridDim.x * 4) {
        const int r = bin / rr;
        const int pb = bin - r * rr;
        const int ph = pb / R, pw = pb - ph * R;
        const size_t f = (size_t)img0 * H * W * 256 + lane * 4;
        const float* mk = mask + (size_t)r * H * W;
        const float x1 = rois[r * 4 + 0] * scale, y1 = rois[r * 4 + 1] * scale;
        const float x2 = rois[r * 4 + 2] * scale, y2 = rois[r * 4 + 3] * scale;
        const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
        const float bw = rw / (float)R, bh = rh / (float)R;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int iy = 0; iy < SR; ++iy) {
            const float yy = y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)SR;
            for (int ix = 0; ix < SR; ++ix) {
                const float xx = x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)SR;
                if (yy < -1.f || yy > (float)H || xx < -1.f || xx > (float)W) continue;
                float y = yy <= 0.f ? 0.f : yy, x = xx <= 0.f ? 0.f : xx;
                int yl = (int)y, xl = (int)x, yh, xh;
                if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
                if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
                const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
                const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                const f32x4 v1 = apse_ld4(feat, f + ((size_t)yl * W + xl) * 256, st) * mk[(size_t)yl * W + xl];
                const f32x4 v2 = apse_ld4(feat, f + ((size_t)yl * W + xh) * 256, st) * mk[(size_t)yl * W + xh];
                const f32x4 v3 = apse_ld4(feat, f + ((size_t)yh * W + xl) * 256, st) * mk[(size_t)yh * W + xl];
                const f32x4 v4 = apse_ld4(feat, f + ((size_t)yh * W + xh) * 256, st) * mk[(size_t)yh * W + xh];
                acc += w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
            }
        }
        acc = acc / (float)(SR * SR);
        for (int k = 0; k < 4; ++k) out[((size_t)r * 256 + lane * 4 + k) * rr + pb] = acc[k];
    }
}

// F.normalize(x, p=2, dim=1, eps=1e-12) on [n][D] rows, one wave per row (D <= 256, D % 64 == 0 not required).
__global__ __launch_bounds__(64) void l2_normalize_rows(const float* __restrict__ x, float* __restrict__ y, int D,
                                                        const int* __restrict__ total, int n_max) {
    const int r = blockIdx.x, lane = threadIdx.x;
    if (r >= n_max || (total && r >= *total)) return;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) { const float v = x[(size_t)r * D + i]; s += v * v; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    float nrm = sqrtf(s);
    nrm = nrm < 1e-12f ? 1e-12f : nrm;
    for (int i = lane; i < D; i += 64) y[(size_t)r * D + i] = x[(size_t)r * D + i] / nrm;
}

// ---- AssociationHead FC (dcnn/networks/association_head.py:16-27: Linear(256 * 10 * 10 -> 128) + F.normalize) --------------------
// A handful of rows (one per detection) against a 13 MB filter matrix: a pure filter stream.  As a split-K convolution it was a
// 31 us launch of 128 blocks + a reduce pass + the normalise kernel (42 us, three kernel boundaries).  Here every block takes a
// 128-wide slice of K for ALL output channels: its filter slice (N x 512 B) goes global -> registers once, the live rows are
// multiplied in chunks of 16 with v_mfma_f32_16x16x4_f32 (exact f32), partial sums go to a [slice][row][N] workspace; the
// second kernel adds the slices in a fixed tree (bitwise reproducible), the bias, and normalises the row.
template <int TPW>   // 16-column tiles per wave: N = 64 TPW
__global__ __launch_bounds__(256) void assoc_fc_slices(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ ws,
                                                       const int* __restrict__ total, int n_max, int K) {
    constexpr int N = 64 * TPW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, ks = lane >> 4;
    const int live = total ? (*total < n_max ? *total : n_max) : n_max;
    if (live <= 0) return;
    const size_t kbase = (size_t)blockIdx.x * 128 + 4 * ks;
    f32x4 b[TPW][8];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) b[t][j] = *reinterpret_cast<const f32x4*>(w + (size_t)((wave * TPW + t) * 16 + col) * K + kbase + 16 * j);
    for (int m0 = 0; m0 < live; m0 += 16) {
        const int row = m0 + col;
        const bool in = row < live;
        f32x4 a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)(in ? row : 0) * K + kbase + 16 * j);
            a[j] = in ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 acc[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][i], b[t][j][i], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 4 * ks + r;
                if (m < live) ws[((size_t)blockIdx.x * n_max + m) * N + (wave * TPW + t) * 16 + col] = acc[t][r];
            }
    }
}

// one block of 1024 threads per live row: the slices are added in a FIXED tree -- 1024 / N groups of consecutive slices, each
// summed in order (eight loads in flight), then the groups in order -- + bias -> raw; F.normalize(raw, eps = 1e-12) -> y
__global__ __launch_bounds__(1024) void assoc_fc_finish(const float* __restrict__ ws, const float* __restrict__ bias, float* __restrict__ raw,
                                                        float* __restrict__ y, const int* __restrict__ total, int n_max, int N, int slices) {
    __shared__ float grp[1024];
    __shared__ float part[4];
    const int r = blockIdx.x;
    if (r >= n_max || (total && r >= *total)) return;
    const int groups = 1024 / N;                       // 4, 8 or 16
    const int n = threadIdx.x % N, g = threadIdx.x / N;
    const int per = (slices + groups - 1) / groups;
    const int s_lo = g * per, s_hi = (s_lo + per) < slices ? (s_lo + per) : slices;
    float v = 0.f;
    for (int s0 = s_lo; s0 < s_hi; s0 += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = (s0 + u < s_hi) ? ws[((size_t)(s0 + u) * n_max + r) * N + n] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    grp[g * N + n] = v;
    __syncthreads();
    float sq = 0.f;
    if (threadIdx.x < N) {
        v = grp[n];
        for (int q = 1; q < groups; ++q) v += grp[q * N + n];
        v += bias ? bias[n] : 0.f;
        raw[(size_t)r * N + n] = v;
        sq = v * v;
    }
    if (threadIdx.x < 256) {
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sq;
    }
    __syncthreads();
    float nrm = sqrtf(part[0] + part[1] + part[2] + part[3]);
    nrm = nrm < 1e-12f ? 1e-12f : nrm;
    if (threadIdx.x < N) y[(size_t)r * N + n] = v / nrm;
}

// D[o][n] = sum_k (a[o][k] - b[n][k])^2, one wave per pair.
__global__ __launch_bounds__(64) void sqdist_matrix(const float* __restrict__ a, const float* __restrict__ b, int O, int N,
                                                    int D, float* __restrict__ out) {
    const int o = blockIdx.y, n = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) { const float d = a[(size_t)o * D + i] - b[(size_t)n * D + i]; s += d * d; }
    for (int k = 32; k > 0; k >>= 1) s += __shfl_xor(s, k);
    if (lane == 0) out[(size_t)o * N + n] = s;
}

extern "C" {
int apse_k_roi_align(const FpnMaps* F, const float* rois, const int* roi_img, const int* cnt, const int* total, int per_img,
                     int n_max, int R, void* out, int out_st, hipStream_t s) {
    if (R > RA_MAXR || n_max <= 0) return n_max <= 0 ? APSE_OK : APSE_E_INVALID;
    int blocks = n_max;
    if (roi_img && blocks > 2048) blocks = 2048;
    // ~8192 blocks keep the chip's wave slots full: with few rois (batch 1: 1000 proposals, 8 detections) each roi's bins are
    // dealt over several blocks, each of which rebuilds the (cheap) tap tables.  With one block per roi 1000 blocks x 4 waves
    // walk 12 bins each, one after the other: 155 us per frame against 81 us for one wave per bin.
#ifndef RA_TARGET_BLOCKS
#define RA_TARGET_BLOCKS 8192
#endif
    int parts = RA_TARGET_BLOCKS / blocks;
    const int max_parts = (R * R + 3) / 4;
    parts = parts < 1 ? 1 : (parts > max_parts ? max_parts : parts);
    if (F->st == 1) hipLaunchKernelGGL(roi_align_nhwc<1>, dim3(blocks, parts), dim3(256), 0, s, *F, rois, roi_img, cnt, total, per_img, n_max, R, out, out_st);
    else if (F->st == 2) hipLaunchKernelGGL(roi_align_nhwc<2>, dim3(blocks, parts), dim3(256), 0, s, *F, rois, roi_img, cnt, total, per_img, n_max, R, out, out_st);
    else hipLaunchKernelGGL(roi_align_nhwc<0>, dim3(blocks, parts), dim3(256), 0, s, *F, rois, roi_img, cnt, total, per_img, n_max, R, out, out_st);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
int apse_k_roi_pool(const void* feat, int st, int H, int W, const float* rois, const int* roi_img, const int* total, int n_max,
                    int R, float scale, float* out, int img0, int nchw, hipStream_t s) {
    if (n_max <= 0) return APSE_OK;
    int blocks = n_max * R * R;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(roi_pool_nhwc, dim3(blocks), dim3(256), 0, s, feat, st, H, W, rois, roi_img, total, n_max, R, scale,
                       out, img0, nchw);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
int apse_k_mask_resize(const uint8_t* masks, int n, int H, int W, int OH, int OW, float* out, hipStream_t s) {
    if (n <= 0) return APSE_OK;
    size_t blocks = ((size_t)n * OH * OW + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(mask_resize_bilinear, dim3((unsigned)blocks), dim3(256), 0, s, masks, n, H, W, OH, OW, out);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
int apse_k_roi_align_masked(const void* feat, int st, int H, int W, int img0, const float* rois, const float* mask, int n, int R,
                            int SR, float scale, float* out, hipStream_t s) {
    if (n <= 0) return APSE_OK;
    int blocks = (n * R * R + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(roi_align_masked, dim3(blocks), dim3(256), 0, s, feat, st, H, W, img0, rois, mask, n, R, SR, scale, out);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
int apse_k_l2_normalize(const float* x, float* y, int D, const int* total, int n_max, hipStream_t s) {
    if (n_max <= 0) return APSE_OK;
    hipLaunchKernelGGL(l2_normalize_rows, dim3(n_max), dim3(64), 0, s, x, y, D, total, n_max);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
// x [n_max][K] f32, w [>= N][K] f32, ws [K / 128][n_max][N]; raw / y [n_max][N].  Eligible: K % 128 == 0, N in {64, 128, 256}.
bool apse_assoc_fc_ok(int K, int N) { return K > 0 && (K & 127) == 0 && (N == 64 || N == 128 || N == 256); }
int apse_k_assoc_fc(const float* x, const float* w, const float* bias, float* ws, const int* total, int n_max, int K, int N, float* raw,
                    float* y, hipStream_t s) {
    if (!apse_assoc_fc_ok(K, N)) return APSE_E_INVALID;
    if (n_max <= 0) return APSE_OK;
    const int slices = K / 128;
    if (N == 64) hipLaunchKernelGGL(assoc_fc_slices<1>, dim3(slices), dim3(256), 0, s, x, w, ws, total, n_max, K);
    else if (N == 128) hipLaunchKernelGGL(assoc_fc_slices<2>, dim3(slices), dim3(256), 0, s, x, w, ws, total, n_max, K);
    else hipLaunchKernelGGL(assoc_fc_slices<4>, dim3(slices), dim3(256), 0, s, x, w, ws, total, n_max, K);
    hipLaunchKernelGGL(assoc_fc_finish, dim3(n_max), dim3(1024), 0, s, ws, bias, raw, y, total, n_max, N, slices);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
int apse_k_sqdist(const float* a, const float* b, int O, int N, int D, float* out, hipStream_t s) {
    if (O <= 0 || N <= 0) return APSE_OK;
    hipLaunchKernelGGL(sqdist_matrix, dim3(N, O), dim3(64), 0, s, a, b, O, N, D, out);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
}
