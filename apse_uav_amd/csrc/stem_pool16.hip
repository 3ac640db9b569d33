// stem_s2d_pool16 (gfx950, 16-bit storage modes): the ResNet stem -- 7x7 / stride-2 convolution + FrozenBN + ReLU -- and the
// 3x3 / stride-2 max-pool behind it in ONE kernel (detectron2 BasicStem, reached from
// /root/reference/dcnn/networks/track_rcnn.py:42).
//
// Why.  The stem output is the largest activation of the network ([B][384][672][64]: 33 MB per image in 16 bits): the
// two-kernel form writes it (264 MB at batch 8) and the pool reads it back (389 MB counted, each cell 2.25 times) to keep a
// quarter of it.  Here a block computes the (2 TPH + 1) x (2 TPW + 1) convolution outputs under a TPH x TPW tile of pooled
// pixels into LDS and pools from there: HBM sees the space-to-depth input (16.6 MB per image) and the pooled map (8.3 MB).
//
// Arithmetic = the unfused path's, bit for bit: the stem on the space-to-depth(2) input is a 4x4 / stride-1 convolution over
// 16 channels (elementwise.hip, input_store; detector.hip, stem weights); one v_mfma_f32_32x32x16 per tap (ky, kx) in the
// order ky, kx -- the order conv_igemm walks the same K = 256 -- one f32 accumulator per output, + bias, ReLU, one rounding
// to the storage type, max over the 3x3 window (values are >= 0, so the 16-bit patterns order like the numbers).
//
// Block = 4 waves.  Wave w holds the filters of output channels 32 (w & 1) .. + 31 in registers (16 taps x 16 B per lane)
// for the whole (persistent) block and takes five of the ten 32-row tiles of the 297 convolution outputs; A fragments come
// straight from the input tile in LDS: lane (row, slot) reads the 16 bytes of channels 8 slot .. + 7 of pixel (oy + ky, ox + kx).
#include "apse_common.h"
#include <type_traits>

#define SP_TPH 4
#define SP_TPW 16
#define SP_CH (2 * SP_TPH + 1)        // 9 convolution rows
#define SP_CW (2 * SP_TPW + 1)        // 33 convolution columns
#define SP_IH (SP_CH + 3)             // 12 input rows
#define SP_IW (SP_CW + 3)             // 36 input pixels
#define SP_M (SP_CH * SP_CW)          // 297 convolution outputs per tile
#ifndef SP_OCC
#define SP_OCC 2                      // blocks per CU (3 needs <= 168 registers: 60 spilled)
#endif
#define SP_MT 5                       // 32-row tiles per wave (2 x 5 x 32 = 320 >= 297)

typedef _Float16 sp_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short sp_u16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 sp_mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 sp_mfma(sp_f16x8 a, sp_f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// x: [B][IH][IW][16] (space-to-depth input), w16: [>= 64][4][64], bias: [>= 64], y: [B][PHo][PWo][64]; conv map = IH x IW.
template <int PR>
__global__ __launch_bounds__(256, SP_OCC) void stem_s2d_pool16(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w16,
                                                          const float* __restrict__ bias, uint16_t* __restrict__ y, int B, int IH,
                                                          int IW, int PHo, int PWo, int tiles_y, int tiles_x) {
    typedef typename std::conditional<PR == 1, bf16x8, sp_f16x8>::type op8;
    __shared__ __attribute__((aligned(16))) char in_t[SP_IH * SP_IW * 32];            // 13.8 KB
    __shared__ __attribute__((aligned(16))) uint16_t cv_t[SP_M * 64];                 // 38 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int nt = wave & 1, mh = wave >> 1;

    // filters of this wave's 32 output channels: tap (ky, kx) -> lane (col fr, slot fh): 8 channels
    f32x4 bw[16];
#pragma unroll
    for (int t = 0; t < 16; ++t)
        bw[t] = *reinterpret_cast<const f32x4*>(w16 + ((size_t)(32 * nt + fr) * 4 + (t >> 2)) * 64 + (t & 3) * 16 + fh * 8);
    const float bia = bias[32 * nt + fr];

    // A-fragment base addresses of this wave's five row tiles (rows past 297 read pixel 0: computed, never stored)
    int abase[SP_MT];
#pragma unroll
    for (int i = 0; i < SP_MT; ++i) {
        int m = (mh + 2 * i) * 32 + fr;
        m = m < SP_M ? m : 0;
        const int cy = m / SP_CW, cx = m - cy * SP_CW;
        abase[i] = (cy * SP_IW + cx) * 32 + fh * 16;
    }

    const int tiles = B * tiles_y * tiles_x;
    // input tile of `tile` -> registers: 12 rows x 36 pixels x 32 B = 864 16-byte pieces, zeros outside the map
    constexpr int NPC = (SP_IH * SP_IW * 2 + 255) / 256;       // pieces per thread (4, the last one partly idle)
    f32x4 pre[NPC];
    auto fetch_tile = [&](int tile) {
        const int b = tile / (tiles_y * tiles_x);
        const int tr = tile - b * (tiles_y * tiles_x);
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int iy0 = 2 * ty * SP_TPH - 3, ix0 = 2 * tx * SP_TPW - 3;     // first input pixel (pad 2 above / left of output -1)
#pragma unroll
        for (int q = 0; q < NPC; ++q) {
            const int e = tid + 256 * q;
            const int half = e & 1, pix = e >> 1;
            const int r = pix / SP_IW, cpx = pix - r * SP_IW;
            const int iy = iy0 + r, ix = ix0 + cpx;
            pre[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (e < SP_IH * SP_IW * 2 && (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW)
                pre[q] = *reinterpret_cast<const f32x4*>(x + (((size_t)b * IH + iy) * IW + ix) * 16 + half * 8);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int q = 0; q < NPC; ++q) {
            const int e = tid + 256 * q;
            if (e < SP_IH * SP_IW * 2) *reinterpret_cast<f32x4*>(in_t + (e >> 1) * 32 + (e & 1) * 16) = pre[q];
        }
    };
    if ((int)blockIdx.x < tiles) { fetch_tile(blockIdx.x); store_tile(); }
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int b = tile / (tiles_y * tiles_x);
        const int tr = tile - b * (tiles_y * tiles_x);
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int py0 = ty * SP_TPH, px0 = tx * SP_TPW;
        const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;        // first convolution output of the tile
        __syncthreads();                                       // input tile complete; the previous tile's pooling is done
        // the next tile's input: requested now, parked in registers through the MFMA phase, written behind it
        const bool more = tile + (int)gridDim.x < tiles;
        if (more) fetch_tile(tile + gridDim.x);
        // ---- 16 taps x 5 row tiles, as 3 + 2 (48 accumulator registers at a time), each followed by bias, ReLU, rounding ->
        // convolution tile in LDS
#pragma unroll
        for (int g0 = 0; g0 < SP_MT; g0 += 3) {
            f32x16 acc[3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int toff = ((t >> 2) * SP_IW + (t & 3)) * 32;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if (g0 + i >= SP_MT) continue;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(in_t + abase[g0 + i] + toff);
                    acc[i] = sp_mfma(__builtin_bit_cast(op8, a), __builtin_bit_cast(op8, bw[t]), acc[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (g0 + i >= SP_MT) continue;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int m = (mh + 2 * (g0 + i)) * 32 + (v & 3) + 8 * (v >> 2) + 4 * fh;
                    float val = acc[i][v] + bia;
                    val = apse_relu(val);
                    uint16_t bits;
                    if constexpr (PR == 1) bits = __builtin_bit_cast(uint16_t, (__bf16)val);
                    else bits = __builtin_bit_cast(uint16_t, (_Float16)val);
                    if (m < SP_M) cv_t[m * 64 + 32 * nt + fr] = bits;
                }
            }
        }
        __syncthreads();                                       // convolution tile complete; nobody reads the input tile any more
        if (more) store_tile();
        // ---- 3x3 / stride-2 max over the tile: thread -> (pooled pixel, 8 channels); window cells outside the map are skipped
        // (the centre cell is always inside)
        for (int e = tid; e < SP_TPH * SP_TPW * 8; e += 256) {
            const int cg = e & 7, pp = e >> 3;
            const int pyl = pp / SP_TPW, pxl = pp - pyl * SP_TPW;
            const int py = py0 + pyl, px = px0 + pxl;
            sp_u16x8 m8 = *reinterpret_cast<const sp_u16x8*>(cv_t + ((2 * pyl + 1) * SP_CW + 2 * pxl + 1) * 64 + cg * 8);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t == 4) continue;
                const int cy = cy0 + 2 * pyl + t / 3, cx = cx0 + 2 * pxl + t % 3;
                const sp_u16x8 v8 = *reinterpret_cast<const sp_u16x8*>(cv_t + ((2 * pyl + t / 3) * SP_CW + 2 * pxl + t % 3) * 64 + cg * 8);
                if ((unsigned)cy < (unsigned)IH && (unsigned)cx < (unsigned)IW) m8 = __builtin_elementwise_max(m8, v8);
            }
            if (py < PHo && px < PWo)
                APSE_NT_STORE(m8, reinterpret_cast<sp_u16x8*>(y + (((size_t)b * PHo + py) * PWo + px) * 64 + cg * 8));
        }
    }
}

extern "C" int apse_k_stem_pool16(const void* x, const uint16_t* w16, const float* bias, void* y, int B, int IH, int IW, int prec,
                                  hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (prec != 1 && prec != 2) return APSE_E_INVALID;
    const int PHo = (IH + 2 - 3) / 2 + 1, PWo = (IW + 2 - 3) / 2 + 1;
    const int tiles_y = (PHo + SP_TPH - 1) / SP_TPH, tiles_x = (PWo + SP_TPW - 1) / SP_TPW;
    const long tiles = (long)B * tiles_y * tiles_x;
    int blocks = tiles < 256 * SP_OCC ? (int)tiles : 256 * SP_OCC;
    if (ev0) hipEventRecord(ev0, s);
    if (prec == 1)
        hipLaunchKernelGGL(stem_s2d_pool16<1>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const uint16_t*>(x), w16, bias,
                           reinterpret_cast<uint16_t*>(y), B, IH, IW, PHo, PWo, tiles_y, tiles_x);
    else
        hipLaunchKernelGGL(stem_s2d_pool16<2>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const uint16_t*>(x), w16, bias,
                           reinterpret_cast<uint16_t*>(y), B, IH, IW, PHo, PWo, tiles_y, tiles_x);
    if (ev1) hipEventRecord(ev1, s);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
