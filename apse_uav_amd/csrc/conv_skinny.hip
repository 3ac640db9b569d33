// conv_skinny16 (gfx950): 1x1 convolutions with at most 16 output channels over K = 256 inputs -- the RPN head (3 objectness
// + 12 deltas per cell, one launch over the rows of all five levels) and the mask predictor (4 class logits per mask pixel)
// (detectron2 StandardRPNHead.objectness_logits / anchor_deltas and mask_rcnn_conv_upsample_head.predictor, reached from
// /root/reference/dcnn/networks/track_rcnn.py:44-51).
//
// These layers read a wide activation row (1 KB in f32) to produce 64 bytes: pure HBM streams (RPN head at 4K: 88 MB in,
// 5.5 MB out per frame).  On the tiled kernel they ran as 128x32 tiles -- half of every MFMA on padding, an LDS round trip
// and two barriers per 64-deep step -- at 2.4 TB/s (f32 batch 1) / 1.9 TB/s (fp16 batch 8, converting while staging).
// Here a wave keeps the whole 16 x 256 filter in registers (64 per lane) and walks 16-row tiles: the activations go
// global -> registers -> v_mfma_f32_16x16x4_f32 directly (lane = (row, k slot): one 16-byte load covers 4 k of f32 or 8 k
// of a 16-bit map; the four slots of a row read one contiguous 64-byte sector), several tiles in flight per wave, no LDS,
// no barrier.  Exact f32 arithmetic on f32 filters in every mode (decision layers: their logits feed the top-k / sigmoid
// thresholds), one accumulator per output, fixed k order.
#include "apse_common.h"
#include <stdlib.h>

// XT: storage of x (0 f32, 1 bf16, 2 f16).  NT tiles of 16 rows in flight per wave.
template <int XT>
__global__ __launch_bounds__(256, 2) void conv_skinny16(const ConvParams p) {
    constexpr int NT = XT ? 4 : 2;
    constexpr int KPL = XT ? 8 : 4;               // k per 16-byte load
    constexpr int NL = 256 / (4 * KPL);           // loads per row and lane (16 / 8)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, ks = lane >> 4;
    // filter fragments: MFMA (j, i) multiplies k = 4 KPL j + KPL ks + i; rows >= Cout of the packed filter are zero
    float bw[64];
#pragma unroll
    for (int j = 0; j < NL; ++j)
#pragma unroll
        for (int i = 0; i < KPL; ++i) bw[j * KPL + i] = p.w[(size_t)col * 256 + 4 * KPL * j + KPL * ks + i];
    const float bi = p.bias ? p.bias[col] : 0.f;
    int M = p.M;
    if (p.m_count) {
        const long live = (long)(*p.m_count) * p.m_per_item;
        M = live < M ? (int)live : M;
    }
    const int groups = (M + 16 * NT - 1) / (16 * NT);
    for (int g = blockIdx.x * 4 + wave; g < groups; g += gridDim.x * 4) {
        const int m0 = g * 16 * NT;
        f32x4 a[NT][NL];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int row = m0 + 16 * t + col;
            const bool in = row < M;
            const size_t e = (size_t)(in ? row : 0) * 256 + KPL * ks;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                f32x4 v;
                if constexpr (XT == 0) v = *reinterpret_cast<const f32x4*>(p.x + e + 4 * KPL * j);
                else v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const uint16_t*>(p.x) + e + 4 * KPL * j);
                a[t][j] = in ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            if constexpr (XT == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][j][i], bw[j * 4 + i], acc[t], 0, 0, 0);
            } else {
                f32x8 v[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) v[t] = apse_cvt8(__builtin_bit_cast(uint4, a[t][j]), XT);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[t][i], bw[j * 8 + i], acc[t], 0, 0, 0);
            }
        }
        // lane holds rows 4 ks + r (r = 0..3) of column col
        if (col < p.Cout) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 16 * t + 4 * ks + r;
                    float v = acc[t][r] + bi;
                    if (p.relu) v = apse_relu(v);
                    if (m < M) p.y[(size_t)m * p.y_ld + p.y_coff + col] = v;
                }
        }
    }
}

// Eligible: f32 filters and f32 output, 1x1 / stride 1 / pad 0 over exactly 256 input channels, at most 16 output channels,
// no residual, unsplit, plain NHWC output; x in any storage type; count-limited launches are handled (rows past the live count
// are neither read nor written).  A property of the layer only (never of M).
bool apse_conv_skinny_ok(const ConvParams& p) {
    static const bool off = getenv("APSE_NO_SKINNY") != nullptr;          // A/B switch
    if (off || p.prec != 0 || p.w == nullptr || p.y_st != 0) return false;
    if (p.KH != 1 || p.KW != 1 || p.pad != 0 || p.stride != 1) return false;
    if (p.cin_log2 != 8 || p.KWCp != 256 || p.Cout > 16 || p.Cout < 1) return false;
    if (p.res_mode != 0 || p.out_mode != 0 || p.splitk != 1 || p.tile_cnt) return false;
    if (p.x_st < 0 || p.x_st > 2) return false;
    return true;
}

int apse_launch_conv_skinny(const ConvParams& p, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (!apse_conv_skinny_ok(p)) return APSE_E_INVALID;
    const int nt = p.x_st ? 4 : 2;
    const long groups = ((long)p.M + 16 * nt - 1) / (16 * nt);
    long blocks = (groups + 3) / 4;
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);          // persistent waves: the filter fragments are loaded once
    if (ev0) hipEventRecord(ev0, s);
    if (p.x_st == 0) hipLaunchKernelGGL(conv_skinny16<0>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    else if (p.x_st == 1) hipLaunchKernelGGL(conv_skinny16<1>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv_skinny16<2>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    if (ev1) hipEventRecord(ev1, s);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
