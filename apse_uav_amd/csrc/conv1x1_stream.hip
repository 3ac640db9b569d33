// Memory-streaming 1x1 convolution for gfx950 (CDNA4): the "expansion" layers of the bottlenecks and the FPN laterals
// (/root/reference/dcnn/networks/track_rcnn.py:42 -> detectron2 BottleneckBlock.conv3 / shortcut, FPN lateral convs).
//
// These layers have a tiny K (64 / 128 / 256 input channels) and a wide N: per output row they read K elements, read N
// residual elements and write N elements -- 4..9 bytes of HBM traffic per MAC-pair, far on the memory side of the roofline
// (res4 conv3 in fp16 at batch 8: 148 MB per launch against 17 GFLOP).  The tiled implicit-GEMM kernel (conv_igemm.hip) spends
// such a launch in per-tile prologue -> 4 k-steps -> C-tile shuffle -> residual -> store sequences with little in flight
// (measured 1.8 TB/s).  This kernel is built for the traffic instead:
//   * one block = 4 waves = 128 output rows x ALL (or a slice of the) N columns: every activation element is used by exactly one
//     wave, so the A operand goes global -> registers directly, once (K/16 x 16 B per lane, the whole K stays in VGPRs) --
//     no LDS, no barrier, no re-read;
//   * the filters (N x K, L2-resident) stream through a double-buffered, XOR-swizzled LDS stage of 128 columns x 128 bytes
//     shared by the four waves, fetched two stages ahead through registers;
//   * N is walked in chunks of 128 columns: the residual rows of a chunk are requested while its MFMAs run, the finished
//     chunk goes through a wave-private LDS transpose (no block barrier) and leaves as whole 128-byte line segments
//     (16 B per lane: bias + residual + ReLU + rounding fused), i.e. stores and residual loads of one chunk are in flight while
//     the next chunk computes.  Two blocks per CU (launch bounds) interleave their phases.
// Arithmetic: the same MFMA instructions and the same sequential k order per accumulator as conv_igemm's unsplit
// single-k-group shapes (exact f32 fma chain for PR = 0), so f32 results are bit-identical to that kernel's.
#include "apse_common.h"
#include <stdlib.h>
#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned s1_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 s1_zero16() { return f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f32x16 s1_mfma16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 s1_mfma16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// PR: 0 = f32 operands (x / filters f32), 1 = bf16, 2 = f16 (x stored in that type, filters pre-rounded in w16).
// KA: the layer's K (input channels): 64, 128 or 256 -- the whole A strip of a wave lives in registers.
// NC: columns per chunk, 128 or 64 (64: half the accumulators -- f32 with K = 128 then fits two blocks per CU; round 3).
template <int PR, int KA, int NC = 128>
__global__ __launch_bounds__(256, 2) void conv1x1_stream(const ConvParams p, const int chunks_per_block) {
    constexpr int ESH = PR ? 1 : 2;                 // log2(bytes per element)
    constexpr int KSTEP = PR ? 64 : 32;             // elements per 128-byte stage row
    constexpr int EPSLOT = 16 >> ESH;               // elements per 16-byte slot
    constexpr int NST = KA / KSTEP;                 // k stages per chunk
    constexpr int NJ = NC / 32;                     // 32-column MFMA tiles per chunk
    constexpr int CLD = 68;                         // row stride (floats) of the wave-private transpose tile [32][64]
    static_assert(NST >= 1 && KA % KSTEP == 0, "K must be a whole number of stages");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ws = smem;                                            // [2][NC][128 B]
    float* Cw = reinterpret_cast<float*>(smem + 2 * NC * 128);  // [4 waves][32][CLD]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int srow = tid >> 3, slot = tid & 7;                  // filter staging: 32 rows per pass, 8 slots of 16 B
    const int M = p.M, N = p.Cout;
    const int m0 = blockIdx.x * 128 + wave * 32;
    const int chunks_total = N / NC;
    const int c_begin = blockIdx.y * chunks_per_block;
    const int c_end = (c_begin + chunks_per_block < chunks_total) ? c_begin + chunks_per_block : chunks_total;
    if (c_begin >= c_end) return;
    const int ohw = p.OH * p.OW;

    // ---- A strip: row m0 + fr, all K, straight into registers (zeros past M through the descriptor's range check)
    __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0,
                                                                    (int)(((unsigned)(p.B * p.H * p.W) << p.cin_log2) << ESH), 0x00020000);
    f32x4 a[NST][4];
    {
        const int m = m0 + fr;
        unsigned base = 0xfffffff0u;
        if (m < M) {
            int pix = m;
            if (p.stride != 1) {
                const int b = m / ohw, rem = m - b * ohw, oy = rem / p.OW, ox = rem - oy * p.OW;
                pix = (b * p.H + oy * p.stride) * p.W + ox * p.stride;
            }
            base = ((unsigned)pix << p.cin_log2) << ESH;
        }
#pragma unroll
        for (int s = 0; s < NST; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned off = (m < M) ? base + (unsigned)((s * KSTEP + (2 * c + fh) * EPSLOT) << ESH) : 0xfffffff0u;
                a[s][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off, 0, 0));
            }
    }

    // ---- filter stages: global stage g = (chunk - c_begin) * NST + s
    const char* wbase = (PR ? reinterpret_cast<const char*>(p.w16) : reinterpret_cast<const char*>(p.w)) + ((size_t)(slot * EPSLOT) << ESH);
    const int G = (c_end - c_begin) * NST;
    f32x4 wr[NJ];
    auto w_fetch = [&](int g) {
        const int ch = c_begin + g / NST, s = g - (g / NST) * NST;
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const size_t n = (size_t)ch * NC + srow + 32 * i;
            wr[i] = *reinterpret_cast<const f32x4*>(wbase + ((n * (size_t)p.KWCp + (size_t)s * KSTEP) << ESH));
        }
    };
    auto w_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const int row = srow + 32 * i;
            *reinterpret_cast<f32x4*>(Ws + buf * NC * 128 + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4)) = wr[i];
        }
    };
    w_fetch(0);
    w_store(0);
    if (G > 1) w_fetch(1);
    __syncthreads();

    float* cw = Cw + wave * 32 * CLD;
    const int erow = lane >> 3, ecol = (lane & 7) * 8;          // epilogue: lane -> (row erow + 8 i, columns ecol .. ecol + 7)
    // Output and residual rows go through buffer descriptors: byte offsets of this lane's four rows once per block (rows past M get
    // an offset beyond the range: the load returns zeros, the store is dropped) -- no branch and no 64-bit arithmetic per chunk.
    constexpr unsigned ES = PR ? 2u : 4u, DEAD = 0x80000000u;      // (apse_conv1x1_stream_ok keeps every tensor below 0x70000000 bytes)
    unsigned yrow[4], rrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + erow + 8 * i;
        yrow[i] = m < M ? (unsigned)m * (unsigned)N * ES : DEAD;
        rrow[i] = yrow[i];
        if (p.res_mode == 2 && m < M) {
            const int b = m / ohw, rem = m - b * ohw, oy = rem / p.OW, ox = rem - oy * p.OW;
            rrow[i] = (unsigned)((b * (p.OH >> 1) + (oy >> 1)) * (p.OW >> 1) + (ox >> 1)) * (unsigned)N * ES;
        }
    }
    const unsigned res_rows = p.res_mode == 2 ? (unsigned)(p.B * (p.OH >> 1) * (p.OW >> 1)) : (unsigned)M;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((unsigned)M * (unsigned)N * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.y), 0,
                                                                           (int)(p.res ? res_rows * (unsigned)N * ES : 0u), 0x00020000);
    int g = 0;
    for (int ch = c_begin; ch < c_end; ++ch) {
        f32x16 acc[NJ];                 // written by the chunk's first MFMAs (C operand = the constant 0)
#pragma unroll
        for (int s = 0; s < NST; ++s, ++g) {
            const int buf = g & 1;
            // stage g + 1 (already in registers) -> the other LDS buffer (last read in stage g - 1, a barrier ago);
            // stage g + 2 -> registers
            if (g + 1 < G) w_store(buf ^ 1);
            if (g + 2 < G) w_fetch(g + 2);
            const char* Wb = Ws + buf * NC * 128;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 bfr[NJ];
                const int ls = 2 * c + fh;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int row = j * 32 + fr;
                    bfr[j] = *reinterpret_cast<const f32x4*>(Wb + row * 128 + ((ls ^ ((row >> 1) & 7)) << 4));
                }
                if constexpr (PR != 0) {
                    typedef typename std::conditional<PR == 1, bf16x8, f16x8>::type op8;
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[j] = s1_mfma16(__builtin_bit_cast(op8, a[s][c]), __builtin_bit_cast(op8, bfr[j]), (s == 0 && c == 0) ? s1_zero16() : acc[j]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][c][k], bfr[j][k], (s == 0 && c == 0 && k == 0) ? s1_zero16() : acc[j], 0, 0, 0);
                }
            }
            __syncthreads();
        }
        // ---- chunk epilogue, wave-private: two halves of 64 columns through the transpose tile
        const int n0 = ch * NC;
#pragma unroll
        for (int hh = 0; hh < NC / 64; ++hh) {
            const int nb = n0 + hh * 64 + ecol;
            // residual rows of this half: requested before the accumulators go through LDS
            constexpr int RV = PR ? 1 : 2;              // 16-byte vectors per lane and row: 8 x 16 bit, or 2 x 4 f32
            f32x4 rr[4][RV];
            if (p.res_mode != 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int q = 0; q < RV; ++q)
                        rr[i][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, (int)(rrow[i] + (unsigned)nb * ES + q * 16u), 0, 0));
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    cw[((v & 3) + 8 * (v >> 2) + 4 * fh) * CLD + jj * 32 + fr] = acc[2 * hh + jj][v];
            __builtin_amdgcn_wave_barrier();
            f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
            if (p.bias) {
                b0 = *reinterpret_cast<const f32x4*>(p.bias + nb);
                b1 = *reinterpret_cast<const f32x4*>(p.bias + nb + 4);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = erow + 8 * i;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(cw + r * CLD + ecol) + b0;
                f32x4 v1 = *reinterpret_cast<const f32x4*>(cw + r * CLD + ecol + 4) + b1;
                if (p.res_mode != 0) {
                    if constexpr (PR == 0) { v0 += rr[i][0]; v1 += rr[i][RV - 1]; }
                    else {
                        const f32x4 raw = rr[i][0];
                        f32x4 x0, x1;
                        if constexpr (PR == 1) {
                            const unsigned q0 = __float_as_uint(raw[0]), q1 = __float_as_uint(raw[1]), q2 = __float_as_uint(raw[2]), q3 = __float_as_uint(raw[3]);
                            x0 = f32x4{__uint_as_float(q0 << 16), __uint_as_float(q0 & 0xffff0000u), __uint_as_float(q1 << 16), __uint_as_float(q1 & 0xffff0000u)};
                            x1 = f32x4{__uint_as_float(q2 << 16), __uint_as_float(q2 & 0xffff0000u), __uint_as_float(q3 << 16), __uint_as_float(q3 & 0xffff0000u)};
                        } else {
                            const f16x8 hv = __builtin_bit_cast(f16x8, raw);
                            x0 = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
                            x1 = f32x4{(float)hv[4], (float)hv[5], (float)hv[6], (float)hv[7]};
                        }
                        v0 += x0; v1 += x1;
                    }
                }
                if (p.relu) {
                    // x > 0 ? x : 0 as ONE v_max_f32 with the constant first (-0 and NaN give +0, like the select).  Written in C the
                    // compiler puts a canonicalising v_max x, x in front of it (IEEE mode): 128 instead of 64 instructions per chunk
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        asm("v_max_f32 %0, 0, %1" : "=v"(v0[k]) : "v"(v0[k]));
                        asm("v_max_f32 %0, 0, %1" : "=v"(v1[k]) : "v"(v1[k]));
                    }
                }
                const int yoff = (int)(yrow[i] + (unsigned)nb * ES);
                if constexpr (PR == 0) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(s1_u32x4, v0), yrsrc, yoff, 0, APSE_NT ? 2 : 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(s1_u32x4, v1), yrsrc, yoff + 16, 0, APSE_NT ? 2 : 0);
                } else if constexpr (PR == 1) {
                    bf16x8 o;
                    o[0] = (__bf16)v0[0]; o[1] = (__bf16)v0[1]; o[2] = (__bf16)v0[2]; o[3] = (__bf16)v0[3];
                    o[4] = (__bf16)v1[0]; o[5] = (__bf16)v1[1]; o[6] = (__bf16)v1[2]; o[7] = (__bf16)v1[3];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(s1_u32x4, o), yrsrc, yoff, 0, APSE_NT ? 2 : 0);
                } else {
                    f16x8 o;
                    o[0] = (_Float16)v0[0]; o[1] = (_Float16)v0[1]; o[2] = (_Float16)v0[2]; o[3] = (_Float16)v0[3];
                    o[4] = (_Float16)v1[0]; o[5] = (_Float16)v1[1]; o[6] = (_Float16)v1[2]; o[7] = (_Float16)v1[3];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(s1_u32x4, o), yrsrc, yoff, 0, APSE_NT ? 2 : 0);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ------------------------------------------------------------------------------------------------ dispatch
// Eligible: 1x1, pad 0, stride 1 or 2, NHWC rows of exactly K = 64 / 128 / 256 channels, N a multiple of 128, plain NHWC
// output (no deconv scatter, no channel offset), unsplit, not count-limited.  16-bit: activations stored in the operand
// type and pre-rounded filters; f32: f32 everywhere.  The choice depends on the layer only, never on M (a frame's results
// do not change with the batch it runs in).
bool apse_conv1x1_stream_ok(const ConvParams& p) {
    static const bool off = getenv("APSE_NO_STREAM1X1") != nullptr;           // A/B switch for the sweeps
    if (off || p.no_stream) return false;
    const int K = 1 << p.cin_log2;
    if (p.KH != 1 || p.KW != 1 || p.pad != 0 || (p.stride != 1 && p.stride != 2)) return false;
    if (K != 64 && K != 128 && K != 256) return false;
    const bool nc64 = p.prec == 0 && K != 64;                     // the 64-column-chunk instantiations (f32, K = 128 / 256)
    if (p.KWCp != K || (p.Cout & (nc64 ? 63 : 127)) != 0) return false;
    if (p.out_mode != 0 || p.y_coff != 0 || p.y_ld != p.Cout || p.splitk != 1 || p.m_count || p.tile_cnt) return false;
    if (p.res_mode != 0 && p.res_mode != 1 && p.res_mode != 2) return false;
    if ((((size_t)p.B * p.H * p.W) << p.cin_log2) * (p.prec ? 2 : 4) >= 0xfffffff0ull) return false;
    if (((size_t)p.M + 128) * (size_t)p.Cout * (p.prec ? 2 : 4) >= 0x70000000ull) return false;       // output / residual rows are addressed with 32-bit byte offsets
    // residual and output live in the operand's storage type (f32 mode: f32; 16-bit modes with 16-bit storage: that type)
    if (p.y_st != p.prec || (p.res_mode != 0 && p.res_st != p.prec)) return false;
    // f32: K = 64; (round 3) K = 128 and K = 256 with 64-column chunks -- half the accumulators, so the A strip (64 / 128 registers)
    // still leaves two blocks per CU (190 / 254 VGPRs).  Isolated sweep (profiles/r03_sweep_f32_stream.txt): res3 conv3 (K = 128)
    // 26.5 us against 33.0 tiled; lateral 2 (K = 256, 64 512 rows) 81 against 92; res4 conv3 (K = 256, 4 032 rows: one chunk per
    // block) and the stride-2 res3 shortcut measure equal to the tiled kernel, so K = 256 is taken for large maps only.  Same bits
    // as the tiled kernel either way.
    if (p.prec == 0) {
        static const bool no128 = getenv("APSE_NO_STREAM_F32K128") != nullptr, no256 = getenv("APSE_NO_STREAM_F32K256") != nullptr;
        return p.x_st == 0 && p.w != nullptr &&
               (K == 64 || (K == 128 && !no128) || (K == 256 && !no256 && p.stride == 1 && (size_t)p.OH * p.OW >= 32768));
    }
    return p.x_st == p.prec && p.w16 != nullptr;
}

template <int PR, int KA, int NC = 128>
static int launch_stream(const ConvParams& p, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    const size_t lds = 2 * NC * 128 + 4 * 32 * 68 * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_stream<PR, KA, NC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int mblocks = (p.M + 127) / 128, chunks = p.Cout / NC;
    // at least one block per CU (then 256..511 blocks: measured best for res4 conv3, 37 us against 42 / 50 with 2x / 4x as
    // many): split the N chunks over blockIdx.y when the row blocks alone do not fill the chip
    static const int want = getenv("APSE_STREAM_BLOCKS") ? atoi(getenv("APSE_STREAM_BLOCKS")) : 256;
    int ysplit = 1;
    while (mblocks * ysplit < want && ysplit < chunks) ysplit *= 2;
    if (ysplit > chunks) ysplit = chunks;
    const int per = (chunks + ysplit - 1) / ysplit;
    if (ev0) hipEventRecord(ev0, s);
    hipLaunchKernelGGL((conv1x1_stream<PR, KA, NC>), dim3(mblocks, (chunks + per - 1) / per), dim3(256), lds, s, p, per);
    if (ev1) hipEventRecord(ev1, s);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}

int apse_launch_conv1x1_stream(const ConvParams& p, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    const int K = 1 << p.cin_log2;
    if (p.prec == 0) return K == 64 ? launch_stream<0, 64>(p, s, ev0, ev1) : (K == 128 ? launch_stream<0, 128, 64>(p, s, ev0, ev1) : launch_stream<0, 256, 64>(p, s, ev0, ev1));
    if (p.prec == 1) {
        if (K == 64) return launch_stream<1, 64>(p, s, ev0, ev1);
        if (K == 128) return launch_stream<1, 128>(p, s, ev0, ev1);
        return launch_stream<1, 256>(p, s, ev0, ev1);
    }
    if (K == 64) return launch_stream<2, 64>(p, s, ev0, ev1);
    if (K == 128) return launch_stream<2, 128>(p, s, ev0, ev1);
    return launch_stream<2, 256>(p, s, ev0, ev1);
}
