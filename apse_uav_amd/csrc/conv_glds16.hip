// Implicit-GEMM convolution for 16-bit operands with a three-stage LDS ring filled by LDS-DMA (gfx950 / CDNA4).
//
// The deep-K layers of the 16-bit modes (3x3 convolutions of res3..res5, the FPN output and RPN convolutions, fc1:
// /root/reference/dcnn/networks/track_rcnn.py:42-51 through detectron2) spend their time in conv_igemm_f32<4,2,2,2,1,0,1,PR>
// (256x128 tile).  There every operand byte goes global -> VGPR -> ds_write_b128 -> LDS: 48 KB of LDS writes per 64-deep
// k-step at ~79 B/clk (MI355X_MICROARCH.md, LDS table) is ~620 LDS cycles beside ~510 cycles of fragment reads, against the
// 1024 matrix-pipe cycles of the step -- the LDS port and the one-step-ahead register prefetch, not the MFMAs, pace the loop
// (measured 800-900 TFLOP/s, MFMA busy 30 %).  This kernel keeps tile shape, XOR swizzle, fragment reads and MFMA order of
// that kernel (bit-identical results) and changes the feed:
//   * operands go global -> LDS directly (`buffer_load_dwordx4 ... lds`): no staging registers, no ds_write, 6 wave
//     instructions per wave and stage; the buffer descriptor's range check supplies the zeros of padding taps / rows past M;
//   * the LDS image is lane-linear per wave instruction (8 rows x 128 B), so the swizzle is applied on the SOURCE side: lane
//     (row r, slot s) fetches chunk s ^ swz(r) (cdna_hip_programming.md rule 21: linear destination, swizzled source, swizzled read);
//   * three stages of 48 KB: stage t + 2 is issued right after the barrier that opens stage t, so a fetch has two whole
//     k-steps (>= 2000 cycles) to land; per stage one `s_waitcnt vmcnt(6)` (this wave's 6 DMAs of the NEXT stage may stay
//     in flight) + one raw `s_barrier` -- never `__syncthreads()`, whose fence would drain the DMAs (`vmcnt(0)`).
#include "apse_common.h"
#include <stdlib.h>
#include <type_traits>

typedef _Float16 g16_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 g16_mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 g16_mfma(g16_f16x8 a, g16_f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

#ifndef GL_ABLATE
#define GL_ABLATE 0      // diagnostics only (tools/gpu_glds_ablate.sh): 1 no MFMA, 2 DMAs of the first two stages only, 3 no fragment reads, 4 two of three A fetches out of range, 5 no output stores, 6 two stages only
#endif
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned g16_u32x2 __attribute__((ext_vector_type(2)));

// BM x BN = 256 x 128: 8 waves as 4 x 2, 64 x 64 per wave, three stages of 48 KB.  256 x 256: 8 waves as 2 x 4, 128 x 64 per wave,
// two stages of 64 KB -- per matrix-pipe cycle a third fewer bytes through L2 -> LDS (32 instead of 47 B/clk/CU, of ~64 the port
// gives) and a quarter fewer fragment bytes out of LDS; the C tile (256 x 256 f32 does not fit) leaves in two halves of 128
// columns.  128 x 128 (round 4): 8 waves as 4 x 2, 32 x 64 per wave, three stages of 32 KB -- for the layers whose M gives about
// one 128 x 128 tile per CU but only half a wave of 256-row tiles (res4 at batch 4: M = 16 128, N = 256 -> 252 tiles).
//
// Built, measured and dropped (round 4): a ping-pong schedule -- the eight waves as two groups (one wave of each per SIMD) in
// opposite phase, a block of fragment reads / a barrier / the block's MFMAs back to back with s_setprio 1 / a barrier, group 1 one
// slot behind group 0, counted vmcnt at the stage seams (bit-identical, race-screened by the equality tests).  On one box, old
// loop -> ping-pong, us per launch: 256x256 300 -> 325 (fp16 batch 8) and 320 -> 355 (bf16 batch 4); 256x128 42.7 -> 51.7 and
// 37.0 -> 44.5; 128x128 25.1 -> 26.2 (profiles/r04b_kstats_pingpong.txt).  With 8 or 16 MFMAs per block the four barriers per
// stage and the exposed ds_read latency in front of each MFMA block cost more than the forced alternation gains: two free-running
// waves per SIMD already overlap one wave's reads with the other's MFMAs.
template <int PR, int GL_BM, int GL_BN>
__global__ __launch_bounds__(512) void conv_glds16(const ConvParams p) {
    typedef typename std::conditional<PR == 1, bf16x8, g16_f16x8>::type op8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int GL_STAGE = (GL_BM + GL_BN) * 128;   // bytes per stage: 64 elements (128 B) per tile row
    constexpr int GL_NS = (GL_BM + GL_BN) * 128 * 3 <= 160 * 1024 ? 3 : 2;      // three stages where they fit (all but 256 x 256)
    constexpr int WNW = GL_BN / 64;                   // waves along N (64 columns each)
    constexpr int TMW = (GL_BM / (8 / WNW)) / 32;     // 32-row MFMA tiles per wave along M: 1 (128 x 128), 2 (256 x 128) or 4 (256 x 256)
    constexpr int APW = GL_BM / 64, BPW = GL_BN / 64; // DMA pieces (8-row groups) per wave and stage: A 2 or 4, B 2 or 4
    constexpr int NPW = APW + BPW;
    constexpr int LDC = 128 + 4;
    float* Cs = reinterpret_cast<float*>(smem);                     // epilogue view [256][LDC] (aliases the ring)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNW, wn = wave % WNW;
    const int fr = lane & 31, fh = lane >> 5;
    const int M = p.M;
    const int tiles_n = (p.Cout + GL_BN - 1) / GL_BN;
    const int tiles_m = (M + GL_BM - 1) / GL_BM;
    const int nwg = tiles_m * tiles_n;
    int bid;
    {   // XCD-aware bijective remap (blocks are dealt round-robin over the 8 XCDs): contiguous tile ranges per XCD
        const int wg = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * GL_BM, n0 = tile_n * GL_BN;
    const int ohw = p.OH * p.OW;
    const int steps_per_row = p.KWCp >> 6;
    const int T = (GL_ABLATE == 6) ? 2 : p.KH * steps_per_row;      // 6: two stages only = the fixed per-tile cost
    const size_t w_row = (size_t)p.KH * p.KWCp;

    // ---- this lane's DMA rows: A groups 4 wave .. 4 wave + 3, B groups 2 wave, 2 wave + 1 (a group = 8 rows x 128 B = one
    // wave instruction); LDS slot (row, lane & 7) receives chunk cs = (lane & 7) ^ swz(row)
    // Per lane and A piece: ONE byte offset (the row's centre pixel + the lane's swizzled chunk) and a bit mask of the filter taps whose
    // input pixel lies inside the map.  Everything that changes from stage to stage is wave-uniform -- the tap's displacement and the
    // channel slice -- and rides in the instruction's scalar offset; a masked tap (zero padding) or a row past M turns the vector
    // offset into 0xffffffff, which the descriptor's range check answers with zeros.  3 vector instructions per piece and stage
    // (round 4; the earlier form recomputed pixel coordinates and two range tests per piece: ~9).  The descriptor's base is moved
    // back by the displacement of tap (0, 0), so that every tap's scalar offset is >= 0; no access ever goes below p.x (such taps
    // are masked).
    unsigned a_voff[APW], a_taps[APW];
    const unsigned xbytes = (((unsigned)(p.B * p.H * p.W)) << p.cin_log2) << 1;
    const unsigned xback = (((unsigned)(p.pad * p.W + p.pad)) << p.cin_log2) << 1;
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int row = (wave * APW + i) * 8 + (lane >> 3);
        const int m = m0 + row;
        const int cs = ((lane & 7) ^ ((row >> 1) & 7)) << 3;       // element offset of the chunk inside the 64-element step
        a_voff[i] = 0u; a_taps[i] = 0u;
        if (m < M) {
            const int b = m / ohw, rem = m - b * ohw, oy = rem / p.OW, ox = rem - oy * p.OW;
            const int cy = oy * p.stride, cx = ox * p.stride;                    // centre pixel: tap (pad, pad)
            a_voff[i] = ((((unsigned)((b * p.H + cy) * p.W + cx)) << p.cin_log2) + (unsigned)cs) << 1;
            unsigned tm = 0xffffffffu;
            if (p.pad > 0) {
                tm = 0u;
                for (int ky = 0; ky < p.KH; ++ky)
                    for (int kx = 0; kx < p.KW; ++kx)
                        if ((unsigned)(cy - p.pad + ky) < (unsigned)p.H && (unsigned)(cx - p.pad + kx) < (unsigned)p.W) tm |= 1u << (ky * p.KW + kx);
            }
            a_taps[i] = tm;
        }
    }
    unsigned b_off[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int row = (wave * BPW + i) * 8 + (lane >> 3);
        const int cs = ((lane & 7) ^ ((row >> 1) & 7)) << 3;
        b_off[i] = (unsigned)((((size_t)(n0 + row) * w_row) + cs) << 1);       // bytes; filters are padded to Cout_p = 128 k rows
    }
    __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x) - xback), 0, (int)(xbytes + xback), 0x00020000);
    __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w16), 0,
                                                                    (int)((((unsigned)(tiles_n * GL_BN)) * (unsigned)w_row) << 1), 0x00020000);
    int ld_r = 0, ld_q = 0;
    // One DMA "piece" = one wave instruction (8 rows x 128 B).  Pieces 0..APW-1: this wave's A groups, APW..: its B groups.
    // piece 0 also computes the step's scalars and advances (filter row, step) -- pieces of a stage are issued in order.
    int cur_tap = 0;
    unsigned cur_xoff = 0, cur_woff = 0;
    auto issue_piece = [&](int buf, int j) {
        if (j == 0) {
            const int q0 = ld_q << 6;                                // element offset of this step inside the filter row's run: kx * Cin + channel slice
            const int kx = q0 >> p.cin_log2;
            cur_tap = ld_r * p.KW + kx;
            cur_xoff = (unsigned)((((ld_r * p.W + kx) << p.cin_log2) + (q0 & ((1 << p.cin_log2) - 1))) << 1);
            cur_woff = (unsigned)((ld_r * p.KWCp + q0) << 1);
            if (++ld_q == steps_per_row) { ld_q = 0; ++ld_r; }
        }
        if (j < APW) {
            unsigned off = a_voff[j] | (((a_taps[j] >> (cur_tap & 31)) & 1u) - 1u);                       // tap inside the map: the offset; else all ones
            if (GL_ABLATE == 4 && (ld_q % 3) != 0) off = 0xffffffffu;      // what if two of three A fetches were free (tap reuse)?
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(smem + buf * GL_STAGE + wave * (APW * 1024) + j * 1024), 16, (int)off, (int)cur_xoff, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(smem + buf * GL_STAGE + GL_BM * 128 + wave * (BPW * 1024) + (j - APW) * 1024), 16,
                                                     (int)b_off[j - APW], (int)cur_woff, 0, 0);
        }
    };
    auto issue = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NPW; ++j) issue_piece(buf, j);
    };

    f32x16 acc[TMW][2];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    constexpr int AHEAD = GL_NS - 1;                  // stages in flight beyond the one being computed
    issue(0);
    if (AHEAD > 1 && T > 1) issue(1);
    for (int t = 0; t < T; ++t) {
        // stage t has landed once this wave's DMAs of the later stages are the only ones outstanding; the barrier then vouches for
        // every wave's share and for the end of all reads of stage t - 1, whose buffer stage t + AHEAD is about to overwrite
        if (GL_ABLATE == 2 || AHEAD == 1 || t + 1 >= T) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");   // three stages: the NPW pieces of stage t + 1 may stay in flight
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const bool more = (t + AHEAD < T) && (GL_ABLATE != 2);
        const int nbuf = (t + AHEAD) % GL_NS;
        const char* As = smem + (t % GL_NS) * GL_STAGE;
        const char* Bs = As + GL_BM * 128;
        f32x4 af[2][TMW], bf[2][2];
        auto load_frags = [&](int c, int fb) {
            const int ls = 2 * c + fh;
            if (GL_ABLATE == 3 && t > 0) return;
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                const int row = (wm * TMW + i) * 32 + fr;
                af[fb][i] = *reinterpret_cast<const f32x4*>(As + row * 128 + ((ls ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = (wn * 2 + j) * 32 + fr;
                bf[fb][j] = *reinterpret_cast<const f32x4*>(Bs + row * 128 + ((ls ^ ((row >> 1) & 7)) << 4));
            }
        };
        load_frags(0, 0);
        // 4 TMW x 2 MFMAs per wave and stage; the DMA pieces of stage t + AHEAD ride in their gaps (one piece in front of every
        // second MFMA, pinned with sched_barriers) instead of in a burst behind the barrier
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c + 1 < 4) load_frags(c + 1, (c + 1) & 1);
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int g = (c * TMW + i) * 2 + j;
                    if ((g & 1) && (g >> 1) < NPW) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (more) issue_piece(nbuf, g >> 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (GL_ABLATE == 1) { asm volatile("" ::"v"(af[c & 1][i]), "v"(bf[c & 1][j])); continue; }
                    acc[i][j] = g16_mfma(__builtin_bit_cast(op8, af[c & 1][i]), __builtin_bit_cast(op8, bf[c & 1][j]), acc[i][j]);
                }
        }
    }
    asm volatile("s_barrier" ::: "memory");       // every wave is done with the ring: the C tile may overwrite it
    __builtin_amdgcn_sched_barrier(0);

    // -------------------------------------------------------------- epilogue (same arithmetic order as conv_igemm: C + bias, + residual, ReLU)
    // 128 columns at a time through the [256][132] f32 view: the waves holding that column half write, everybody stores
    constexpr int C4 = 32;                       // float4 chunks per 128-column row
    constexpr int RPP = 512 / C4;                // rows per pass
    const int c4 = tid % C4, r0 = tid / C4;
#pragma unroll
    for (int nh = 0; nh < GL_BN / 128; ++nh) {
        if (nh > 0) __syncthreads();             // the previous half has been read out
        if ((wn >> 1) == nh) {
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int row = (wm * TMW + i) * 32 + (v & 3) + 8 * (v >> 2) + 4 * fh;
                        Cs[row * LDC + ((wn & 1) * 2 + j) * 32 + fr] = acc[i][j][v];
                    }
        }
        __syncthreads();
        const int n = n0 + nh * 128 + c4 * 4;
        if (n < p.Cout) {
            f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
            const bool vec = ((p.y_ld & 3) == 0) && ((p.y_coff & 3) == 0) && (p.y_coff + n + 4 <= p.y_ld);
            // Fast path (every trunk layer): output and residual stored in the operand type, whole 8-byte groups.  Rows go through
            // buffer descriptors sized to M rows -- a row past M is out of range by construction (loads give zeros, stores are
            // dropped) -- with ONE per-thread byte offset and a uniform step per pass: no branch, no 64-bit arithmetic, no run-time
            // storage-type switch per element (the general form below costs ~2x the instructions of this one).
            const bool fast = vec && p.y_st == PR && (p.res_mode == 0 || (p.res_mode == 1 && p.res_st == PR && p.res)) &&
                              ((size_t)M + GL_BM) * (size_t)p.y_ld * 2 < 0xfffffff0ull && ((size_t)M + GL_BM) * (size_t)p.Cout * 2 < 0xfffffff0ull;
            if (fast) {
                const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((unsigned)M * (unsigned)p.y_ld * 2u), 0x00020000);
                const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res_mode ? p.res : p.y), 0,
                                                                                       (int)(p.res_mode ? (unsigned)M * (unsigned)p.Cout * 2u : 0u), 0x00020000);
                const unsigned ystep = (unsigned)(RPP * p.y_ld) * 2u, rstep = (unsigned)(RPP * p.Cout) * 2u;
                const unsigned yo = (unsigned)((m0 + r0) * p.y_ld + p.y_coff + n) * 2u, ro = (unsigned)((m0 + r0) * p.Cout + n) * 2u;
#pragma unroll
                for (int g0 = 0; g0 < GL_BM / RPP; g0 += 8) {
                    g16_u32x2 rg[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        rg[i] = p.res_mode ? __builtin_bit_cast(g16_u32x2, __builtin_amdgcn_raw_buffer_load_b64(rrsrc, (int)(ro + (unsigned)(g0 + i) * rstep), 0, 0))
                                           : g16_u32x2{0u, 0u};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int r = r0 + (g0 + i) * RPP;
                        f32x4 val = *reinterpret_cast<const f32x4*>(Cs + r * LDC + c4 * 4);
                        val += bias4;
                        f32x4 rv;
                        if constexpr (PR == 1) {
                            rv = f32x4{__uint_as_float(rg[i][0] << 16), __uint_as_float(rg[i][0] & 0xffff0000u), __uint_as_float(rg[i][1] << 16),
                                       __uint_as_float(rg[i][1] & 0xffff0000u)};
                        } else {
                            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                            const h4 hv = __builtin_bit_cast(h4, rg[i]);
                            rv = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
                        }
                        val += rv;
                        if (p.relu) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) asm("v_max_f32 %0, 0, %1" : "=v"(val[k]) : "v"(val[k]));      // x > 0 ? x : 0 (-0, NaN -> +0)
                        }
                        g16_u32x2 o;
                        if constexpr (PR == 1) {
                            typedef __bf16 b4 __attribute__((ext_vector_type(4)));
                            b4 t; t[0] = (__bf16)val[0]; t[1] = (__bf16)val[1]; t[2] = (__bf16)val[2]; t[3] = (__bf16)val[3];
                            o = __builtin_bit_cast(g16_u32x2, t);
                        } else {
                            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                            h4 t; t[0] = (_Float16)val[0]; t[1] = (_Float16)val[1]; t[2] = (_Float16)val[2]; t[3] = (_Float16)val[3];
                            o = __builtin_bit_cast(g16_u32x2, t);
                        }
                        __builtin_amdgcn_raw_buffer_store_b64(o, yrsrc, (int)(yo + (unsigned)(g0 + i) * ystep), 0, APSE_NT ? 2 : 0);
                    }
                }
                continue;
            }
            auto res_load = [&](int m) -> f32x4 {
                f32x4 rv = {0.f, 0.f, 0.f, 0.f};
                if (m < M) {
                    if (p.res_mode == 1) rv = apse_ld4(p.res, (size_t)m * p.Cout + n, p.res_st);
                    else if (p.res_mode == 2) {
                        const int b = m / ohw, rem = m - b * ohw, oy = rem / p.OW, ox = rem - oy * p.OW;
                        rv = apse_ld4(p.res, ((size_t)b * ((p.OH >> 1) * (p.OW >> 1)) + (oy >> 1) * (p.OW >> 1) + (ox >> 1)) * p.Cout + n, p.res_st);
                    }
                }
                return rv;
            };
#pragma unroll
            for (int g0 = 0; g0 < GL_BM / RPP; g0 += 8) {
                f32x4 rg[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) rg[i] = p.res_mode != 0 ? res_load(m0 + r0 + (g0 + i) * RPP) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = r0 + (g0 + i) * RPP, m = m0 + r;
                    if (m >= M) continue;
                    f32x4 val = *reinterpret_cast<const f32x4*>(Cs + r * LDC + c4 * 4);
                    val += bias4;
                    val += rg[i];
                    if (p.relu) {
                        val[0] = apse_relu(val[0]); val[1] = apse_relu(val[1]); val[2] = apse_relu(val[2]); val[3] = apse_relu(val[3]);
                    }
                    const size_t dst = (size_t)m * p.y_ld + p.y_coff + n;
                    if (GL_ABLATE == 5) { if (val[0] == 12345.678f) apse_st4(p.y, dst, val, p.y_st); }      // 5: no output stores
                    else if (vec) apse_st4(p.y, dst, val, p.y_st);
                    else for (int k = 0; k < 4; ++k) if (n + k < p.Cout) apse_st1(p.y, dst + k, val[k], p.y_st);
                }
            }
        }
    }
}

// Eligible: 16-bit operands already stored in the operand type, >= 64 input channels (a 64-element step never straddles a
// pixel), unsplit, plain NHWC output, not count-limited, tensors below 4 GiB (32-bit byte offsets in the descriptors).
bool apse_conv_glds16_ok(const ConvParams& p) {
    static const bool off = getenv("APSE_NO_GLDS") != nullptr;            // A/B switch for the sweeps
    if (off) return false;
    if (p.prec != 1 && p.prec != 2) return false;
    if (!p.w16 || p.x_st != p.prec || p.cin_log2 < 6 || (p.KWCp & 63) != 0) return false;
    if (p.splitk != 1 || p.out_mode != 0 || p.m_count || p.tile_cnt) return false;
    if ((p.Cout & 3) != 0) return false;
    if (p.pad > 0 && p.KH * p.KW > 32) return false;                     // one validity bit per filter tap and lane
    const size_t xbytes = (((size_t)p.B * p.H * p.W) << p.cin_log2) * 2;
    const size_t wbytes = (size_t)((p.Cout + 127) / 128 * 128) * p.KH * p.KWCp * 2;
    const size_t xback = (((size_t)p.pad * p.W + p.pad) << p.cin_log2) * 2;      // the activation descriptor starts this far in front of p.x
    return xbytes + xback < 0xfffffff0ull && wbytes < 0xfffffff0ull;
}

template <int PR, int BM, int BN>
static int launch_glds(const ConvParams& p, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    const size_t lds_ring = (size_t)((BM + BN) * 128 * 3 <= 160 * 1024 ? 3 : 2) * (BM + BN) * 128, lds_c = (size_t)BM * (128 + 4) * sizeof(float);
    const size_t lds = lds_ring > lds_c ? lds_ring : lds_c;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_glds16<PR, BM, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.Cout + BN - 1) / BN);
    if (ev0) hipEventRecord(ev0, s);
    hipLaunchKernelGGL((conv_glds16<PR, BM, BN>), dim3(tiles), dim3(512), lds, s, p);
    if (ev1) hipEventRecord(ev1, s);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}

// 256 x 256 tiles where they still give every CU about two tiles (N a multiple of 256: the filters are padded to 128 rows only)
bool apse_conv_glds16_wide(const ConvParams& p) {
    static const bool off = getenv("APSE_GLDS_NARROW") != nullptr;        // A/B switch for the sweeps
    return !off && (p.Cout & 255) == 0 && ((p.M + 255) / 256) * (p.Cout / 256) >= 500;
}
// 128 x 128 tiles (round 4) where 256-row tiles would leave CUs without work: fewer than ~0.8 tiles of 256 x 128 per CU
// (res4 at batch 4: M = 16 128, N = 256 -> 126 tiles of 256 x 128 on 256 CUs, 252 of 128 x 128)
bool apse_conv_glds16_small(const ConvParams& p) {
    static const bool off = getenv("APSE_GLDS_NO128") != nullptr;         // A/B switch for the sweeps
    return !off && ((p.M + 255) / 256) * ((p.Cout + 127) / 128) < 200;
}

// 128 x 256 tiles (round 4): every output column in one tile, so the activation rows -- which come from beyond L2 at ~30 GB/s per
// CU (MI355X_MICROARCH.md, "Indexed rows: gather into LDS"), the rate that paces these kernels -- are fetched ONCE, and half as many
// rows per tile keep all CUs busy where 256-row tiles would not.  For N = 256 layers with about one such tile per CU.
bool apse_conv_glds16_tall(const ConvParams& p) {
    static const bool off = getenv("APSE_GLDS_NO128X256") != nullptr;     // A/B switch for the sweeps
    const int t = ((p.M + 127) / 128) * (p.Cout / 256);
    return !off && p.Cout == 256 && t >= 200 && t <= 640;
}

int apse_launch_conv_glds16(const ConvParams& p, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (apse_conv_glds16_wide(p)) return p.prec == 1 ? launch_glds<1, 256, 256>(p, s, ev0, ev1) : launch_glds<2, 256, 256>(p, s, ev0, ev1);
    if (apse_conv_glds16_tall(p)) return p.prec == 1 ? launch_glds<1, 128, 256>(p, s, ev0, ev1) : launch_glds<2, 128, 256>(p, s, ev0, ev1);
    if (apse_conv_glds16_small(p)) return p.prec == 1 ? launch_glds<1, 128, 128>(p, s, ev0, ev1) : launch_glds<2, 128, 128>(p, s, ev0, ev1);
    return p.prec == 1 ? launch_glds<1, 256, 128>(p, s, ev0, ev1) : launch_glds<2, 256, 128>(p, s, ev0, ev1);
}
