// Implicit-GEMM convolution, 16-bit matrix cores (v_mfma_f32_32x32x16_bf16 / _f16) with f32 storage (gfx950).
//
// Same GEMM view, tiling, LDS swizzle, persistent tile loop, split-K and fused epilogue as
// conv_igemm.hip; the difference is the operand precision: activations and filters are read as f32
// from HBM, rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) while being staged into LDS,
// multiplied on the bf16 MFMA (16x the f32 matrix rate) and accumulated / stored in f32.  An LDS row is
// 64 bf16 = 128 bytes, so the 16-byte-slot XOR swizzle and the fragment addressing are unchanged: one
// ds_read_b128 is exactly the 8 k-values a lane feeds to one 32x32x16 MFMA (k = 8*(lane>>5) + j).
// Used for BASELINE configs 3/5 (cfg.APSE.DTYPE = "bf16"); decision layers (RPN logits/deltas, box
// predictor, mask logits, association FC) stay on the exact-f32 kernel.
#include "apse_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <typename ET, int WM, int WN, int TM, int TN, int KS>
__global__ __launch_bounds__(256) void conv_igemm_bf16(const ConvParams p) {
    typedef ET et8 __attribute__((ext_vector_type(8)));
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int AP = BM / 32;   // staging passes (32 rows x 8 slots of 16 B per pass)
    constexpr int BP = BN / 32;
    constexpr int LDC = BN + 4;
    constexpr int STORE_AT = 4 * KS - 2;          // chunk before which the next k-slice is written to LDS
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                              // [2][KS][BM][64 bf16]  (128-byte rows, 16-byte slots)
    char* Bs = As + 2 * KS * BM * 128;            // [2][KS][BN][64 bf16]
    float* Cs = reinterpret_cast<float*>(smem);   // epilogue view [BM][LDC]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 31, fh = lane >> 5;
    const int srow = tid >> 3, slot = tid & 7;

    int M = p.M;
    if (p.m_count) {
        int lim = (*p.m_count) * p.m_per_item;
        M = lim < M ? lim : M;
    }
    const int tiles_n = (p.Cout + BN - 1) / BN;
    const int tiles_m = (M + BM - 1) / BM;        // live tiles only (M may come from a device count)
    const int nwg = tiles_m * tiles_n;
    const int z = blockIdx.y;
    // a "step" is KS sub-steps of 32 k each between two barriers (KS = 2 for the small tiles, whose
    // 1024-cycle MFMA sub-step is too short to cover an L2/HBM round trip)
    // sub-steps of 64 k here (p.steps_total counts 32-k units of the padded run)
    const int spr64 = (p.KWCp + 63) >> 6;
    const int steps64 = p.KH * spr64;
    const int steps_big = (steps64 + KS - 1) / KS;
    const int per = (steps_big + p.splitk - 1) / p.splitk;
    const int s_begin = z * per;
    const int s_end = (s_begin + per < steps_big) ? s_begin + per : steps_big;
    const int ohw = p.OH * p.OW;
    const size_t w_row = (size_t)p.KH * p.KWCp;
    const bool direct = (p.splitk == 1);

    bool warm_pending = p.next_w && z == 0;     // the block's first tile also warms a slice of the next layer's filters
    // Persistent over tiles: the grid is min(tiles, cap); count-limited launches (packed detection
    // lists) therefore spend nothing on tiles past the device-side count.
    for (int wg = blockIdx.x; wg < nwg; wg += gridDim.x) {
        // XCD-aware remap (blocks are dealt round-robin over 8 XCDs): give each XCD a contiguous
        // range of tiles so neighbours that share an activation tile share an L2.  Bijective form.
        int bid;
        {
            const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
            bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
        const int m0 = tile_m * BM, n0 = tile_n * BN;

        // per-thread staging rows
        int a_iy0[AP], a_ix0[AP], a_pix[AP];
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int m = m0 + srow + 32 * i;
            if (m < M) {
                const int b = m / ohw;
                const int rem = m - b * ohw;
                const int oy = rem / p.OW;
                const int ox = rem - oy * p.OW;
                a_iy0[i] = oy * p.stride - p.pad;
                a_ix0[i] = ox * p.stride - p.pad;
                a_pix[i] = b * p.H * p.W;
            } else {
                a_iy0[i] = -(1 << 28);   // never valid
                a_ix0[i] = 0;
                a_pix[i] = 0;
            }
        }

        et8 ra[KS][AP], rb[KS][BP];
        auto cvt8 = [](const f32x4 lo, const f32x4 hi) {
            et8 r;
            r[0] = (ET)lo[0]; r[1] = (ET)lo[1]; r[2] = (ET)lo[2]; r[3] = (ET)lo[3];
            r[4] = (ET)hi[0]; r[5] = (ET)hi[1]; r[6] = (ET)hi[2]; r[7] = (ET)hi[3];
            return r;
        };
        auto load_step = [&](int sb) {
#pragma unroll
            for (int u = 0; u < KS; ++u) {
                int ss = sb * KS + u;
                const bool live = ss < steps64;
                ss = live ? ss : steps64 - 1;
                const int r = ss / spr64;
                const int q = ((ss - r * spr64) << 6) + (slot << 3);       // 8 consecutive k per thread
                const bool qa = live && q < p.KWCp, qb = live && (q + 4) < p.KWCp;
                const int dpa = q >> p.cin_log2, dpb = (q + 4) >> p.cin_log2;
#pragma unroll
                for (int i = 0; i < AP; ++i) {
                    const int iy = a_iy0[i] + r;
                    if (p.x_st != 0) {
                        // activations already stored in the operand type: one 16-byte load, no conversion
                        // (Cin >= 8 here, so the 8 k-values of a slot belong to one pixel)
                        et8 v;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (ET)0.f;
                        if (qa && (unsigned)iy < (unsigned)p.H && (unsigned)(a_ix0[i] + dpa) < (unsigned)p.W) {
                            const int off = ((a_pix[i] + iy * p.W + a_ix0[i]) << p.cin_log2) + q;
                            v = *reinterpret_cast<const et8*>(reinterpret_cast<const ET*>(p.x) + off);
                        }
                        ra[u][i] = v;
                        continue;
                    }
                    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
                    if ((unsigned)iy < (unsigned)p.H) {
                        const int off = ((a_pix[i] + iy * p.W + a_ix0[i]) << p.cin_log2) + q;
                        if (qa && (unsigned)(a_ix0[i] + dpa) < (unsigned)p.W) lo = *reinterpret_cast<const f32x4*>(p.x + off);
                        if (qb && (unsigned)(a_ix0[i] + dpb) < (unsigned)p.W) hi = *reinterpret_cast<const f32x4*>(p.x + off + 4);
                    }
                    ra[u][i] = cvt8(lo, hi);
                }
#pragma unroll
                for (int i = 0; i < BP; ++i) {
                    const int n = n0 + srow + 32 * i;
                    const size_t wo = (size_t)n * w_row + (size_t)r * p.KWCp + q;
                    if (p.w16) {                     // filters pre-rounded to bf16: one 16-byte load per slot
                        et8 v;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (ET)0.f;
                        if (qa) v = *reinterpret_cast<const et8*>(p.w16 + wo);
                        rb[u][i] = v;
                    } else {
                        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
                        if (qa) lo = *reinterpret_cast<const f32x4*>(p.w + wo);
                        if (qb) hi = *reinterpret_cast<const f32x4*>(p.w + wo + 4);
                        rb[u][i] = cvt8(lo, hi);
                    }
                }
            }
        };
        auto store_step = [&](int buf) {
#pragma unroll
            for (int u = 0; u < KS; ++u) {
#pragma unroll
                for (int i = 0; i < AP; ++i) {
                    const int row = srow + 32 * i;
                    const int ps = slot ^ ((row >> 1) & 7);
                    *reinterpret_cast<et8*>(As + ((buf * KS + u) * BM + row) * 128 + ps * 16) = ra[u][i];
                }
#pragma unroll
                for (int i = 0; i < BP; ++i) {
                    const int row = srow + 32 * i;
                    const int ps = slot ^ ((row >> 1) & 7);
                    *reinterpret_cast<et8*>(Bs + ((buf * KS + u) * BN + row) * 128 + ps * 16) = rb[u][i];
                }
            }
        };

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

        if (s_begin < s_end) {
            ApseWarm warm;
            const bool warm_now = warm_pending;
            warm_pending = false;
            if (warm_now) apse_warm_issue(warm, p.next_w, p.next_w_bytes, blockIdx.x, gridDim.x, tid);
            load_step(s_begin);
            store_step(0);
            if (warm_now) apse_warm_retire(warm);
            __syncthreads();
            for (int sb = s_begin; sb < s_end; ++sb) {
                const int buf = (sb - s_begin) & 1;
                if (sb + 1 < s_end) load_step(sb + 1);
                // fragments are double-buffered in registers: the ds_read_b128s of chunk c+1 are issued
                // before the MFMAs of chunk c, so LDS latency hides behind the matrix pipe.
                et8 af[2][TM], bf[2][TN];
                auto load_frags = [&](int cc, int fb) {
                    const int u = cc >> 2, c = cc & 3;
                    const char* Ab = As + (buf * KS + u) * BM * 128;
                    const char* Bb = Bs + (buf * KS + u) * BN * 128;
                    const int ls = 2 * c + fh;          // slot = 16 k of MFMA step c, half fh (k = 8*fh + j)
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int row = (wm * TM + i) * 32 + fr;
                        af[fb][i] = *reinterpret_cast<const et8*>(Ab + row * 128 + ((ls ^ ((row >> 1) & 7)) << 4));
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int row = (wn * TN + j) * 32 + fr;
                        bf[fb][j] = *reinterpret_cast<const et8*>(Bb + row * 128 + ((ls ^ ((row >> 1) & 7)) << 4));
                    }
                };
                load_frags(0, 0);
#pragma unroll
                for (int cc = 0; cc < 4 * KS; ++cc) {
                    if (cc + 1 < 4 * KS) load_frags(cc + 1, (cc + 1) & 1);
                    if (cc == STORE_AT && sb + 1 < s_end) store_step(buf ^ 1);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = mfma16(af[cc & 1][i], bf[cc & 1][j], acc[i][j]);
                }
                __syncthreads();
            }
        }

        // -------------------------------------------------------------- epilogue
        // Accumulators -> LDS (C layout, padded rows) -> 16-byte vector stores: one thread handles whole
        // float4 chunks of a row, so bias / residual / output move as dwordx4 and the per-element address
        // arithmetic of the 32x32 C/D map (col = lane&31, row = (v&3) + 8*(v>>2) + 4*(lane>>5)) disappears.
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int row = (wm * TM + i) * 32 + (v & 3) + 8 * (v >> 2) + 4 * fh;
                    Cs[row * LDC + (wn * TN + j) * 32 + fr] = acc[i][j][v];
                }
        __syncthreads();
        constexpr int C4 = BN / 4;                 // float4 chunks per tile row
        constexpr int RPP = 256 / C4;              // rows covered per pass
        const int c4 = tid % C4;
        const int n = n0 + c4 * 4;
        bool vec_direct;
        if (p.out_mode == 1) vec_direct = (p.cdec & 3) == 0;
        else vec_direct = ((p.y_ld & 3) == 0) && ((p.y_coff & 3) == 0) && (p.y_coff + n + 4 <= p.y_ld);
        const bool vec_ws = (p.Cout & 3) == 0;
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        int co = n, g = 0;
        if (n < p.Cout) {
            if (p.out_mode == 1) { g = n / p.cdec; co = n - g * p.cdec; }
            if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + co);     // bias is padded to Cout_p
        }
        // bias + residual + ReLU + store of one float4 of output row m (the fused epilogue proper)
        auto finish = [&](f32x4 val, int m) {
            val += bias4;
            size_t dst;                 // element index into y
            if (p.out_mode == 0) {
                if (p.res_mode == 1) {
                    val += apse_ld4(p.res, (size_t)m * p.Cout + n, p.res_st);
                } else if (p.res_mode == 2) {
                    const int b = m / ohw;
                    const int rem = m - b * ohw;
                    const int oy = rem / p.OW, ox = rem - oy * p.OW;
                    const int hw2 = (p.OH >> 1) * (p.OW >> 1);
                    val += apse_ld4(p.res, ((size_t)b * hw2 + (oy >> 1) * (p.OW >> 1) + (ox >> 1)) * p.Cout + n, p.res_st);
                }
                dst = (size_t)m * p.y_ld + p.y_coff + n;
            } else {
                const int b = m / ohw;
                const int rem = m - b * ohw;
                const int oy = rem / p.OW, ox = rem - oy * p.OW;
                dst = (((size_t)b * 2 * p.OH + 2 * oy + (g >> 1)) * (2 * p.OW) + 2 * ox + (g & 1)) * p.cdec + co;
            }
            if (p.relu) {
                val[0] = val[0] > 0.f ? val[0] : 0.f; val[1] = val[1] > 0.f ? val[1] : 0.f;
                val[2] = val[2] > 0.f ? val[2] : 0.f; val[3] = val[3] > 0.f ? val[3] : 0.f;
            }
            if (vec_direct) {
                apse_st4(p.y, dst, val, p.y_st);
            } else {
                for (int k = 0; k < 4; ++k) if (n + k < p.Cout) apse_st1(p.y, dst + k, val[k], p.y_st);
            }
        };
        if (direct) {
            if (n < p.Cout)
                for (int r = tid / C4; r < BM; r += RPP) {
                    const int m = m0 + r;
                    if (m >= M) break;
                    finish(*reinterpret_cast<const f32x4*>(Cs + r * LDC + c4 * 4), m);
                }
        } else {
            // split-K: this block's partial tile goes to its slab of the workspace
            if (n < p.Cout)
                for (int r = tid / C4; r < BM; r += RPP) {
                    const int m = m0 + r;
                    if (m >= M) break;
                    const f32x4 val = *reinterpret_cast<const f32x4*>(Cs + r * LDC + c4 * 4);
                    float* dst = p.ws + ((size_t)z * p.M + m) * p.Cout + n;
                    if (vec_ws) *reinterpret_cast<f32x4*>(dst) = val;
                    else for (int k = 0; k < 4; ++k) if (n + k < p.Cout) dst[k] = val[k];
                }
            if (p.tile_cnt) {
                // In-launch reduction by the last-arriving K-slice of this tile (cdna_hip_programming.md, section 5
                // "In-launch split-K reduction"): plain slab stores -> every wave drains its stores -> barrier ->
                // one lane: agent-scope release, arrival ticket; the block that draws splitk-1 acquires and sums
                // the slabs in the fixed order z = 0..splitk-1 (bitwise reproducible, independent of arrival order).
                int* flag = reinterpret_cast<int*>(Cs + BM * LDC);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const int ticket = __hip_atomic_fetch_add(p.tile_cnt + bid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int last = ticket == p.splitk - 1;
                    if (last) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __hip_atomic_store(p.tile_cnt + bid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
                    }
                    *flag = last;
                }
                __syncthreads();
                if (*flag && n < p.Cout) {
                    for (int r = tid / C4; r < BM; r += RPP) {
                        const int m = m0 + r;
                        if (m >= M) break;
                        f32x4 val = {0.f, 0.f, 0.f, 0.f};
                        const float* src = p.ws + (size_t)m * p.Cout + n;
                        for (int zz = 0; zz < p.splitk; ++zz) {
                            if (vec_ws) val += *reinterpret_cast<const f32x4*>(src + (size_t)zz * p.M * p.Cout);
                            else for (int k = 0; k < 4; ++k) if (n + k < p.Cout) val[k] += src[(size_t)zz * p.M * p.Cout + k];
                        }
                        finish(val, m);
                    }
                }
            }
        }
        __syncthreads();       // Cs is overwritten by the next tile's staging
    }
}


template <typename ET, int WM, int WN, int TM, int TN, int KS>
static int launch_bf16(const ConvParams& p, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int tiles = ((p.M + BM - 1) / BM) * ((p.Cout + BN - 1) / BN);
    const size_t lds_stage = (size_t)2 * KS * (BM + BN) * 128, lds_c = (size_t)BM * (BN + 4) * sizeof(float) + 16;
    const size_t lds = lds_stage > lds_c ? lds_stage : lds_c;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_bf16<ET, WM, WN, TM, TN, KS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    // count-limited launches are persistent over the live tiles: size the grid for about twice the expected
    // count (dead blocks still cost their launch), never above 1024 blocks
    int grid_x = tiles;
    if (p.m_count) {
        int want = 1024;
        if (p.m_hint > 0) {
            want = 2 * ((p.m_hint + BM - 1) / BM) * ((p.Cout + BN - 1) / BN);
            if (want < 64) want = 64;
            if (want > 1024) want = 1024;
        }
        if (grid_x > want) grid_x = want;
    }
    hipLaunchKernelGGL((conv_igemm_bf16<ET, WM, WN, TM, TN, KS>), dim3(grid_x, p.splitk), dim3(256), lds, s, p);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}

// The caller (apse_launch_conv) adds the split-K reduce pass.  prec 1 = bf16, 2 = f16 operands.
template <typename ET>
static int launch_et(const ConvParams& p, int cfg, hipStream_t s) {
    switch (cfg) {
        case 0: return launch_bf16<ET, 2, 2, 2, 2, 1>(p, s);
        case 1: return launch_bf16<ET, 2, 2, 1, 1, 2>(p, s);
        case 2: return launch_bf16<ET, 4, 1, 1, 1, 2>(p, s);
        case 3: return launch_bf16<ET, 4, 1, 1, 2, 1>(p, s);
        case 4: return launch_bf16<ET, 2, 2, 1, 1, 2>(p, s);      // the f32 kernel's extra shapes map to their nearest 16-bit one
        case 5: return launch_bf16<ET, 4, 1, 1, 1, 2>(p, s);
        case 6: case 7: return launch_bf16<ET, 2, 2, 1, 1, 2>(p, s);
        case 8: return launch_bf16<ET, 2, 2, 2, 2, 1>(p, s);
        default: return APSE_E_INVALID;
    }
}
int apse_launch_conv_bf16(const ConvParams& p, int cfg, hipStream_t s) {
    return p.prec == 2 ? launch_et<_Float16>(p, cfg, s) : launch_et<__bf16>(p, cfg, s);
}
