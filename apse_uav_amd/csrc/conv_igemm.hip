// Implicit-GEMM convolution for gfx950 (CDNA4), exact-f32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Replaces, on the reference's path, every cuDNN/cuBLAS contraction reached through
// detectron2's backbone / FPN / RPN head / box head / mask head and the association FC
// (/root/reference/dcnn/networks/track_rcnn.py:42-51, dcnn/networks/association_head.py:23).
//
// GEMM view:  D[m][n] = sum_k A[m][k] * Wt[n][k]
//   m = (b, oy, ox) output pixel, n = output channel, k = (r, q) with q running over the
//   contiguous NHWC run of KW pixels x Cin channels of filter row r (so one k-step of 32 floats
//   is one 128-byte contiguous read per output pixel; out-of-image pixels are zero-filled).
// Block = 256 threads = 4 waves (one per SIMD), or 512 threads = two k groups of 4 waves (WK = 2); wave tile =
// TM x TN MFMA tiles of 32x32.
// LDS: A and B k-slices [rows][32 f32] double-buffered, 16-byte slots XOR-swizzled with
// (row>>1)&7 so both the ds_write_b128 staging and the ds_read_b128 fragment reads are
// bank-conflict-free (MI355X_MICROARCH.md, LDS table: b128 reads are served per 16-lane group
// over 64 banks).  One ds_read_b128 feeds FOUR MFMA k-steps: lane half h takes k = 8c+4h+j for
// step j (the k order inside a chunk is a free choice as long as A and B agree).
// Global->LDS staging goes through registers, one 16-byte "piece" per MFMA gap: the pieces of k-step s+1
// (s+2 for the small tiles, which keep two register sets) are fetched during the first half of step s and
// handed to LDS during its second half (an f32 MFMA k-step is 1024 cycles/wave at 64x64 and 4096 at
// 128x128); one barrier per k-step.  See DESIGN.md section 3 for the measured effect of each choice.
#include "apse_common.h"
#include <type_traits>

// XT = 0: x is f32 and is read through a buffer descriptor: a tap outside the image (or a k sub-step past the end)
// gets an offset beyond the buffer, for which the hardware range check returns zeros -> the staging code has no
// branches, the whole k-step is one basic block and the compiler spreads the address arithmetic and the loads over
// the gaps of the MFMA stream.  XT = 1: x may be 16-bit (typed loads, conditional; only the small-Cout head layers
// of the 16-bit modes take this instantiation).
// WK = 2: the block has 8 waves; wave group kg = wave / 4 owns the k sub-steps u with u % WK == kg of every step and
// the groups' partial tiles are added (fixed order) in the epilogue.  For layers with about one tile per CU
// (res4 at batch 1) this puts two waves on every SIMD without a second pass over a split-K workspace.
// PR = 0: f32 operands (v_mfma_f32_32x32x2_f32).  PR = 1 / 2: bf16 / f16 operands already STORED in that type
// (activations x_st == PR, filters pre-rounded in w16; v_mfma_f32_32x32x16_*): an LDS row is then 64 elements = the same
// 128 bytes, a staged piece the same 16 bytes, so the staging, swizzle and epilogue are shared; only the element
// count per k sub-step (64 instead of 32) and the MFMA issue differ.  (Round 4: the older kernel that converted f32-stored
// activations to 16-bit operands while staging -- conv_igemm_bf16.hip -- is gone: no BASELINE configuration stores f32 and
// multiplies in 16 bits; such a request is refused.)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 apse_mfma16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 apse_mfma16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <int WM, int WN, int TM, int TN, int KS, int XT, int WK, int PR = 0>
__global__ __launch_bounds__(64 * WM * WN * WK) void conv_igemm_f32(const ConvParams p) {
    static_assert(PR == 0 || XT == 0, "16-bit operands use the descriptor path only");
    // XT = 2 / 3: descriptor path with bf16 / f16 activations WIDENED to f32 operands (the exact-f32 decision heads of the
    // 16-bit modes: RPN logits / deltas, box predictor, mask logits): 8-byte fetches of 4 elements, widened on the way to LDS
    constexpr bool DESC = XT != 1;
    constexpr int EPS = PR ? 64 : 32;              // elements per k sub-step (one 128-byte LDS row)
    constexpr int ESH = PR ? 1 : 2;                // log2(bytes per element)
    constexpr int EPSLOT = 16 >> ESH;              // elements per 16-byte slot
    constexpr int NT = 64 * WM * WN * WK;          // WM x WN waves per k group (4, or 8 for the 256x128 tile)
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int SR = NT / 8;           // rows staged per pass (8 slots of 16 B per row)
    constexpr int AP = BM / SR;          // staging passes
    constexpr int BP = BN / SR;
    static_assert(AP >= 1 && BP >= 1 && KS % WK == 0 && (DESC || WK == 1), "unsupported shape");
    constexpr int LDC = BN + 4;
    constexpr bool DP = DESC && (TM * TN <= 2 || (PR != 0 && WM * WN == 8 && APSE_DP16));      // 16-bit big tiles: a second register set (fetch two k-steps ahead) where the budget allows
    constexpr int NSET = DP ? 2 : 1;
    constexpr int STORE_AT = 4 * KS - 2;          // chunk before which the next k-slice is written to LDS
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);   // [2][KS][BM*32]
    float* Bs = As + 2 * KS * BM * 32;            // [2][KS][BN*32]
    float* Cs = reinterpret_cast<float*>(smem);   // epilogue view [WK][BM][LDC]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int kg = wave / (WM * WN);               // k group of this wave (0 when WK = 1)
    const int wm = (wave % (WM * WN)) / WN, wn = wave % WN;
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
    const int steps_per_row = p.KWCp / EPS;
    const int steps_total = PR ? p.KH * steps_per_row : p.steps_total;
    const int steps_big = (steps_total + KS - 1) / KS;
    const int per = (steps_big + p.splitk - 1) / p.splitk;
    const int s_begin = z * per;
    const int s_end = (s_begin + per < steps_big) ? s_begin + per : steps_big;
    const int ohw = p.OH * p.OW;
    const size_t w_row = (size_t)p.KH * p.KWCp;
    const bool direct = (p.splitk == 1);
    __amdgpu_buffer_rsrc_t xrsrc;
    if constexpr (DESC)
        xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)((((unsigned)(p.B * p.H * p.W) << p.cin_log2)) << (XT >= 2 ? 1 : ESH)), 0x00020000);

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
        int a_iy0[AP], a_ix0[AP], a_pix[AP], a_base[AP];
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int m = m0 + srow + SR * i;
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
            a_base[i] = (a_pix[i] + a_iy0[i] * p.W + a_ix0[i]) << p.cin_log2;   // element offset of tap (r = 0, q = 0); only used when valid
        }

        // DP (small tiles): two register sets, a k-step's operands are fetched TWO steps ahead and handed to LDS one step
        // ahead, so a fetch has 1.5 steps (>= 3000 MFMA cycles) to land instead of half a step
        f32x4 ra[NSET][KS][AP], rb[NSET][KS][BP];
        // (filter row, 32-float step inside the row) of the next k sub-step to fetch, advanced incrementally
        int ld_r, ld_q;
        {
            const int ss0 = s_begin * KS;
            ld_r = ss0 / steps_per_row;
            ld_q = ss0 - ld_r * steps_per_row;
        }
        const char* wrow[BP];                       // byte pointers: f32 filters or the pre-rounded 16-bit copy
#pragma unroll
        for (int i = 0; i < BP; ++i)
            wrow[i] = (PR ? reinterpret_cast<const char*>(p.w16) : reinterpret_cast<const char*>(p.w)) +
                      ((((size_t)(n0 + srow + SR * i) * w_row) + (size_t)(slot * EPSLOT)) << ESH);
        // One "piece" = one 16-byte fetch of this thread (pieces 0..AP-1: A rows, AP..AP+BP-1: B rows of sub-step u).
        // The k-step issues its pieces one per MFMA group, pinned there with sched_barriers, so the address
        // arithmetic and the fetches ride in the gaps of the matrix pipe instead of in front of it.
        int cur_r = 0, cur_ry = 0, cur_q = 0, cur_rowoff = 0, cur_woff = 0, cur_dpx = 0;
        auto fetch_piece = [&](int set, int u, int j) {
            if (j == 0) {                      // scalars of sub-step u
                const bool live = ld_r < p.KH;
                cur_r = live ? ld_r : p.KH - 1;
                cur_ry = live ? ld_r : (1 << 28);
                cur_q = ld_q * EPS + slot * EPSLOT;
                cur_dpx = cur_q >> p.cin_log2;
                cur_rowoff = ((cur_r * p.W) << p.cin_log2) + cur_q;
                cur_woff = (cur_r * p.KWCp + ld_q * EPS) << ESH;      // bytes
                ++ld_q;
                if (ld_q == steps_per_row) { ld_q = 0; ++ld_r; }
            }
            if (j < AP) {
                const int iy = a_iy0[j] + cur_ry;            // cur_ry is far out of range on a dead sub-step
                const int px = a_ix0[j] + cur_dpx;
                const int okm = -(int)(((unsigned)iy < (unsigned)p.H) & ((unsigned)px < (unsigned)p.W));   // all ones when the tap is inside
                if constexpr (XT >= 2) {             // 4 elements of 2 bytes; widened in store_piece, once the data has landed
                    const unsigned off = (((unsigned)(a_base[j] + cur_rowoff) << 1) & (unsigned)okm) | (0xfffffff0u & ~(unsigned)okm);
                    const auto raw = __builtin_amdgcn_raw_buffer_load_b64(xrsrc, (int)off, 0, 0);
                    ra[set][u][j] = f32x4{__uint_as_float(raw[0]), __uint_as_float(raw[1]), 0.f, 0.f};
                } else {
                    const unsigned off = (((unsigned)(a_base[j] + cur_rowoff) << ESH) & (unsigned)okm) | (0xfffffff0u & ~(unsigned)okm);
                    ra[set][u][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off, 0, 0));
                }
            } else {
                rb[set][u][j - AP] = *reinterpret_cast<const f32x4*>(wrow[j - AP] + cur_woff);      // 16 bytes of filter row
            }
        };
        auto store_piece = [&](int set, int buf, int u, int j) {
            if (j < AP) {
                const int row = srow + SR * j;
                const int ps = slot ^ ((row >> 1) & 7);
                f32x4 v = ra[set][u][j];
                if constexpr (XT == 2) {             // bf16 pairs -> f32
                    const unsigned lo = __float_as_uint(v[0]), hi = __float_as_uint(v[1]);
                    v = f32x4{__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
                } else if constexpr (XT == 3) {      // f16 pairs -> f32
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    const h2 a = __builtin_bit_cast(h2, __float_as_uint(v[0])), b = __builtin_bit_cast(h2, __float_as_uint(v[1]));
                    v = f32x4{(float)a[0], (float)a[1], (float)b[0], (float)b[1]};
                }
                *reinterpret_cast<f32x4*>(As + (buf * KS + u) * BM * 32 + row * 32 + ps * 4) = v;
            } else {
                const int row = srow + SR * (j - AP);
                const int ps = slot ^ ((row >> 1) & 7);
                *reinterpret_cast<f32x4*>(Bs + (buf * KS + u) * BN * 32 + row * 32 + ps * 4) = rb[set][u][j - AP];
            }
        };
        auto load_step_bl = [&](int set) {
#pragma unroll
            for (int u = 0; u < KS; ++u)
#pragma unroll
                for (int j = 0; j < AP + BP; ++j) fetch_piece(set, u, j);
        };
        auto load_step = [&](int sb) {
            if constexpr (DESC) { load_step_bl(0); return; }
#pragma unroll
            for (int u = 0; u < KS; ++u) {
                int ss = sb * KS + u;
                const bool live = ss < p.steps_total;       // odd tail of a KS = 2 schedule: A is zero-filled
                ss = live ? ss : p.steps_total - 1;
                const int r = ss / steps_per_row;
                const int q = ((ss - r * steps_per_row) << 5) + (slot << 2);
                const int dpx = q >> p.cin_log2;
#pragma unroll
                for (int i = 0; i < AP; ++i) {
                    const int iy = a_iy0[i] + r;
                    const int px = a_ix0[i] + dpx;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (live && (unsigned)iy < (unsigned)p.H && (unsigned)px < (unsigned)p.W) {
                        const int off = ((a_pix[i] + iy * p.W + a_ix0[i]) << p.cin_log2) + q;
                        v = apse_ld4(p.x, (size_t)off, p.x_st);      // 16-bit activations widen exactly to f32
                    }
                    ra[0][u][i] = v;
                }
#pragma unroll
                for (int i = 0; i < BP; ++i) {
                    const int n = n0 + srow + 32 * i;
                    rb[0][u][i] = *reinterpret_cast<const f32x4*>(p.w + (size_t)n * w_row + (size_t)r * p.KWCp + q);
                }
            }
        };
        auto store_step = [&](int buf) {
            if constexpr (DESC) {                  // the piece form knows the block's row stride (SR)
#pragma unroll
                for (int u = 0; u < KS; ++u)
#pragma unroll
                    for (int j = 0; j < AP + BP; ++j) store_piece(0, buf, u, j);
                return;
            }
#pragma unroll
            for (int u = 0; u < KS; ++u) {
#pragma unroll
                for (int i = 0; i < AP; ++i) {
                    const int row = srow + 32 * i;
                    const int ps = slot ^ ((row >> 1) & 7);
                    *reinterpret_cast<f32x4*>(As + (buf * KS + u) * BM * 32 + row * 32 + ps * 4) = ra[0][u][i];
                }
#pragma unroll
                for (int i = 0; i < BP; ++i) {
                    const int row = srow + 32 * i;
                    const int ps = slot ^ ((row >> 1) & 7);
                    *reinterpret_cast<f32x4*>(Bs + (buf * KS + u) * BN * 32 + row * 32 + ps * 4) = rb[0][u][i];
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
            if constexpr (DESC) {
                constexpr int NP = KS * (AP + BP);          // fetch pieces per k-step
                constexpr int G = 16 * KS / WK;             // MFMA groups per k-step and wave (TM*TN MFMAs each)
                constexpr int SP = (G / 2) / NP > 0 ? (G / 2) / NP : 1;
                static_assert(PR != 0 || (NP * SP <= G / 2 + SP - 1 && G / 2 + (NP - 1) * SP < G), "piece schedule does not fit the k-step");
                if constexpr (DP) load_step_bl(1);          // operands of the second step, in flight across the first
                // one k-step; PAR = parity of the step inside this K slice = LDS buffer it reads
                auto kstep = [&](auto PAR) {
                    constexpr int buf = decltype(PAR)::value;
                    constexpr int FS = DP ? buf : 0;          // register set the fetches of this step fill
                    constexpr int SS = DP ? (buf ^ 1) : 0;    // register set handed to LDS buffer buf^1 in this step
                    f32x4 af[2][TM], bf[2][TN];
                    auto load_frags = [&](int cc, int fb) {
                        const int u = (cc >> 2) * WK + kg, c = cc & 3;      // this wave group's sub-steps only
                        const float* Ab = As + (buf * KS + u) * BM * 32;
                        const float* Bb = Bs + (buf * KS + u) * BN * 32;
                        const int ls = 2 * c + fh;
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const int row = (wm * TM + i) * 32 + fr;
                            af[fb][i] = *reinterpret_cast<const f32x4*>(Ab + row * 32 + ((ls ^ ((row >> 1) & 7)) << 2));
                        }
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const int row = (wn * TN + j) * 32 + fr;
                            bf[fb][j] = *reinterpret_cast<const f32x4*>(Bb + row * 32 + ((ls ^ ((row >> 1) & 7)) << 2));
                        }
                    };
                    load_frags(0, 0);
                    if constexpr (PR != 0) {
                        // 16-bit operands: one MFMA per (chunk, tile) -- the gaps are counted per MFMA and may carry
                        // more than one piece (a 64-deep sub-step is a quarter of the f32 step's matrix time)
                        typedef typename std::conditional<PR == 1, bf16x8, f16x8>::type op8;
                        constexpr int MPC = TM * TN;
                        constexpr int NG = (4 * KS / WK) * MPC;
                        constexpr int PPG = (NP + NG / 2 - 1) / (NG / 2);
#pragma unroll
                        for (int cc = 0; cc < 4 * KS / WK; ++cc) {
                            if (cc + 1 < 4 * KS / WK) load_frags(cc + 1, (cc + 1) & 1);
#pragma unroll
                            for (int i = 0; i < TM; ++i)
#pragma unroll
                                for (int j = 0; j < TN; ++j) {
                                    const int g = cc * MPC + i * TN + j;
                                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                    for (int q = 0; q < PPG; ++q) {
                                        const int pc = (g < NG / 2 ? g : g - NG / 2) * PPG + q;
                                        if (pc < NP) {
                                            if (g < NG / 2) fetch_piece(FS, pc / (AP + BP), pc % (AP + BP));
                                            else store_piece(SS, buf ^ 1, pc / (AP + BP), pc % (AP + BP));
                                        }
                                    }
                                    __builtin_amdgcn_sched_barrier(0);
                                    acc[i][j] = apse_mfma16(__builtin_bit_cast(op8, af[cc & 1][i]), __builtin_bit_cast(op8, bf[cc & 1][j]), acc[i][j]);
                                }
                        }
                        __syncthreads();
                        return;
                    }
#pragma unroll
                    for (int cc = 0; cc < 4 * KS / WK; ++cc) {
                        if (cc + 1 < 4 * KS / WK) load_frags(cc + 1, (cc + 1) & 1);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int g = cc * 4 + k;
                            // first half of the step: fetch piece g/SP of the step after next (DP) / the next step
                            // (unconditional: a dead step past the end reads zeros / the next K slice and is never
                            // consumed); second half: hand the NEXT step's operands to LDS buffer buf^1
                            if (g % SP == 0 && g / SP < NP) {
                                __builtin_amdgcn_sched_barrier(0);
                                fetch_piece(FS, (g / SP) / (AP + BP), (g / SP) % (AP + BP));
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            if (g >= G / 2 && (g - G / 2) % SP == 0 && (g - G / 2) / SP < NP) {
                                __builtin_amdgcn_sched_barrier(0);
                                store_piece(SS, buf ^ 1, ((g - G / 2) / SP) / (AP + BP), ((g - G / 2) / SP) % (AP + BP));
                                __builtin_amdgcn_sched_barrier(0);
                            }
#pragma unroll
                            for (int i = 0; i < TM; ++i)
#pragma unroll
                                for (int j = 0; j < TN; ++j)
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cc & 1][i][k], bf[cc & 1][j][k], acc[i][j], 0, 0, 0);
                        }
                    }
                    __syncthreads();
                };
                for (int sb = s_begin; sb < s_end; sb += 2) {
                    kstep(std::integral_constant<int, 0>{});
                    if (sb + 1 < s_end) kstep(std::integral_constant<int, 1>{});
                }
            } else
            for (int sb = s_begin; sb < s_end; ++sb) {
                const int buf = (sb - s_begin) & 1;
                const bool more = sb + 1 < s_end;
                if (more) load_step(sb + 1);
                // fragments are double-buffered in registers: the ds_read_b128s of chunk c+1 are issued
                // before the MFMAs of chunk c, so LDS latency hides behind the matrix pipe.
                f32x4 af[2][TM], bf[2][TN];
                auto load_frags = [&](int cc, int fb) {
                    const int u = cc >> 2, c = cc & 3;
                    const float* Ab = As + (buf * KS + u) * BM * 32;
                    const float* Bb = Bs + (buf * KS + u) * BN * 32;
                    const int ls = 2 * c + fh;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int row = (wm * TM + i) * 32 + fr;
                        af[fb][i] = *reinterpret_cast<const f32x4*>(Ab + row * 32 + ((ls ^ ((row >> 1) & 7)) << 2));
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int row = (wn * TN + j) * 32 + fr;
                        bf[fb][j] = *reinterpret_cast<const f32x4*>(Bb + row * 32 + ((ls ^ ((row >> 1) & 7)) << 2));
                    }
                };
                load_frags(0, 0);
#pragma unroll
                for (int cc = 0; cc < 4 * KS; ++cc) {
                    if (cc + 1 < 4 * KS) load_frags(cc + 1, (cc + 1) & 1);
                    // the next step's tile goes to LDS in the middle of this step's MFMA stream (its global
                    // loads were issued at the top of the step): the ds_writes issue under the matrix pipe
                    // instead of in front of the barrier.  buf^1 was last read in the previous step.
                    if (cc == STORE_AT && more) store_step(buf ^ 1);
#ifdef APSE_SETPRIO
                    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cc & 1][i][k], bf[cc & 1][j][k], acc[i][j], 0, 0, 0);
#ifdef APSE_SETPRIO
                    __builtin_amdgcn_s_setprio(0);
#endif
                }
                __syncthreads();
            }
        }

        // -------------------------------------------------------------- epilogue
        // Accumulators -> LDS (C layout, padded rows) -> 16-byte vector stores: one thread handles whole
        // float4 chunks of a row, so bias / residual / output move as dwordx4 and the per-element address
        // arithmetic of the 32x32 C/D map (col = lane&31, row = (v&3) + 8*(v>>2) + 4*(lane>>5)) disappears.
        constexpr int C4 = BN / 4;                 // float4 chunks per tile row
        constexpr int RPP = NT / C4;               // rows covered per pass
        constexpr int PASSES = BM / RPP;
        constexpr int RG = PASSES < 8 ? PASSES : 8;   // residual rows fetched together (independent loads, one wait)
        const int c4 = tid % C4;
        const int n = n0 + c4 * 4;
        const int r0 = tid / C4;
        // residual operand of output row m (zeros when the layer has none)
        auto res_load = [&](int m) -> f32x4 {
            f32x4 rv = {0.f, 0.f, 0.f, 0.f};
            if (p.out_mode == 0 && m < M && n < p.Cout) {
                if (p.res_mode == 1) {
                    rv = apse_ld4(p.res, (size_t)m * p.Cout + n, p.res_st);
                } else if (p.res_mode == 2) {
                    const int b = m / ohw;
                    const int rem = m - b * ohw;
                    const int oy = rem / p.OW, ox = rem - oy * p.OW;
                    const int hw2 = (p.OH >> 1) * (p.OW >> 1);
                    rv = apse_ld4(p.res, ((size_t)b * hw2 + (oy >> 1) * (p.OW >> 1) + (ox >> 1)) * p.Cout + n, p.res_st);
                }
            }
            return rv;
        };
        // Fast path of the direct epilogue (round 4; every trunk layer takes it): plain NHWC output, output and same-resolution
        // residual in ONE storage type known at compile time (f32, or the operand type of the 16-bit kernels), whole vectors.  Rows
        // go through buffer descriptors sized to M rows -- a row past M is out of range by construction: its loads give zeros, its
        // stores are dropped -- with one byte offset per thread and a uniform step per pass: no branch, no 64-bit arithmetic and
        // no run-time storage switch per element.  Same values in the same order as the general form.
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        const bool want_res = direct && p.res_mode != 0;
        const bool fast_ok = direct && p.out_mode == 0 && n < p.Cout && ((p.y_ld & 3) == 0) && ((p.y_coff & 3) == 0) && (p.y_coff + n + 4 <= p.y_ld) &&
                             (p.res_mode == 0 || (p.res_mode == 1 && p.res_st == p.y_st)) && (p.y_st == 0 || p.y_st == PR) &&
                             ((size_t)M + BM) * (size_t)p.y_ld * 4 < 0xfffffff0ull && ((size_t)M + BM) * (size_t)p.Cout * 4 < 0xfffffff0ull;
        const unsigned fes = p.y_st ? 2u : 4u;
        const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((unsigned)M * (unsigned)p.y_ld * fes), 0x00020000);
        const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(want_res ? p.res : p.y), 0,
                                                                               (int)(want_res ? (unsigned)M * (unsigned)p.Cout * fes : 0u), 0x00020000);
        const unsigned ystep = (unsigned)(RPP * p.y_ld) * fes, rstep = (unsigned)(RPP * p.Cout) * fes;
        const unsigned yo = (unsigned)((m0 + r0) * p.y_ld + p.y_coff + n) * fes, ro = (unsigned)((m0 + r0) * p.Cout + n) * fes;
        auto fast_rload = [&](int pass) -> f32x4 {
            const int off = (int)(ro + (unsigned)pass * rstep);
            if (PR == 0 || p.y_st == 0) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off, 0, 0));
            const u32x2_t q = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rrsrc, off, 0, 0));
            if constexpr (PR == 2) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                const h4 hv = __builtin_bit_cast(h4, q);
                return f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
            } else {
                return f32x4{__uint_as_float(q[0] << 16), __uint_as_float(q[0] & 0xffff0000u), __uint_as_float(q[1] << 16), __uint_as_float(q[1] & 0xffff0000u)};
            }
        };
        // the first group of residual rows is requested BEFORE the accumulators go through LDS: its HBM / MALL round
        // trip (the skip tensor was written two or three layers ago) overlaps the C-tile shuffle and the barrier
        f32x4 rgrp[RG];
        if (want_res && fast_ok) {
#pragma unroll
            for (int i = 0; i < RG; ++i) rgrp[i] = fast_rload(i);
        } else if (want_res) {
#pragma unroll
            for (int i = 0; i < RG; ++i) rgrp[i] = res_load(m0 + r0 + i * RPP);
        } else {
#pragma unroll
            for (int i = 0; i < RG; ++i) rgrp[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int row = (wm * TM + i) * 32 + (v & 3) + 8 * (v >> 2) + 4 * fh;
                    Cs[(kg * BM + row) * LDC + (wn * TN + j) * 32 + fr] = acc[i][j][v];
                }
        __syncthreads();
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
        // C tile value: the k groups' partial tiles are added in the fixed order kg = 0, 1, ...
        auto ctile = [&](int r) -> f32x4 {
            f32x4 v = *reinterpret_cast<const f32x4*>(Cs + r * LDC + c4 * 4);
#pragma unroll
            for (int k2 = 1; k2 < WK; ++k2) v += *reinterpret_cast<const f32x4*>(Cs + (k2 * BM + r) * LDC + c4 * 4);
            return v;
        };
        // bias + residual + ReLU + store of one float4 of output row m (the fused epilogue proper)
        auto finish = [&](f32x4 val, int m, f32x4 resv) {
            val += bias4;
            size_t dst;                 // element index into y
            if (p.out_mode == 0) {
                val += resv;
                dst = (size_t)m * p.y_ld + p.y_coff + n;
            } else {
                const int b = m / ohw;
                const int rem = m - b * ohw;
                const int oy = rem / p.OW, ox = rem - oy * p.OW;
                dst = (((size_t)b * 2 * p.OH + 2 * oy + (g >> 1)) * (2 * p.OW) + 2 * ox + (g & 1)) * p.cdec + co;
            }
            if (p.relu) {
                val[0] = apse_relu(val[0]); val[1] = apse_relu(val[1]); val[2] = apse_relu(val[2]); val[3] = apse_relu(val[3]);
            }
            if (vec_direct) {
                apse_st4(p.y, dst, val, p.y_st);
            } else {
                for (int k = 0; k < 4; ++k) if (n + k < p.Cout) apse_st1(p.y, dst + k, val[k], p.y_st);
            }
        };
        if (fast_ok) {
#pragma unroll
            for (int g0 = 0; g0 < PASSES; g0 += RG) {
                if (g0 > 0 && want_res) {
#pragma unroll
                    for (int i = 0; i < RG; ++i) rgrp[i] = fast_rload(g0 + i);
                }
#pragma unroll
                for (int i = 0; i < RG; ++i) {
                    f32x4 val = ctile(r0 + (g0 + i) * RPP);
                    val += bias4;
                    val += rgrp[i];
                    if (p.relu) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) asm("v_max_f32 %0, 0, %1" : "=v"(val[k]) : "v"(val[k]));      // x > 0 ? x : 0 (-0, NaN -> +0)
                    }
                    const int off = (int)(yo + (unsigned)(g0 + i) * ystep);
                    if (PR == 0 || p.y_st == 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, val), yrsrc, off, 0, APSE_NT ? 2 : 0);
                    else {
                        u32x2_t o;
                        if constexpr (PR == 2) {
                            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                            h4 t; t[0] = (_Float16)val[0]; t[1] = (_Float16)val[1]; t[2] = (_Float16)val[2]; t[3] = (_Float16)val[3];
                            o = __builtin_bit_cast(u32x2_t, t);
                        } else {
                            typedef __bf16 b4 __attribute__((ext_vector_type(4)));
                            b4 t; t[0] = (__bf16)val[0]; t[1] = (__bf16)val[1]; t[2] = (__bf16)val[2]; t[3] = (__bf16)val[3];
                            o = __builtin_bit_cast(u32x2_t, t);
                        }
                        __builtin_amdgcn_raw_buffer_store_b64(o, yrsrc, off, 0, APSE_NT ? 2 : 0);
                    }
                }
            }
        } else if (direct) {
            if (n < p.Cout) {
#pragma unroll
                for (int g0 = 0; g0 < PASSES; g0 += RG) {
                    if (g0 > 0 && want_res) {
#pragma unroll
                        for (int i = 0; i < RG; ++i) rgrp[i] = res_load(m0 + r0 + (g0 + i) * RPP);
                    }
#pragma unroll
                    for (int i = 0; i < RG; ++i) {
                        const int r = r0 + (g0 + i) * RPP;
                        const int m = m0 + r;
                        if (m < M) finish(ctile(r), m, rgrp[i]);
                    }
                }
            }
        } else {
            // split-K: this block's partial tile goes to its slab of the workspace
            if (n < p.Cout)
                for (int r = tid / C4; r < BM; r += RPP) {
                    const int m = m0 + r;
                    if (m >= M) break;
                    const f32x4 val = ctile(r);
                    float* dst = p.ws + ((size_t)z * p.M + m) * p.Cout + n;
                    if (vec_ws) *reinterpret_cast<f32x4*>(dst) = val;
                    else for (int k = 0; k < 4; ++k) if (n + k < p.Cout) dst[k] = val[k];
                }
            if (p.tile_cnt) {
                // In-launch reduction by the last-arriving K-slice of this tile (cdna_hip_programming.md, section 5
                // "In-launch split-K reduction"): plain slab stores -> every wave drains its stores -> barrier ->
                // one lane: agent-scope release, arrival ticket; the block that draws splitk-1 acquires and sums
                // the slabs in the fixed order z = 0..splitk-1 (bitwise reproducible, independent of arrival order).
                int* flag = reinterpret_cast<int*>(Cs + WK * BM * LDC);
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
                        finish(val, m, res_load(m));
                    }
                }
            }
        }
        __syncthreads();       // Cs is overwritten by the next tile's staging
    }
}

// Split-K second pass: fixed-order sum of the partial slabs (bitwise reproducible) + epilogue.
// The same, four columns per thread: plain NHWC output, no or same-resolution residual, Cout / y_ld / y_coff multiples of 4 (every
// split-K layer of the network).  Every element is the same sum in the same order (z ascending, + bias, + residual, ReLU) as in the
// scalar kernel below, which stays for the deconvolution scatter and the upsampled residual.
__global__ __launch_bounds__(256) void conv_splitk_reduce4(const ConvParams p) {
    int M = p.M;
    if (p.m_count) {
        int lim = (*p.m_count) * p.m_per_item;
        M = lim < M ? lim : M;
    }
    const unsigned c4 = (unsigned)p.Cout >> 2;
    const unsigned total = (unsigned)M * c4;
    for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
        const unsigned m = e / c4, n = (e - m * c4) << 2;
        const f32x4* src = reinterpret_cast<const f32x4*>(p.ws + (size_t)m * p.Cout + n);
        const size_t slab4 = ((size_t)p.M * p.Cout) >> 2;
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < p.splitk; ++z) val += src[z * slab4];
        if (p.bias) val += *reinterpret_cast<const f32x4*>(p.bias + n);
        if (p.res_mode == 1) val += apse_ld4(p.res, (size_t)m * p.Cout + n, p.res_st);
        if (p.relu) { val[0] = apse_relu(val[0]); val[1] = apse_relu(val[1]); val[2] = apse_relu(val[2]); val[3] = apse_relu(val[3]); }
        apse_st4(p.y, (size_t)m * p.y_ld + p.y_coff + n, val, p.y_st);
    }
}

__global__ __launch_bounds__(256) void conv_splitk_reduce(const ConvParams p) {
    int M = p.M;
    if (p.m_count) {
        int lim = (*p.m_count) * p.m_per_item;
        M = lim < M ? lim : M;
    }
    const size_t total = (size_t)M * p.Cout;
    const int ohw = p.OH * p.OW;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(e / p.Cout);
        const int n = (int)(e - (size_t)m * p.Cout);
        float val = 0.f;
        for (int z = 0; z < p.splitk; ++z) val += p.ws[((size_t)z * p.M + m) * p.Cout + n];
        int co = n, g = 0;
        if (p.out_mode == 1) { g = n / p.cdec; co = n - g * p.cdec; }
        if (p.bias) val += p.bias[co];
        auto ld1 = [&](size_t idx) -> float {
            if (p.res_st == 0) return p.res[idx];
            if (p.res_st == 1) return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(p.res)[idx] << 16);
            return (float)reinterpret_cast<const _Float16*>(p.res)[idx];
        };
        size_t dst;
        if (p.out_mode == 0) {
            if (p.res_mode == 1) {
                val += ld1((size_t)m * p.Cout + n);
            } else if (p.res_mode == 2) {
                const int b = m / ohw;
                const int rem = m - b * ohw;
                const int oy = rem / p.OW, ox = rem - oy * p.OW;
                const int hw2 = (p.OH >> 1) * (p.OW >> 1);
                val += ld1(((size_t)b * hw2 + (oy >> 1) * (p.OW >> 1) + (ox >> 1)) * p.Cout + n);
            }
            dst = (size_t)m * p.y_ld + p.y_coff + n;
        } else {
            const int b = m / ohw;
            const int rem = m - b * ohw;
            const int oy = rem / p.OW, ox = rem - oy * p.OW;
            const int dy = g >> 1, dx = g & 1;
            dst = (((size_t)b * 2 * p.OH + 2 * oy + dy) * (2 * p.OW) + 2 * ox + dx) * p.cdec + co;
        }
        if (p.relu) val = apse_relu(val);
        apse_st1(p.y, dst, val, p.y_st);
    }
}

template <int WM, int WN, int TM, int TN, int KS, int XT, int WK = 1, int PR = 0>
static int launch_cfg_x(const ConvParams& p, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int tiles = ((p.M + BM - 1) / BM) * ((p.Cout + BN - 1) / BN);
    const size_t lds_stage = (size_t)2 * KS * (BM + BN) * 32 * sizeof(float), lds_c = (size_t)WK * BM * (BN + 4) * sizeof(float) + 16;
    const size_t lds = lds_stage > lds_c ? lds_stage : lds_c;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f32<WM, WN, TM, TN, KS, XT, WK, PR>),
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
    if (ev0) hipEventRecord(ev0, s);
    hipLaunchKernelGGL((conv_igemm_f32<WM, WN, TM, TN, KS, XT, WK, PR>), dim3(grid_x, p.splitk), dim3(64 * WM * WN * WK), lds, s, p);
    if (ev1) hipEventRecord(ev1, s);
    if (p.splitk > 1 && !p.tile_cnt) {
        const size_t total = (size_t)p.M * p.Cout;
        const bool vec4 = p.out_mode == 0 && (p.res_mode == 0 || p.res_mode == 1) && ((p.Cout | p.y_ld | p.y_coff) & 3) == 0 && total < 0x7fffffffull;
        int blocks = (int)(((vec4 ? total / 4 : total) + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        if (vec4) hipLaunchKernelGGL(conv_splitk_reduce4, dim3(blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(conv_splitk_reduce, dim3(blocks), dim3(256), 0, s, p);
    }
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}

// the descriptor path needs f32 activations of < 4 GiB (byte offsets are 32-bit)
static bool descriptor_ok(const ConvParams& p) { return p.x_st == 0 && (((size_t)p.B * p.H * p.W) << p.cin_log2) * 4 < 0xfffffff0ull; }

template <int WM, int WN, int TM, int TN, int KS>
static int launch_cfg(const ConvParams& p, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    return descriptor_ok(p) ? launch_cfg_x<WM, WN, TM, TN, KS, 0>(p, s, ev0, ev1) : launch_cfg_x<WM, WN, TM, TN, KS, 1>(p, s, ev0, ev1);
}
// 8-wave shapes exist for the descriptor path only; anything else falls back to the 4-wave shape of the same tile
template <int WM, int WN, int TM, int TN, int KS, int KSF>
static int launch_cfg_k2(const ConvParams& p, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    return descriptor_ok(p) ? launch_cfg_x<WM, WN, TM, TN, KS, 0, 2>(p, s, ev0, ev1) : launch_cfg_x<WM, WN, TM, TN, KSF, 1>(p, s, ev0, ev1);
}

// 16-bit operands already stored in the operand type, filter rows a whole number of 64-element steps: the scheduled kernel's
// shape for tile config `cfg` (0 128x128, 1 64x64, 3 128x64, 6 64x64 8 waves, 8 256x128), or -1 (not eligible: refused)
static int fast16_shape(const ConvParams& p, int cfg) {
    const bool fast16 = (p.prec == 1 || p.prec == 2) && p.w16 && p.x_st == p.prec && (p.KWCp & 63) == 0 && p.cin_log2 >= 3 &&
                        ((((size_t)p.B * p.H * p.W) << p.cin_log2) * 2 < 0xfffffff0ull) && !p.tile_cnt;
    if (!fast16) return -1;
    int c16 = (cfg == 4 || cfg == 5 || cfg == 2) ? 1 : (cfg == 7 ? 6 : cfg);
    // deep-K layers with at least one 256x128 tile per CU: the wider tile halves the filter traffic per FLOP
    // (out2 / rpn_t2 / res4 3x3 at batch 8: +5..7 %)
    // (K >= 512 since the LDS-DMA kernel took over this tile: lateral 3 / res3 conv1 / res4 shortcut +3..9 %; K >= 1024 before)
    if (c16 == 0 && p.steps_total >= 16 && ((p.M + 255) / 256) * ((p.Cout + 127) / 128) >= 250) c16 = 8;
    return c16;
}

int apse_conv_effective_cfg(const ConvParams& p, int cfg) {
    // an explicit request for a special kernel is honoured only when the layer is eligible (-1 otherwise: an error, never a
    // silent launch on a shape the kernel does not handle)
    if (cfg == APSE_CFG_STREAM) return (!p.no_stream && apse_conv1x1_stream_ok(p)) ? APSE_CFG_STREAM : -1;
    if (cfg == APSE_CFG_GLDS) return apse_conv_glds16_ok(p) ? APSE_CFG_GLDS : -1;
    if (cfg == APSE_CFG_SKINNY) return apse_conv_skinny_ok(p) ? APSE_CFG_SKINNY : -1;
    if (!p.no_stream && apse_conv_skinny_ok(p)) return APSE_CFG_SKINNY;
    if (!p.no_stream && apse_conv1x1_stream_ok(p)) return APSE_CFG_STREAM;
    // wherever the 16-bit path would take the register-staged 256x128 tile, the LDS-DMA kernel of the same tile runs instead
    if (!p.no_stream && apse_conv_glds16_ok(p)) {
        const int c16 = fast16_shape(p, cfg);
        if (c16 == 8) return APSE_CFG_GLDS;                                  // (a caller forcing cfg 8 keeps the register-staged tile)
        // round 4: deep-K layers with about ONE 128x128 tile per CU (res4 at batch 4, res5 at batch 8, fc1: 200..320 tiles) take the
        // LDS-DMA kernel's 128x128 tile instead of the register-staged 128x64 / 128x128 ones
        const int t128 = ((p.M + 127) / 128) * ((p.Cout + 127) / 128);
        if ((c16 == 0 || c16 == 3) && p.steps_total >= 16 && t128 >= 200 && t128 <= 320 && apse_conv_glds16_small(p)) return APSE_CFG_GLDS;
    }
    return cfg;
}

int apse_launch_conv(const ConvParams& p, int cfg, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (p.M <= 0 || p.Cout <= 0 || p.steps_total <= 0) return APSE_E_INVALID;
    if (p.res_mode != 0 && (p.Cout & 3) != 0) return APSE_E_INVALID;      // residual rows are read as float4
    const int eff = apse_conv_effective_cfg(p, cfg);
    if (eff < 0) return APSE_E_INVALID;                                   // a streaming kernel asked for, layer not eligible
    if (eff == APSE_CFG_STREAM) return apse_launch_conv1x1_stream(p, s, ev0, ev1);
    if (eff == APSE_CFG_GLDS) return apse_launch_conv_glds16(p, s, ev0, ev1);
    if (eff == APSE_CFG_SKINNY) return apse_launch_conv_skinny(p, s, ev0, ev1);
    if (p.prec == 1 || p.prec == 2) {
        const int c16 = fast16_shape(p, cfg);
        if (c16 >= 0) {
            int rc = APSE_E_INVALID;
            if (p.prec == 1) {
                if (c16 == 0) rc = launch_cfg_x<2, 2, 2, 2, 1, 0, 1, 1>(p, s, ev0, ev1);
                else if (c16 == 1) rc = launch_cfg_x<2, 2, 1, 1, 2, 0, 1, 1>(p, s, ev0, ev1);
                else if (c16 == 3) rc = launch_cfg_x<4, 1, 1, 2, 1, 0, 1, 1>(p, s, ev0, ev1);
                else if (c16 == 6) rc = launch_cfg_x<2, 2, 1, 1, 2, 0, 2, 1>(p, s, ev0, ev1);
                else if (c16 == 8) rc = launch_cfg_x<4, 2, 2, 2, 1, 0, 1, 1>(p, s, ev0, ev1);      // 256x128, 8 waves
            } else {
                if (c16 == 0) rc = launch_cfg_x<2, 2, 2, 2, 1, 0, 1, 2>(p, s, ev0, ev1);
                else if (c16 == 1) rc = launch_cfg_x<2, 2, 1, 1, 2, 0, 1, 2>(p, s, ev0, ev1);
                else if (c16 == 3) rc = launch_cfg_x<4, 1, 1, 2, 1, 0, 1, 2>(p, s, ev0, ev1);
                else if (c16 == 6) rc = launch_cfg_x<2, 2, 1, 1, 2, 0, 2, 2>(p, s, ev0, ev1);
                else if (c16 == 8) rc = launch_cfg_x<4, 2, 2, 2, 1, 0, 1, 2>(p, s, ev0, ev1);
            }
            return rc;              // launch_cfg_x adds the split-K reduce pass itself
        }
        // 16-bit operands need 16-bit STORED activations, filter rows of whole 64-element steps and the separate reduce pass
        return APSE_E_INVALID;
    }
    switch (cfg) {
        case 0: return launch_cfg<2, 2, 2, 2, 1>(p, s, ev0, ev1);
        case 1: return launch_cfg<2, 2, 1, 1, 2>(p, s, ev0, ev1);
        case 2: {                                                        // the Cout <= 32 heads: also fed by 16-bit activations
            const bool small = (((size_t)p.B * p.H * p.W) << p.cin_log2) * 4 < 0xfffffff0ull;
            if (small && p.x_st == 1 && p.cin_log2 >= 2) return launch_cfg_x<4, 1, 1, 1, 2, 2>(p, s, ev0, ev1);
            if (small && p.x_st == 2 && p.cin_log2 >= 2) return launch_cfg_x<4, 1, 1, 1, 2, 3>(p, s, ev0, ev1);
            return launch_cfg<4, 1, 1, 1, 2>(p, s, ev0, ev1);
        }
        case 3: return launch_cfg<4, 1, 1, 2, 1>(p, s, ev0, ev1);
        case 4: return launch_cfg<2, 2, 1, 1, 1>(p, s, ev0, ev1);     // 64x64, 32-deep steps: 32 KB of LDS, up to 4-5 blocks per CU
        case 5: return launch_cfg<4, 1, 1, 1, 1>(p, s, ev0, ev1);     // 128x32, 32-deep steps
        case 6: return launch_cfg_k2<2, 2, 1, 1, 2, 2>(p, s, ev0, ev1);  // 64x64, 8 waves: two k groups, 64-deep steps
        case 7: return launch_cfg_k2<2, 2, 1, 1, 4, 2>(p, s, ev0, ev1);  // 64x64, 8 waves, 128-deep steps
        case 8: return launch_cfg<2, 2, 2, 2, 1>(p, s, ev0, ev1);        // (256x128 exists for 16-bit operands only)
        default: return APSE_E_INVALID;
    }
}

// Tile/split heuristic (tools/conv_sweep.py on MI355X): fill >= ~2 resident blocks per CU; prefer the
// largest tile that does; for small-M layers trade tile size against split-K:
//   very large K -> 128x128 tiles with the K range split across blocks (fc1),
//   medium K  -> 64x64 tiles (unsplit once there are >= 192 of them: res4 3x3 / 1x1); with about one tile per CU
//                the 8-wave two-k-group shape (cfg 6), which puts two waves on each SIMD,
//   tiny K, wide N (res4 conv3) -> 128x64 tiles.
int apse_conv_pick_cfg(int M, int Cout, int steps, int* splitk) {
    *splitk = 1;
    auto tiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((Cout + bn - 1) / bn); };
    auto split_for = [&](int t, int min_steps_per_split) {
        int sk = (512 + t / 2) / t;
        if (sk > steps / min_steps_per_split) sk = steps / min_steps_per_split;
        if (sk > 64) sk = 64;
        return sk < 1 ? 1 : sk;
    };
    if (Cout <= 32) {
        // narrow heads over few rows (box predictor: 1000 rows x K = 1024 is 8 tiles of 128 rows): split K so that the launch
        // has ~64 blocks instead of one long k loop on 8 CUs (28.7 -> ~14 us incl. the reduce pass)
        const int t = tiles(128, 32);
        if (t < 32 && steps >= 16) {
            int sk = 64 / t;
            if (sk > steps / 4) sk = steps / 4;
            if (sk > 8) sk = 8;
            *splitk = sk < 1 ? 1 : sk;
        }
        return 2;
    }
    if (Cout <= 64) return tiles(128, 64) >= 192 ? 3 : 1;
    const int t128 = tiles(128, 128);
    if (t128 >= 224) {
        if (t128 < 384 && tiles(128, 64) >= 384) return 3;      // ~one 128x128 tile per CU: 128x64 gives every CU two blocks
        return 0;
    }
    if (t128 >= 32 && steps >= 256) {                            // fc1: K = 12544
        *splitk = split_for(t128, 8);
        return 0;
    }
    const int t64 = tiles(64, 64);
    if (t64 >= 192) return t64 <= 320 ? 6 : 1;     // about one 64x64 tile per CU: the 8-wave shape keeps two waves on every SIMD
    if (steps >= 16) {
        // 8-wave blocks, K split so that there is about one block per CU (each slice keeps >= 8 steps)
        int sk = (256 + t64 / 2) / t64;
        if (t64 > 64 && sk > 2) sk = 2;            // 65..191 tiles: two slices measured best (mask head, res5)
        if (sk > steps / 8) sk = steps / 8;
        if (sk > 64) sk = 64;
        *splitk = sk < 1 ? 1 : sk;
        return 6;
    }
    return 1;
}
