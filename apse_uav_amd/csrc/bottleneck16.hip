// bottleneck64_fused16 (gfx950, 16-bit storage modes): a whole res2 bottleneck -- conv1 1x1 (Cin -> 64) + FrozenBN + ReLU,
// conv2 3x3 (64 -> 64) + FrozenBN + ReLU, conv3 1x1 (64 -> 256) + FrozenBN + residual + ReLU -- in ONE kernel
// (detectron2 BottleneckBlock, reached from /root/reference/dcnn/networks/track_rcnn.py:42; SURVEY.md section 7 step 5
// "bottleneck fusion").
//
// Why.  At 3840x2160 the res2 maps are [B][192][336][.]: per image 33 MB for a 256-channel tensor and 8 MB for a 64-channel
// one in 16 bits.  The three-kernel form moves, per bottleneck and image, x (33) + t1 (8) | t1 (8) + t2 (8) | t2 (8) +
// residual (33) + y (33) = 132 MB through HBM and is bandwidth- / latency-bound in every launch (fp16, batch 8: 85 + 104 +
// 113 us).  Here a block owns an 8 x 16 tile of output pixels: the 10 x 18 halo of conv1 outputs and the conv2 outputs of the
// tile live in LDS only; HBM sees x (1.41 x: the halo is recomputed per tile, neighbouring tiles of an XCD share it in L2),
// the residual rows (an L2 hit when the residual is x itself) and y.
//
// Arithmetic = the three-kernel form's, bit for bit: the same v_mfma_f32_32x32x16 per 16 input channels in the same k order
// per accumulator (conv1: channels ascending; conv2: taps (ky, kx) row-major, 64 channels each; conv3: channels ascending),
// f32 bias add, ReLU, ONE rounding to the storage type where the three-kernel form stores t1 / t2 / y; conv2 sees zeros for
// halo pixels outside the map (they are zero padding of t1, NOT conv1 of a zero pixel); residual added in f32 after the bias.
//
// Block = 4 waves, two blocks per CU.  Phase 1 (conv1 on the 180 -> 192 halo rows = 6 row tiles of 32): wave w takes row tile
// w for both 32-column tiles and one column tile of row tile 4 + (w & 1); A strips global -> registers (each halo pixel's
// Cin channels are one contiguous run), filters through the weight stream.  Phase 2 (conv2): wave w owns output rows
// 32 w .. + 31; A fragments are 16-byte reads of the t1 tile at (pixel + tap).  Phase 3 (conv3): the wave's 32 x 64 strip of
// t2 is its A operand, N = 256 in two chunks of 128 columns with the streaming kernel's wave-private epilogue (transpose tile,
// residual rows requested before the accumulators go through LDS, whole 128-byte segments out).
// Weight stream: five stages per tile -- conv1's whole filter (Cin / 64 slices of 64 x 128 B), conv2's taps three at a time
// (3 x 64 x 128 B), conv3's whole filter (256 x 128 B) -- through ONE 32 KB XOR-swizzled LDS stage shared by the four waves; a
// stage is fetched global -> registers while the previous one computes and handed to LDS between two barriers.  (First form of
// this kernel: 15 stages of 8 KB, double-buffered, one barrier each -- with 0.1-0.3 us of MFMA work per stage every stage waited
// out an L2 round trip: 28 us per tile.)  The sequence is the same for every tile, so the stream runs across tile seams.
#include "apse_common.h"
#include <type_traits>

#define BN_TH 8
#define BN_TW 16
#define BN_HW (BN_TW + 2)              // 18 halo columns
#define BN_HROWS ((BN_TH + 2) * BN_HW)  // 180 halo pixels
#define BN_CLD 68                      // row stride (floats) of the wave-private transpose tile [32][64]

struct BneckParams {
    const uint16_t* x;        // [B][H][W][K1]
    const uint16_t* res;      // [B][H][W][256]: residual of conv3 (x itself, or the projection shortcut's output)
    uint16_t* y;              // [B][H][W][256]
    const uint16_t *w1, *w2, *w3;      // packed filters: [>= 64][K1], [>= 64][3][192], [256][64]
    const float *b1, *b2, *b3;
    int B, H, W;
    int tiles_y, tiles_x;
};

typedef _Float16 bn_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 bn_mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 bn_mfma(bn_f16x8 a, bn_f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <int PR>
__device__ __forceinline__ uint16_t bn_round(float v) {
    if constexpr (PR == 1) return __builtin_bit_cast(uint16_t, (__bf16)v);
    else return __builtin_bit_cast(uint16_t, (_Float16)v);
}

// 16-byte slot of logical slot `ls` in 128-byte row `row` (the swizzle of conv_igemm / conv1x1_stream)
__device__ __forceinline__ int bn_slot(int row, int ls) { return row * 128 + ((ls ^ ((row >> 1) & 7)) << 4); }

template <int PR, int K1>
__global__ __launch_bounds__(256, 2) void bottleneck64_fused16(const BneckParams p) {
    typedef typename std::conditional<PR == 1, bf16x8, bn_f16x8>::type op8;
    constexpr int NS1 = K1 / 64;                  // conv1 filter slices
    constexpr int GT = 5;                         // weight stages per tile: conv1 | taps 0-2 | taps 3-5 | taps 6-8 | conv3
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ws = smem;                              // [256 rows][128 B] weight stage (32 KB)
    char* t1 = smem + 256 * 128;                  // [192][128 B] conv1 outputs of the halo tile (24 KB)
    char* t2 = t1 + 192 * 128;                    // [128][128 B] conv2 outputs of the tile (16 KB)
    float* Cw = reinterpret_cast<float*>(t1);     // phase-3 epilogue: [4 waves][32][BN_CLD] f32 (34.8 KB) over t1 + t2 (dead by then)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr0 = lane & 31, fh0 = lane >> 5;
    const int srow = tid >> 3, slot = tid & 7;
    // per-lane bases of the t1 / t2 element stores (see the phase-1 epilogue): [(v >> 1) & 1]
    int eo[2];
    eo[0] = (((fr0 >> 3) ^ (fh0 << 1)) << 4) + (fr0 & 7) * 2 + fh0 * 512;
    eo[1] = (((fr0 >> 3) ^ (1 | (fh0 << 1))) << 4) + (fr0 & 7) * 2 + fh0 * 512;
    const int tiles_img = p.tiles_y * p.tiles_x;
    const int tiles = p.B * tiles_img;
    const int my_tiles = ((int)blockIdx.x < tiles) ? (tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (my_tiles == 0) return;
    const int G = my_tiles * GT;

    // ---- weight stream.  Q = position of a stage inside its tile (compile time).  Stage rows: conv1: slice * 64 + n; conv2:
    // (tap % 3) * 64 + n; conv3: n.  Filters are read through buffer descriptors: per thread a row offset, the rest constants.
    constexpr int NW = NS1 * 2 > 8 ? NS1 * 2 : 8;              // 16-byte pieces per thread of the largest stage
    f32x4 wr[NW];
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w1), 0, 64 * K1 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w2), 0, 64 * 576 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w3), 0, 256 * 64 * 2, 0x00020000);
    const unsigned map_bytes = (unsigned)(p.B * p.H * p.W) * 512u;      // residual and output maps: [B][H][W][256] 16-bit
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.res), 0, (int)map_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)map_bytes, 0x00020000);
    auto w_fetch = [&](auto QC) {
        constexpr int Q = decltype(QC)::value;
        if constexpr (Q == 0) {
#pragma unroll
            for (int i = 0; i < 2 * NS1; ++i)
                wr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, (srow + 32 * (i & 1)) * (K1 * 2) + slot * 16, (i >> 1) * 128, 0));
        } else if constexpr (Q < 4) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                constexpr int T0 = 3 * (Q - 1);
                const int t = T0 + (i >> 1);
                wr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs2, (srow + 32 * (i & 1)) * (576 * 2) + slot * 16,
                                                                                         ((t / 3) * 192 + (t % 3) * 64) * 2, 0));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                wr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs3, (srow + 32 * i) * 128 + slot * 16, 0, 0));
        }
    };
    auto w_store = [&](auto QC) {
        constexpr int Q = decltype(QC)::value;
        constexpr int n = Q == 0 ? 2 * NS1 : (Q < 4 ? 6 : 8);
#pragma unroll
        for (int i = 0; i < n; ++i) *reinterpret_cast<f32x4*>(Ws + bn_slot(srow + 32 * i, slot)) = wr[i];
    };
    w_fetch(std::integral_constant<int, 0>{});
    int g = 0;
    // one stage (position Q of its tile; its filters are in `wr`): barrier (the previous stage's readers are done) -> registers to
    // LDS -> request the next stage -> barrier -> `body()`
    auto stage = [&](auto QC, auto body) {
        constexpr int Q = decltype(QC)::value;
        __syncthreads();
        w_store(std::integral_constant<int, Q>{});
        if (g + 1 < G) w_fetch(std::integral_constant<int, (Q + 1) % GT>{});
        __syncthreads();
        body();
        ++g;
    };

    __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0,
                                                                    (int)((unsigned)(p.B * p.H * p.W) * (unsigned)(K1 * 2)), 0x00020000);
    const int rt1 = 4 + (wave & 1), nt1 = wave >> 1;          // the wave's extra unit of phase 1: row tile rt1, column tile nt1
    float* cw = Cw + wave * 32 * BN_CLD;
    const int erow = lane >> 3, ecol = (lane & 7) * 8;        // phase-3 epilogue: lane -> (row erow + 8 i, columns ecol .. + 7)

    for (int it = 0; it < my_tiles; ++it) {
        // opaque copies: the LDS / fragment addresses below are tile-invariant, and hoisted out of this loop they would occupy
        // ~100 registers for its whole duration; recomputed per tile they cost a few VALU instructions
        int fr = fr0, fh = fh0;
        asm volatile("" : "+v"(fr), "+v"(fh));
        // XCD-aware order: the blocks of one XCD (blockIdx & 7) walk a contiguous range of tiles (they share halos in its L2)
        int tile;
        {
            const int wg = blockIdx.x + it * gridDim.x;
            const int q = tiles >> 3, r = tiles & 7, xcd = wg & 7, idx = wg >> 3;
            tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        const int b = tile / tiles_img;
        const int tr = tile - b * tiles_img;
        const int tyi = tr / p.tiles_x, txi = tr - tyi * p.tiles_x;
        const int oy0 = tyi * BN_TH, ox0 = txi * BN_TW;

        // ================================================================== phase 1: conv1 on the halo rows
        f32x4 a0[NS1][4], a1[NS1][4];
        unsigned vm0, vm1;                                   // bit r: halo row 32 rt + r is a pixel of the map
        {
            auto row_off = [&](int r, bool& valid) -> unsigned {
                const int hy = r / BN_HW, hx = r - hy * BN_HW;
                const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
                valid = r < BN_HROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                return valid ? (unsigned)((b * p.H + iy) * p.W + ix) * (unsigned)(K1 * 2) : 0xfffffff0u;
            };
            bool v0, v1;
            const unsigned o0 = row_off(wave * 32 + fr, v0), o1 = row_off(rt1 * 32 + fr, v1);
            vm0 = (unsigned)__ballot(v0 && fh == 0);
            vm1 = (unsigned)__ballot(v1 && fh == 0);
#pragma unroll
            for (int s = 0; s < NS1; ++s)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned ko = (unsigned)((s * 64 + (2 * c + fh) * 8) * 2);
                    a0[s][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(v0 ? o0 + ko : 0xfffffff0u), 0, 0));
                    a1[s][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(v1 ? o1 + ko : 0xfffffff0u), 0, 0));
                }
        }
        f32x16 acc[3];                                        // [0], [1]: row tile `wave`, column tiles 0 / 1; [2]: the extra unit
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
        stage(std::integral_constant<int, 0>{}, [&]() {
#pragma unroll
            for (int s = 0; s < NS1; ++s)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int ls = 2 * c + fh;
                    const f32x4 bf0 = *reinterpret_cast<const f32x4*>(Ws + bn_slot(s * 64 + fr, ls));
                    const f32x4 bf1 = *reinterpret_cast<const f32x4*>(Ws + bn_slot(s * 64 + 32 + fr, ls));
                    acc[0] = bn_mfma(__builtin_bit_cast(op8, a0[s][c]), __builtin_bit_cast(op8, bf0), acc[0]);
                    acc[1] = bn_mfma(__builtin_bit_cast(op8, a0[s][c]), __builtin_bit_cast(op8, bf1), acc[1]);
                    acc[2] = bn_mfma(__builtin_bit_cast(op8, a1[s][c]), __builtin_bit_cast(op8, nt1 ? bf1 : bf0), acc[2]);
                }
        });
        {
            // bias, ReLU, zero outside the map, one rounding -> t1
            const float bi0 = p.b1[fr], bi1 = p.b1[32 + fr];
            const float bi2 = nt1 ? bi1 : bi0;
            // element (row R0 + rl, column L * 8 + fr) of the swizzled tile, rl = (v & 3) + 8 (v >> 2) + 4 fh: with R0 a multiple of 32 the
            // swizzle term (row >> 1) & 7 has the bits ((v >> 1) & 1, fh, (v >> 2) & 1), so the byte address is one of two per-lane bases
            // (eo[(v >> 1) & 1], set up once per block) plus a compile-time immediate -- no address arithmetic per element.
            char* const w3 = t1 + rt1 * 4096;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int rl = (v & 3) + 8 * (v >> 2) + 4 * fh;
                float u0 = acc[0][v] + bi0, u1 = acc[1][v] + bi1, u2 = acc[2][v] + bi2;
                u0 = apse_relu(u0); u1 = apse_relu(u1); u2 = apse_relu(u2);
                if (!((vm0 >> rl) & 1u)) { u0 = 0.f; u1 = 0.f; }
                if (!((vm1 >> rl) & 1u)) u2 = 0.f;
                const int imm = (v & 3) * 128 + (v >> 2) * 1024, gp = (v >> 2) & 1, eb = eo[(v >> 1) & 1];
                *reinterpret_cast<uint16_t*>(t1 + wave * 4096 + eb + imm + (gp << 6)) = bn_round<PR>(u0);
                *reinterpret_cast<uint16_t*>(t1 + wave * 4096 + eb + imm + ((gp ^ 1) << 6)) = bn_round<PR>(u1);
                *reinterpret_cast<uint16_t*>(w3 + eb + imm + ((gp ^ nt1) << 6)) = bn_round<PR>(u2);
            }
        }
        // (the first barrier of the next stage is also "the halo tile of conv1 outputs is complete")

        // residual rows of the tile (16 pieces of 16 B per lane: rows erow + 8 i, 8 channels of each 64-column chunk): requested
        // here so that they arrive under phase 2 -- fetched chunk by chunk inside phase 3, each exposed a full L2 round trip
        f32x4 rr[4][4];
        unsigned pixb[4];                         // BYTE offsets into res / y: B * H * W * 512 < 0x70000000 (checked by the launcher); a pixel
                                                  // outside the map gets an offset past the descriptors' range (loads give zeros, stores are dropped)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = wave * 32 + erow + 8 * i;
            const int oy = oy0 + (q >> 4), ox = ox0 + (q & 15);
            pixb[i] = (oy < p.H && ox < p.W) ? ((unsigned)((b * p.H + oy) * p.W + ox) * 256u + (unsigned)ecol) * 2u : 0x80000000u;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
                rr[ch][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, (int)(pixb[i] + ch * 128u), 0, 0));       // 8 x 16 bit
        }

        // ================================================================== phase 2: conv2 3x3 from the t1 tile
        const int q2 = wave * 32 + fr;                        // output pixel of this lane's A row: (q2 / 16, q2 % 16)
        const int hbase = (q2 >> 4) * BN_HW + (q2 & 15);      // halo row of tap (0, 0)
#pragma unroll
        for (int v = 0; v < 16; ++v) { acc[0][v] = 0.f; acc[1][v] = 0.f; }
        auto p2 = [&](auto QC) {
            constexpr int Q = decltype(QC)::value;                // stage 1 .. 3: taps 3 (Q - 1) .. + 2
            stage(std::integral_constant<int, Q>{}, [&]() {
#pragma unroll
                for (int tl = 0; tl < 3; ++tl) {
                    constexpr int T0 = 3 * (Q - 1);
                    const int t = T0 + tl;
                    const int hr = hbase + (t / 3) * BN_HW + (t % 3);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int ls = 2 * c + fh;
                        const f32x4 af = *reinterpret_cast<const f32x4*>(t1 + bn_slot(hr, ls));
                        const f32x4 bf0 = *reinterpret_cast<const f32x4*>(Ws + bn_slot(tl * 64 + fr, ls));
                        const f32x4 bf1 = *reinterpret_cast<const f32x4*>(Ws + bn_slot(tl * 64 + 32 + fr, ls));
                        acc[0] = bn_mfma(__builtin_bit_cast(op8, af), __builtin_bit_cast(op8, bf0), acc[0]);
                        acc[1] = bn_mfma(__builtin_bit_cast(op8, af), __builtin_bit_cast(op8, bf1), acc[1]);
                    }
                }
            });
        };
        p2(std::integral_constant<int, 1>{}); p2(std::integral_constant<int, 2>{}); p2(std::integral_constant<int, 3>{});
        {
            const float bi0 = p.b2[fr], bi1 = p.b2[32 + fr];
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float u0 = acc[0][v] + bi0, u1 = acc[1][v] + bi1;
                u0 = apse_relu(u0); u1 = apse_relu(u1);
                const int imm = (v & 3) * 128 + (v >> 2) * 1024, gp = (v >> 2) & 1, eb = eo[(v >> 1) & 1];
                *reinterpret_cast<uint16_t*>(t2 + wave * 4096 + eb + imm + (gp << 6)) = bn_round<PR>(u0);
                *reinterpret_cast<uint16_t*>(t2 + wave * 4096 + eb + imm + ((gp ^ 1) << 6)) = bn_round<PR>(u1);
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the wave's own t2 rows are written

        // ================================================================== phase 3: conv3 + residual
        f32x4 a3[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) a3[c] = *reinterpret_cast<const f32x4*>(t2 + bn_slot(wave * 32 + fr, 2 * c + fh));
        // (the first barrier of the conv3 stage is also "every wave holds its strip": t1 / t2 then become the transpose tiles)
        auto p3 = [&](auto CC) {
            constexpr int ch = decltype(CC)::value;               // 64-column chunk of the 256 output channels
            const int nb = ch * 64 + ecol;
            f32x16 ac3[2];
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // C operand of the first MFMAs
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ls = 2 * c + fh;
                const f32x4 bf0 = *reinterpret_cast<const f32x4*>(Ws + bn_slot(ch * 64 + fr, ls));
                const f32x4 bf1 = *reinterpret_cast<const f32x4*>(Ws + bn_slot(ch * 64 + 32 + fr, ls));
                ac3[0] = bn_mfma(__builtin_bit_cast(op8, a3[c]), __builtin_bit_cast(op8, bf0), c == 0 ? zero16 : ac3[0]);
                ac3[1] = bn_mfma(__builtin_bit_cast(op8, a3[c]), __builtin_bit_cast(op8, bf1), c == 0 ? zero16 : ac3[1]);
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    cw[((v & 3) + 8 * (v >> 2) + 4 * fh) * BN_CLD + jj * 32 + fr] = ac3[jj][v];
            __builtin_amdgcn_wave_barrier();
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.b3 + nb), b1 = *reinterpret_cast<const f32x4*>(p.b3 + nb + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = erow + 8 * i;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(cw + r * BN_CLD + ecol) + b0;
                f32x4 v1 = *reinterpret_cast<const f32x4*>(cw + r * BN_CLD + ecol + 4) + b1;
                const f32x4 raw = rr[ch][i];
                f32x4 x0, x1;
                if constexpr (PR == 1) {
                    const unsigned q0 = __float_as_uint(raw[0]), q1 = __float_as_uint(raw[1]), q2b = __float_as_uint(raw[2]), q3 = __float_as_uint(raw[3]);
                    x0 = f32x4{__uint_as_float(q0 << 16), __uint_as_float(q0 & 0xffff0000u), __uint_as_float(q1 << 16), __uint_as_float(q1 & 0xffff0000u)};
                    x1 = f32x4{__uint_as_float(q2b << 16), __uint_as_float(q2b & 0xffff0000u), __uint_as_float(q3 << 16), __uint_as_float(q3 & 0xffff0000u)};
                } else {
                    const bn_f16x8 hv = __builtin_bit_cast(bn_f16x8, raw);
                    x0 = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
                    x1 = f32x4{(float)hv[4], (float)hv[5], (float)hv[6], (float)hv[7]};
                }
                v0 += x0; v1 += x1;
#pragma unroll
                for (int k = 0; k < 4; ++k) { v0[k] = apse_relu(v0[k]); v1[k] = apse_relu(v1[k]); }
                typedef unsigned bn_u32x4 __attribute__((ext_vector_type(4)));
                bn_u32x4 ob;
                if constexpr (PR == 1) {
                    bf16x8 o;
                    o[0] = (__bf16)v0[0]; o[1] = (__bf16)v0[1]; o[2] = (__bf16)v0[2]; o[3] = (__bf16)v0[3];
                    o[4] = (__bf16)v1[0]; o[5] = (__bf16)v1[1]; o[6] = (__bf16)v1[2]; o[7] = (__bf16)v1[3];
                    ob = __builtin_bit_cast(bn_u32x4, o);
                } else {
                    bn_f16x8 o;
                    o[0] = (_Float16)v0[0]; o[1] = (_Float16)v0[1]; o[2] = (_Float16)v0[2]; o[3] = (_Float16)v0[3];
                    o[4] = (_Float16)v1[0]; o[5] = (_Float16)v1[1]; o[6] = (_Float16)v1[2]; o[7] = (_Float16)v1[3];
                    ob = __builtin_bit_cast(bn_u32x4, o);
                }
                __builtin_amdgcn_raw_buffer_store_b128(ob, yrsrc, (int)(pixb[i] + ch * 128u), 0, APSE_NT ? 2 : 0);
            }
            __builtin_amdgcn_wave_barrier();
        };
        stage(std::integral_constant<int, 4>{}, [&]() {
            p3(std::integral_constant<int, 0>{}); p3(std::integral_constant<int, 1>{});
            p3(std::integral_constant<int, 2>{}); p3(std::integral_constant<int, 3>{});
        });
    }
}

// Eligible: 16-bit storage (x, residual, y and the filters in the operand type), 64 mid channels, 256 output channels,
// Cin = 64 or 256, stride 1, input below 4 GiB, residual / output maps below 1.75 GiB.  Returns APSE_E_INVALID otherwise (the caller keeps the three-kernel form).
extern "C" int apse_k_bottleneck64_fused16(const void* x, const void* res, void* y, const uint16_t* w1, const float* b1,
                                           const uint16_t* w2, const float* b2, const uint16_t* w3, const float* b3, int B, int H,
                                           int W, int K1, int prec, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if ((prec != 1 && prec != 2) || (K1 != 64 && K1 != 256) || B < 1 || H < 1 || W < 1) return APSE_E_INVALID;
    if ((size_t)B * H * W * K1 * 2 >= 0xfffffff0ull || (size_t)B * H * W * 512 >= 0x70000000ull) return APSE_E_INVALID;   // (byte offsets; 0x80000000 marks a dead row)
    BneckParams p;
    p.x = reinterpret_cast<const uint16_t*>(x); p.res = reinterpret_cast<const uint16_t*>(res); p.y = reinterpret_cast<uint16_t*>(y);
    p.w1 = w1; p.w2 = w2; p.w3 = w3; p.b1 = b1; p.b2 = b2; p.b3 = b3;
    p.B = B; p.H = H; p.W = W;
    p.tiles_y = (H + BN_TH - 1) / BN_TH; p.tiles_x = (W + BN_TW - 1) / BN_TW;
    const long tiles = (long)B * p.tiles_y * p.tiles_x;
    const int blocks = tiles < 512 ? (int)tiles : 512;             // two per CU, persistent over the tiles
    const size_t lds = 256 * 128 + 192 * 128 + 128 * 128;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck64_fused16<1, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck64_fused16<1, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck64_fused16<2, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck64_fused16<2, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    if (ev0) hipEventRecord(ev0, s);
    if (prec == 1 && K1 == 64) hipLaunchKernelGGL((bottleneck64_fused16<1, 64>), dim3(blocks), dim3(256), lds, s, p);
    else if (prec == 1) hipLaunchKernelGGL((bottleneck64_fused16<1, 256>), dim3(blocks), dim3(256), lds, s, p);
    else if (K1 == 64) hipLaunchKernelGGL((bottleneck64_fused16<2, 64>), dim3(blocks), dim3(256), lds, s, p);
    else hipLaunchKernelGGL((bottleneck64_fused16<2, 256>), dim3(blocks), dim3(256), lds, s, p);
    if (ev1) hipEventRecord(ev1, s);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
