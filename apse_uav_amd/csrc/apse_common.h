// Shared declarations for the apse_uav MI355X (gfx950) hot-path library.
// Internal header: the public C-ABI is include/apse_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define APSE_OK 0
#define APSE_E_INVALID (-1)
#define APSE_E_HIP (-2)
#define APSE_E_NOMEM (-3)
#define APSE_E_STATE (-4)
#define APSE_E_MISSING (-5)

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---------------------------------------------------------------- convolution as implicit GEMM
// Activations are NHWC f32 with C a power of two >= 4.  Weights are [Cout_p][KH][KWCp] where the
// (kw, cin) run of one filter row is contiguous (matches the NHWC run of KW pixels) and padded
// with zeros to a multiple of 32 floats; Cout_p = Cout rounded up to 128 (zero rows).
struct ConvParams {
    const float* x;
    const float* w;
    const uint16_t* w16;  // same layout as w, pre-rounded to bf16 (bf16 kernel only; nullptr -> round w on the fly)
    const float* bias;    // [Cout_p] or nullptr
    const float* res;     // residual (res_mode != 0)
    float* y;
    float* ws;            // split-K partials [splitk][M][Cout_ws]
    int* tile_cnt;        // per-tile arrival counters (zero between launches) -> in-launch split-K reduction; nullptr -> reduce kernel
    const int* m_count;   // optional device int: number of valid items; M_eff = min(M, *m_count * m_per_item)
    int m_per_item;
    int m_hint;           // expected live rows of a count-limited launch (sizes its grid; any count is still handled), 0 = none
    int B, H, W, cin_log2;
    int OH, OW, Cout;
    int KH, KW, stride, pad;
    int KWCp;             // padded run length (multiple of 32)
    int M;                // B*OH*OW
    int relu;
    int res_mode;         // 0 none, 1 same shape, 2 nearest-2x upsample of a [B][OH/2][OW/2][Cout] map
    int out_mode;         // 0 NHWC [M][y_ld] at channel offset y_coff; 1 deconv 2x2 scatter (n = (dy*2+dx)*Cdec + co)
    int y_ld, y_coff;
    int cdec;
    int splitk;           // >= 1
    int steps_total;      // KH * KWCp/32
    const void* next_w;   // filters of the NEXT layer (or nullptr): each block touches a slice so they are L2/MALL-warm
    unsigned next_w_bytes;
    int prec;             // 0: exact f32 MFMA; 1 / 2: bf16 / f16 operands (rounded at LDS staging unless stored so), f32 accumulate
    int x_st, res_st, y_st;   // storage type of x / res / y: 0 f32, 1 bf16, 2 f16 (pointers are then 16-bit element arrays)
    int no_stream;            // 1: keep this launch on the tiled kernel (a caller forcing a tile shape: tests, sweeps)
};

// cfg: 0=128x128 1=64x64 2=128x32 3=128x64.  ev0/ev1 (optional) are recorded right before / after the
// implicit-GEMM kernel itself (a split-K reduce pass, if any, follows ev1).
int apse_launch_conv(const ConvParams& p, int cfg, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
int apse_conv_pick_cfg(int M, int Cout, int steps, int* splitk);
// conv1x1_stream.hip: the memory-streaming kernel for small-K / wide-N 1x1 layers (cfg label APSE_CFG_STREAM in profiles)
#define APSE_CFG_STREAM 9
#define APSE_CFG_BNECK 10            // bottleneck16.hip: a whole 64-channel bottleneck (conv1 + conv2 + conv3 + residual) in one launch (profile label only)
#define APSE_CFG_GLDS 11             // conv_glds16.hip: 256x128 tile, 16-bit operands, LDS-DMA ring
#define APSE_CFG_STEMPOOL 12         // stem_pool16.hip: stem convolution + ReLU + max-pool of the 16-bit modes (profile label only)
#define APSE_CFG_SKINNY 13           // conv_skinny.hip: <= 16 output channels over K = 256, activations straight into 16x16x4 MFMAs (RPN head, mask predictor)
#define APSE_NCFG 14
bool apse_conv1x1_stream_ok(const ConvParams& p);
int apse_launch_conv1x1_stream(const ConvParams& p, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
bool apse_conv_glds16_ok(const ConvParams& p);
bool apse_conv_glds16_small(const ConvParams& p);   // the 128x128 tile of that kernel would be picked (few 256-row tiles)
bool apse_conv_skinny_ok(const ConvParams& p);
int apse_launch_conv_skinny(const ConvParams& p, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
int apse_launch_conv_glds16(const ConvParams& p, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
// which kernel apse_launch_conv runs for (p, cfg): a special kernel's label (APSE_CFG_STREAM, ...), or cfg itself (a tiled shape)
int apse_conv_effective_cfg(const ConvParams& p, int cfg);
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------- typed 4-element access (device)
// Activations are f32 or 16-bit (bf16 / f16) in HBM depending on the mode; arithmetic is always f32.
#ifdef __HIPCC__
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 apse_ld4(const void* base, size_t idx, int st) {
    if (st == 0) return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx);
    f32x4 r;
    if (st == 1) {
        const uint2 raw = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + idx);
        r[0] = __uint_as_float(raw.x << 16); r[1] = __uint_as_float(raw.x & 0xffff0000u);
        r[2] = __uint_as_float(raw.y << 16); r[3] = __uint_as_float(raw.y & 0xffff0000u);
    } else {
        const f16x4_t h = *reinterpret_cast<const f16x4_t*>(reinterpret_cast<const uint16_t*>(base) + idx);
        r[0] = (float)h[0]; r[1] = (float)h[1]; r[2] = (float)h[2]; r[3] = (float)h[3];
    }
    return r;
}
// Layer outputs are written with the nontemporal hint: they are consumed by the NEXT launch (after the end-of-kernel
// write-back), never by this one, and streaming them keeps the filters / input tiles of the running launch in L2
// (res4 conv3: 643 -> 622 us per frame; neutral elsewhere).
#ifndef APSE_NT
#define APSE_NT 1
#endif
#if APSE_NT
#define APSE_NT_STORE(v, ptr) __builtin_nontemporal_store(v, ptr)
#else
#define APSE_NT_STORE(v, ptr) (*(ptr) = (v))
#endif
// ReLU as the select `x > 0 ? x : 0` computes it (-0 and NaN give +0) in ONE instruction: v_max_f32 with the constant first.  Written
// in C the compiler emits a canonicalising `v_max x, x` in front of the max (IEEE mode), i.e. two instructions per element.
__device__ __forceinline__ float apse_relu(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ void apse_st4(void* base, size_t idx, f32x4 v, int st) {
    if (st == 0) { APSE_NT_STORE(v, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + idx)); return; }
    if (st == 1) {
        bf16x4_t b;
        b[0] = (__bf16)v[0]; b[1] = (__bf16)v[1]; b[2] = (__bf16)v[2]; b[3] = (__bf16)v[3];
        APSE_NT_STORE(b, reinterpret_cast<bf16x4_t*>(reinterpret_cast<uint16_t*>(base) + idx));
    } else {
        f16x4_t h;
        h[0] = (_Float16)v[0]; h[1] = (_Float16)v[1]; h[2] = (_Float16)v[2]; h[3] = (_Float16)v[3];
        APSE_NT_STORE(h, reinterpret_cast<f16x4_t*>(reinterpret_cast<uint16_t*>(base) + idx));
    }
}
// Warm the NEXT layer's filters from this launch: the next launch otherwise starts with every block missing on the
// same cold lines (~1-2 us at batch 1, where a layer is only 15-60 us long).  Each block of the first K slice touches
// one slice of at most 16 KB: four independent 16-byte loads per thread, ISSUED in front of the block's first
// prologue fetch and RETIRED (a dummy use) behind it, so they share the prologue's one memory round trip.
// Values are discarded.
struct ApseWarm { f32x4 v[4]; };
__device__ __forceinline__ void apse_warm_issue(ApseWarm& wv, const void* w, unsigned bytes, unsigned blk, unsigned nblk, int tid) {
    const unsigned stride = (bytes / nblk + 1023u) & ~1023u;
    const unsigned len = stride > 16384u ? 16384u : stride;
    const unsigned lo = blk * stride;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned o = (unsigned)(i * 256 + tid) * 16u;
        wv.v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (w && o < len && lo + o + 16u <= bytes) wv.v[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(w) + lo + o);
    }
}
__device__ __forceinline__ void apse_warm_retire(ApseWarm& wv) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(wv.v[i]));
}
// ---- 8 consecutive 16-bit channels (16 bytes) <-> f32
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x8 apse_cvt8(const uint4 raw, int st) {       // 8 consecutive 16-bit channels -> f32
    f32x8 r;
    if (st == 1) {
        r[0] = __uint_as_float(raw.x << 16); r[1] = __uint_as_float(raw.x & 0xffff0000u);
        r[2] = __uint_as_float(raw.y << 16); r[3] = __uint_as_float(raw.y & 0xffff0000u);
        r[4] = __uint_as_float(raw.z << 16); r[5] = __uint_as_float(raw.z & 0xffff0000u);
        r[6] = __uint_as_float(raw.w << 16); r[7] = __uint_as_float(raw.w & 0xffff0000u);
    } else {
        f16x8_t h;
        __builtin_memcpy(&h, &raw, 16);
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = (float)h[k];
    }
    return r;
}

__device__ __forceinline__ void apse_st8(void* base, size_t idx, const f32x8& v, int st) {
    if (st == 0) {
        apse_st4(base, idx, f32x4{v[0], v[1], v[2], v[3]}, 0);
        apse_st4(base, idx + 4, f32x4{v[4], v[5], v[6], v[7]}, 0);
    } else if (st == 1) {
        bf16x8 b;
#pragma unroll
        for (int k = 0; k < 8; ++k) b[k] = (__bf16)v[k];
        APSE_NT_STORE(b, reinterpret_cast<bf16x8*>(reinterpret_cast<uint16_t*>(base) + idx));
    } else {
        f16x8_t h;
#pragma unroll
        for (int k = 0; k < 8; ++k) h[k] = (_Float16)v[k];
        APSE_NT_STORE(h, reinterpret_cast<f16x8_t*>(reinterpret_cast<uint16_t*>(base) + idx));
    }
}

__device__ __forceinline__ void apse_st1(void* base, size_t idx, float v, int st) {
    if (st == 0) reinterpret_cast<float*>(base)[idx] = v;
    else if (st == 1) reinterpret_cast<__bf16*>(base)[idx] = (__bf16)v;
    else reinterpret_cast<_Float16*>(base)[idx] = (_Float16)v;
}
#endif

// ---------------------------------------------------------------- small helpers
static inline int apse_ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline int apse_roundup(int v, int m) { return (v + m - 1) / m * m; }
