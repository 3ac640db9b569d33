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
    int prec;             // 0: exact f32 MFMA; 1: operands rounded to bf16 at LDS staging, f32 accumulate (f32 storage)
};

// cfg: 0=128x128 1=64x64 2=128x32 3=128x64.  ev0/ev1 (optional) are recorded right before / after the
// implicit-GEMM kernel itself (a split-K reduce pass, if any, follows ev1).
int apse_launch_conv(const ConvParams& p, int cfg, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
int apse_conv_pick_cfg(int M, int Cout, int steps, int* splitk);
int apse_launch_conv_bf16(const ConvParams& p, int cfg, hipStream_t s);   // conv_igemm_bf16.hip
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------- small helpers
static inline int apse_ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline int apse_roundup(int v, int m) { return (v + m - 1) / m * m; }
