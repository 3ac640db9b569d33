// Practical MFMA ceiling on this card: a register-only loop of independent MFMAs, no memory traffic.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak
// Prints sustained TFLOP/s for v_mfma_f32_32x32x2_f32 (exact f32) and v_mfma_f32_32x32x16_bf16 so that the
// roofline fractions in bench.py (priced against the 157.3 / 2516 TFLOP/s data-sheet peaks) can be read next to
// what the silicon sustains under its power/clock limits.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// RANDOM = 1: every lane holds 8 different pseudo-random operand values that rotate through the loop, so the
// operand buses toggle like real data (constant operands let the chip hold its top clock: DVFS, see
// MI355X_MICROARCH.md "DVFS give-back").
template <int MODE, int RANDOM>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(a + i); b8[i] = (__bf16)(b - i); }
    float ar[8], br[8];
    bf16x8 a8r[8], b8r[8];
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (int u = 0; u < 8; ++u) {
        h = h * 1664525u + 1013904223u; ar[u] = RANDOM ? ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23)) : a;
        h = h * 1664525u + 1013904223u; br[u] = RANDOM ? ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23)) : b;
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u; a8r[u][i] = RANDOM ? (__bf16)(((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23))) : a8[i];
            h = h * 1664525u + 1013904223u; b8r[u][i] = RANDOM ? (__bf16)(((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23))) : b8[i];
        }
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (MODE == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[(u + t) & 7], br[(u + 2 * t) & 7], acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8r[(u + t) & 7], b8r[(u + 2 * t) & 7], acc[t], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    if (s == 123.456f) out[0] = s;   // keep the loop alive
}

template <int MODE, int RANDOM>
static void run(const char* name, double flop_per_mfma, int blocks_per_cu, int ms_target) {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    float* d;
    hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int iters = 2000;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        mfma_loop<MODE, RANDOM><<<cus * blocks_per_cu, 256, 0, 0>>>(d, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)cus * blocks_per_cu * 4 /*waves*/ * iters * 32.0 * flop_per_mfma;
        printf("%-34s blocks/CU=%d iters=%d  %8.3f ms  %8.1f TFLOP/s\n", name, blocks_per_cu, iters, ms, fl / ms / 1e9);
        if (ms < ms_target) iters *= 4;
    }
    hipFree(d);
}

int main() {
    run<0, 0>("f32 32x32x2  constant operands", 2.0 * 32 * 32 * 2, 2, 200);
    run<0, 1>("f32 32x32x2  random operands", 2.0 * 32 * 32 * 2, 2, 200);
    run<1, 0>("bf16 32x32x16 constant operands", 2.0 * 32 * 32 * 16, 2, 200);
    run<1, 1>("bf16 32x32x16 random operands", 2.0 * 32 * 32 * 16, 2, 200);
    return 0;
}
