// Gate for a split-operand f32 mode: f32 GEMM on the bf16 matrix cores, BOTH operands stored as three bf16
// planes (a = a1 + a2 + a3 exactly; an activation tensor would be written that way by the producing layer's epilogue: 6 instead of
// 4 bytes per element) and fed to the matrix cores by LDS-DMA like csrc/conv_glds16.hip -- no VALU in the k loop.
//   hipcc --offload-arch=gfx950 -O3 -o split_gemm2 split_gemm2.hip && ./split_gemm2
// tools/micro/split_gemm.hip (round 3) split A with VALU while staging it through registers: out2 571 us = the production f32
// kernel's 564.  The gate the judge set for going further: out2 (M 64 512, N 256, K 2 304) <= 400 us with six products AND
// res4.c2 (M 4 032) <= 43 us (the production f32 kernel's time at batch 1).
//   C[M][N] = A[M][K] x B[N][K]^T; planes [3][rows][K] bf16.  Six products a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1.
// Block = WM x WN waves of TM x TN MFMA tiles (32 x 32 x 16 bf16), K step 32 (a plane row = 64 B per stage), two LDS stages filled
// by `buffer_load_dwordx4 ... lds` (16 rows x 64 B per wave instruction, XOR swizzle of the 16-byte slots on the SOURCE side),
// one counted wait + one raw barrier per stage.  Per 32-deep stage a wave issues 12 TM TN MFMAs for 6 (TM + TN) fragment reads:
// with 2 x 2 tiles 0.5 ds_read_b128 per MFMA (the plain bf16 kernels: 1.0) -- the six products re-use every fragment.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define BK 32
#define ROWB 64

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(64 * WM * WN) void split_gemm2(const uint16_t* __restrict__ Ap, const uint16_t* __restrict__ Bp, float* __restrict__ C,
                                                            int M, int N, int K) {
    constexpr int NW = WM * WN, BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int APL = BM * ROWB, BPL = BN * ROWB;                 // bytes of one plane of a stage
    constexpr int STAGE = 3 * (APL + BPL);
    constexpr int APIECES = 3 * BM / 16, BPIECES = 3 * BN / 16;     // wave instructions (16 rows x 64 B) per stage
    constexpr int NPIECES = APIECES + BPIECES;
    constexpr int PPW = (NPIECES + NW - 1) / NW;                    // per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN, fr = lane & 31, fh = lane >> 5;
    const int tiles_n = N / BN;
    const int tiles_m = (M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    int bid;
    {   // XCD-aware bijective remap: contiguous tile ranges per XCD
        const int wg = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
    __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Ap), 0, (int)((size_t)3 * M * K * 2), 0x00020000);
    __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Bp), 0, (int)((size_t)3 * N * K * 2), 0x00020000);
    // this lane's share of a stage: piece q = wave + NW * i -> (operand, plane, 16-row group); lane -> (row = lane >> 2, slot = lane & 3)
    unsigned goff[PPW];      // byte offset of the lane's 16 bytes at k = 0
    int lds_off[PPW];
    bool isb[PPW], live[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wave + NW * i;
        live[i] = q < NPIECES;
        isb[i] = q >= APIECES;
        const int qq = isb[i] ? q - APIECES : q;
        const int rows = isb[i] ? BN : BM;
        const int plane = qq / (rows / 16), grp = qq % (rows / 16);
        const int row = grp * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        const int grow = (isb[i] ? n0 : m0) + row;
        const size_t prow = (size_t)(isb[i] ? N : M);
        const bool in = isb[i] || grow < M;
        goff[i] = in ? (unsigned)(((plane * prow + grow) * K + chunk * 8) * 2) : 0xfffffff0u;
        lds_off[i] = (isb[i] ? 3 * APL + plane * BPL : plane * APL) + grp * 1024;
    }
    auto issue = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (!live[i]) continue;
            const unsigned off = goff[i] == 0xfffffff0u ? goff[i] : goff[i] + (unsigned)(k0 * 2);
            if (isb[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (lds_ptr_t)(smem + buf * STAGE + lds_off[i]), 16, (int)off, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lds_ptr_t)(smem + buf * STAGE + lds_off[i]), 16, (int)off, 0, 0, 0);
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    const int T = K / BK;
    issue(0, 0);
    for (int t = 0; t < T; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");            // stage t landed for every wave; all reads of stage t - 1 are done
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < T) issue((t + 1) & 1, (t + 1) * BK);
        const char* st = smem + (t & 1) * STAGE;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            bf16x8 a[3][TM], b[3][TN];
            const int slot = kc * 2 + fh;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = (wm * TM + i) * 32 + fr;
                    a[p][i] = *reinterpret_cast<const bf16x8*>(st + p * APL + row * ROWB + ((slot ^ ((row >> 2) & 3)) << 4));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = (wn * TN + j) * 32 + fr;
                    b[p][j] = *reinterpret_cast<const bf16x8*>(st + 3 * APL + p * BPL + row * ROWB + ((slot ^ ((row >> 2) & 3)) << 4));
                }
            }
            // smallest terms first; six independent chains per product index keep the pipe full
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
                constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[pr]][i], b[PB[pr]][j], acc[i][j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = m0 + (wm * TM + i) * 32 + (v & 3) + 8 * (v >> 2) + 4 * fh;
                const int n = n0 + (wn * TN + j) * 32 + fr;
                if (m < M) C[(size_t)m * N + n] = acc[i][j][v];
            }
}

static void host_split(float v, uint16_t& p1, uint16_t& p2, uint16_t& p3) {
    uint32_t u; memcpy(&u, &v, 4);
    uint32_t b1 = u & 0xffff0000u; float f1; memcpy(&f1, &b1, 4);
    float r1 = v - f1; uint32_t u1; memcpy(&u1, &r1, 4);
    uint32_t b2 = u1 & 0xffff0000u; float f2; memcpy(&f2, &b2, 4);
    float r2 = r1 - f2; uint32_t u2; memcpy(&u2, &r2, 4);
    p1 = (uint16_t)(b1 >> 16); p2 = (uint16_t)(b2 >> 16); p3 = (uint16_t)(u2 >> 16);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int WM, int WN, int TM, int TN>
static int run(const char* name, const uint16_t* dA, const uint16_t* dB, float* dC, int M, int N, int K, const std::vector<float>& hA,
               const std::vector<float>& hB) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    if (N % BN) { printf("%-8s tile %dx%d: N not a multiple\n", name, BM, BN); return 0; }
    const size_t lds = 2 * 3 * (size_t)(BM + BN) * ROWB;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&split_gemm2<WM, WN, TM, TN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = ((M + BM - 1) / BM) * (N / BN);
    CK(hipMemset(dC, 0, (size_t)M * N * 4));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((split_gemm2<WM, WN, TM, TN>), dim3(grid), dim3(64 * WM * WN), lds, 0, dA, dB, dC, M, N, K);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((split_gemm2<WM, WN, TM, TN>), dim3(grid), dim3(64 * WM * WN), lds, 0, dA, dB, dC, M, N, K);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps;
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0, worst_f32 = 0.0;
    for (int t = 0; t < 600; ++t) {
        const int m = t < 8 ? (t < 4 ? t : M - 1 - (t - 4)) : (int)(((uint64_t)t * 2654435761u) % (uint64_t)M);
        const int n = (int)(((uint64_t)t * 40503u + 17) % (uint64_t)N);
        double ref = 0.0, mag = 0.0;
        float f32sum = 0.f;
        for (int k = 0; k < K; ++k) {
            const double p = (double)hA[(size_t)m * K + k] * (double)hB[(size_t)n * K + k];
            ref += p; mag += fabs(p);
            f32sum += hA[(size_t)m * K + k] * hB[(size_t)n * K + k];
        }
        const double e = fabs((double)hC[(size_t)m * N + n] - ref) / (mag + 1e-30);
        const double ef = fabs((double)f32sum - ref) / (mag + 1e-30);
        worst = e > worst ? e : worst;
        worst_f32 = ef > worst_f32 ? ef : worst_f32;
    }
    printf("%-8s M=%6d N=%5d K=%5d  tile %3dx%3d (%d waves, %3zu KB LDS, %4d blocks): %8.1f us  %7.1f TFLOP/s f32-equivalent (%6.0f TFLOP/s of bf16 MFMA work)  "
           "max |err| / sum|a b| = %.2e  (plain f32 loop: %.2e)\n",
           name, M, N, K, BM, BN, WM * WN, lds >> 10, grid, us, 2.0 * M * N * K / us / 1e6, 12.0 * M * N * K / us / 1e6, worst, worst_f32);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 0;
}

int main() {
    struct Shape { const char* name; int M, N, K; };
    const Shape shapes[] = {{"out2", 64512, 256, 2304}, {"out3", 16128, 256, 2304}, {"res4.c2", 4032, 256, 2304}, {"res4.c1", 4032, 256, 1024},
                            {"res4.c3", 4032, 1024, 256}, {"fc1", 1000, 1024, 12544}};
    for (const Shape& sh : shapes) {
        const int M = sh.M, N = sh.N, K = sh.K;
        std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
        uint32_t st = 12345u + (uint32_t)M;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
        for (auto& v : hA) v = rnd() * (1.0f + 0.37f * rnd());
        for (auto& v : hB) v = 0.05f * rnd();
        std::vector<uint16_t> hAp((size_t)3 * M * K), hBp((size_t)3 * N * K);
        for (size_t i = 0; i < (size_t)M * K; ++i) host_split(hA[i], hAp[i], hAp[(size_t)M * K + i], hAp[(size_t)2 * M * K + i]);
        for (size_t i = 0; i < (size_t)N * K; ++i) host_split(hB[i], hBp[i], hBp[(size_t)N * K + i], hBp[(size_t)2 * N * K + i]);
        uint16_t *dA, *dB; float* dC;
        CK(hipMalloc(&dA, hAp.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dB, hBp.size() * 2));
        CK(hipMemcpy(dA, hAp.data(), hAp.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hBp.data(), hBp.size() * 2, hipMemcpyHostToDevice));
        if (run<4, 2, 2, 2>(sh.name, dA, dB, dC, M, N, K, hA, hB)) return 1;      // 256 x 128, 8 waves of 64 x 64
        if (run<2, 2, 2, 2>(sh.name, dA, dB, dC, M, N, K, hA, hB)) return 1;      // 128 x 128, 4 waves of 64 x 64
        if (run<4, 2, 1, 1>(sh.name, dA, dB, dC, M, N, K, hA, hB)) return 1;      // 128 x 64, 8 waves of 32 x 32
        if (run<2, 2, 1, 1>(sh.name, dA, dB, dC, M, N, K, hA, hB)) return 1;      // 64 x 64, 4 waves of 32 x 32
        if (run<2, 4, 2, 1>(sh.name, dA, dB, dC, M, N, K, hA, hB)) return 1;      // 128 x 128, 8 waves of 64 x 32
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    return 0;
}
