// How fast does ONE CU take operand bytes in, as a function of the bytes it keeps in flight?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 -o fill_depth fill_depth.hip && ./fill_depth
// The 16-bit trunk kernels (csrc/conv_glds16.hip) feed 8 waves per CU from a ring of LDS stages filled by LDS-DMA; their stage
// time does not follow the bytes of a stage.  This program isolates the feed: one 512-thread block per CU (144 KB of LDS, so
// exactly one), every block sweeps its own region of a buffer in 1 KB wave instructions, nothing is computed.
//   mode "lds":  LDS-DMA (`buffer_load_dwordx4 ... lds`) into a ring of 16 KB stages, `depth` stages in flight, one counted
//                s_waitcnt + one s_barrier per stage -- the conv_glds16 loop without its MFMAs;
//   mode "reg":  each wave keeps R `buffer_load_dwordx4` (1 KB each) in flight into registers (8 waves x R KB per CU).
//   mode "mix":  a 48 KB stage on a 3-slot ring, each wave filling PD pieces by LDS-DMA and PR pieces through registers + ds_write.
//                Its "L2" rows re-read ONE 48 KB stage per CU (the region holds only one) and are flattered by the CU's own L1; the
//                Infinity-Cache and HBM rows are comparable with the other modes.
// Region per block: 64 KB (stays in the XCD's L2), 256 KB (64 MB in all: beyond L2, inside the Infinity Cache), 8 MB (2 GB: HBM).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define STAGE_BYTES 16384     // 8 waves x 2 pieces x 1 KB
#define P 2

// ROWB > 0: every wave instruction gathers 8 rows x 128 B at a row stride of ROWB bytes (the A operand of a 64-deep k step read
// from an NHWC map of ROWB / 2 channels) instead of one contiguous 1 KB run; the region is swept exactly once per pass either way.
template <int DEPTH, int ROWB = 0>
__global__ __launch_bounds__(512) void fill_lds(const char* buf, unsigned region, int passes, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NS = DEPTH + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = buf + (size_t)blockIdx.x * region;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)region, 0x00020000);
    const int per_pass = region / STAGE_BYTES, T = per_pass * passes;
    auto issue = [&](int t) {
        const unsigned off = (unsigned)(t % per_pass) * STAGE_BYTES + wave * (P * 1024) + lane * 16;
        char* dst = smem + (t % NS) * STAGE_BYTES + wave * (P * 1024);
#pragma unroll
        for (int j = 0; j < P; ++j) {
            unsigned o = off + j * 1024;
            if (ROWB > 0) {
                const unsigned piece = ((unsigned)(t % per_pass) * STAGE_BYTES + wave * (P * 1024) + j * 1024) >> 10;   // 1 KB piece index in the region
                constexpr unsigned SL = ROWB / 128;                                                                   // k slices per row
                o = (piece / SL) * 8u * ROWB + (piece % SL) * 128u + (unsigned)(lane >> 3) * ROWB + (unsigned)(lane & 7) * 16u;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + j * 1024), 16, (int)o, 0, 0, 0);
        }
    };
    for (int s = 0; s < DEPTH && s < T; ++s) issue(s);
    unsigned acc = 0;
    for (int t = 0; t < T; ++t) {
        if (t + DEPTH - 1 < T) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * P) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (t + DEPTH < T) issue(t + DEPTH);
        acc ^= *reinterpret_cast<const unsigned*>(smem + (t % NS) * STAGE_BYTES + threadIdx.x * 4);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int R>
__global__ __launch_bounds__(512) void fill_reg(const char* buf, unsigned region, int passes, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = buf + (size_t)blockIdx.x * region;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)region, 0x00020000);
    // wave w takes the 1 KB pieces w, w + 8, ... of the region
    const int pieces = region / 8192, T = pieces * passes;      // per wave
    f32x4 v[R];
    auto ld = [&](int t) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((unsigned)(t % pieces) * 8 + wave) * 1024 + lane * 16), 0, 0)); };
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = ld(i);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = R; t < T + R; t += R) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            acc += v[i];
            v[i] = ld(t + i);          // past the end: wraps, a few extra loads
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) sink[0] = 1;
    if (threadIdx.x == 1023) smem[0] = 0;
}

// mode "mix": a stage = 8 waves x (PD LDS-DMA pieces + PR register-load pieces), 1 KB each; the register pieces are requested one
// stage ahead and written to LDS with ds_write_b128 when their stage comes up -- conv_glds16 with one operand on each path.
template <int PD, int PR_>
__global__ __launch_bounds__(512) void fill_mix(const char* buf, unsigned region, int passes, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PT = PD + PR_, SB = 8 * PT * 1024, NS = 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = buf + (size_t)blockIdx.x * region;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)region, 0x00020000);
    const int per_pass = region / SB, T = per_pass * passes;
    f32x4 r[PR_ > 0 ? PR_ : 1];
    auto issue_dma = [&](int t) {
        const unsigned off = (unsigned)(t % per_pass) * SB + wave * (PT * 1024) + lane * 16;
        char* dst = smem + (t % NS) * SB + wave * (PT * 1024);
#pragma unroll
        for (int j = 0; j < PD; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + j * 1024), 16, (int)(off + j * 1024), 0, 0, 0);
    };
    auto issue_reg = [&](int t) {
        const unsigned off = (unsigned)(t % per_pass) * SB + wave * (PT * 1024) + PD * 1024 + lane * 16;
#pragma unroll
        for (int j = 0; j < PR_; ++j) r[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(off + j * 1024), 0, 0));
    };
    issue_dma(0);
    issue_reg(0);
    if (T > 1) issue_dma(1);
    unsigned acc = 0;
    for (int t = 0; t < T; ++t) {
        // registers of stage t -> LDS (the compiler waits for them: everything older, i.e. the DMAs of stages <= t, is then in too)
        char* dst = smem + (t % NS) * SB + wave * (PT * 1024) + PD * 1024 + lane * 16;
#pragma unroll
        for (int j = 0; j < PR_; ++j) *reinterpret_cast<f32x4*>(dst + j * 1024) = r[j];
        if (PR_ == 0) {
            if (t + 1 < T) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (t + 1 < T) issue_reg(t + 1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // not __syncthreads(): its fence would drain the loads in flight
        if (t + 2 < T) issue_dma(t + 2);
        acc ^= *reinterpret_cast<const unsigned*>(smem + (t % NS) * SB + threadIdx.x * 4);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

typedef void (*kern_t)(const char*, unsigned, int, unsigned*);

int main() {
    hipStream_t s;
    hipStreamCreate(&s);
    const size_t total = (size_t)2 << 30;
    char* buf;
    unsigned* sink;
    if (hipMalloc(&buf, total) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, total);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int lds = 144 * 1024;
    struct K { const char* name; kern_t f; int inflight_kb; };
    K ks[] = {
        {"lds depth 1", fill_lds<1>, 16}, {"lds depth 2", fill_lds<2>, 32}, {"lds depth 3", fill_lds<3>, 48}, {"lds depth 4", fill_lds<4>, 64},
        {"lds depth 6", fill_lds<6>, 96}, {"lds depth 8", fill_lds<8>, 128},
        {"lds d3 rows 512B", fill_lds<3, 512>, 48}, {"lds d3 rows 2KB", fill_lds<3, 2048>, 48}, {"lds d3 rows 4KB", fill_lds<3, 4096>, 48},
        {"reg R 2", fill_reg<2>, 16}, {"reg R 4", fill_reg<4>, 32}, {"reg R 8", fill_reg<8>, 64}, {"reg R 12", fill_reg<12>, 96},
        {"reg R 16", fill_reg<16>, 128}, {"reg R 24", fill_reg<24>, 192}, {"reg R 32", fill_reg<32>, 256},
        {"mix 6 dma+0", fill_mix<6, 0>, 96}, {"mix 4 dma+2 reg", fill_mix<4, 2>, 80}, {"mix 2 dma+4 reg", fill_mix<2, 4>, 64}, {"mix 0+6 reg", fill_mix<0, 6>, 48},
    };
    for (auto& k : ks) hipFuncSetAttribute(reinterpret_cast<const void*>(k.f), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    struct L { const char* name; unsigned region; int passes; };
    L ls[] = {{"L2 (64 KB per CU)", 64u << 10, 256}, {"Infinity Cache (256 KB per CU, 64 MB)", 256u << 10, 64}, {"HBM (8 MB per CU, 2 GB)", 8u << 20, 2}};
    printf("%-40s %-16s %12s %14s %12s\n", "served from", "mode", "in flight KB", "GB/s per CU", "TB/s chip");
    for (auto& l : ls)
        for (auto& k : ks) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0, s);
                hipLaunchKernelGGL(k.f, dim3(256), dim3(512), lds, s, buf, l.region, l.passes, sink);
                hipEventRecord(e1, s);
                hipEventSynchronize(e1);
                float ms = 0.f;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            const double bytes = (double)l.region * l.passes;
            printf("%-40s %-16s %12d %14.1f %12.2f\n", l.name, k.name, k.inflight_kb, bytes / (best * 1e-3) / 1e9, bytes * 256 / (best * 1e-3) / 1e12);
            fflush(stdout);
        }
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
    return 0;
}
