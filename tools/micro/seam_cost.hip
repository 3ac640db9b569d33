// What does a layer boundary cost INSIDE one persistent launch, against a kernel boundary, for a res4-shaped hand-off?
//   hipcc --offload-arch=gfx950 -O3 -o seam_cost seam_cost.hip && ./seam_cost
// VERDICT r2 #3 asked for the res4 chain (69 dependent layers, one tile wave each: 252 tiles of 64x64 on 256 CUs) as one
// persistent launch with a device-scope barrier per layer.  Every tile of layer L+1 needs the output of SEVERAL tiles of
// layer L (all channel tiles of its rows, plus the halo rows of a 3x3), so the seam is grid-wide.  This program measures that
// seam with the data movement of such a layer and nothing else:
//   a "phase": 256 blocks x 512 threads; block b reads 64 KB written in the previous phase by four OTHER blocks (16-byte
//   loads) and writes 16 KB of its own (16-byte stores) -- the sizes of a 64x64 f32 output tile and of the activation rows a
//   tile's prologue pulls.
//   (A) N phases as N dependent launches on one stream;
//   (B) N phases inside ONE launch, separated by an XCD-sharded arrive counter + generation word (release fence before the
//       arrive, acquire fence after the wait: MI355X_MICROARCH.md "barrier-xcd"), every spin bounded with a give-up flag;
//   (C) the same barrier with no data movement at all (the bare seam).
// Printed: microseconds per phase.  (B) - (A) is what a seam costs more (or less) than the kernel boundary it would replace.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define NB 256
#define NT 512

struct Sync { int xcnt[8 * 32]; int top; int pad0[31]; int gen; int pad1[31]; int err; };

__device__ __forceinline__ void phase_body(const float4* __restrict__ x, float4* __restrict__ y, int b, int tid) {
    float4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int src = (b + 1 + 61 * k) & (NB - 1);             // four other blocks' slices
        const float4* p = x + (size_t)src * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 v = p[i * NT + tid];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    float4* q = y + (size_t)b * 1024;
    q[tid] = acc;
    acc.x += 1.f;
    q[NT + tid] = acc;
}

__global__ __launch_bounds__(NT) void phase_k(const float4* __restrict__ x, float4* __restrict__ y) { phase_body(x, y, blockIdx.x, threadIdx.x); }

// returns false when the barrier gave up (some block never arrived): the caller leaves the kernel
__device__ __forceinline__ bool grid_seam(Sync* s, int phase, int tid) {
    __shared__ int ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int shard = blockIdx.x & 7;
        const int t = __hip_atomic_fetch_add(&s->xcnt[shard * 32], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (NB / 8) * (phase + 1) - 1) {
            const int tt = __hip_atomic_fetch_add(&s->top, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tt == 8 * (phase + 1) - 1) __hip_atomic_store(&s->gen, phase + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int good = 1;
        for (int spins = 0;; ++spins) {
            if (__hip_atomic_load(&s->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= phase + 1) break;
            if (spins > (1 << 22) || __hip_atomic_load(&s->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&s->err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

__global__ __launch_bounds__(NT) void chain_k(float4* a, float4* b, Sync* s, int nphase, int move) {
    const int tid = threadIdx.x;
    for (int ph = 0; ph < nphase; ++ph) {
        if (move) phase_body((ph & 1) ? b : a, (ph & 1) ? a : b, blockIdx.x, tid);
        if (!grid_seam(s, ph, tid)) return;
    }
}

int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    float4 *a, *b;
    Sync* sy;
    hipMalloc(&a, NB * 1024 * sizeof(float4));
    hipMalloc(&b, NB * 1024 * sizeof(float4));
    hipMalloc(&sy, sizeof(Sync));
    hipMemset(a, 0, NB * 1024 * sizeof(float4));
    hipMemset(b, 0, NB * 1024 * sizeof(float4));
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, chain_k, NT, 0);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("CUs %d, chain_k blocks per CU (occupancy API) %d, grid %d\n", prop.multiProcessorCount, occ, NB);
    if (occ * prop.multiProcessorCount < NB) { printf("grid not co-resident: not running the persistent form\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int N = 1000;
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, st);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(phase_k, dim3(NB), dim3(NT), 0, st, (i & 1) ? b : a, (i & 1) ? a : b);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double t_launch = 1e3 * ms / N;
    printf("(A) %d dependent launches, 64 KB read + 16 KB written per block: %.2f us per phase\n", N, t_launch);
    double t_in[2] = {0, 0};
    for (int move = 1; move >= 0; --move) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemsetAsync(sy, 0, sizeof(Sync), st);
            hipEventRecord(e0, st);
            hipLaunchKernelGGL(chain_k, dim3(NB), dim3(NT), 0, st, a, b, sy, N, move);
            hipEventRecord(e1, st);
            if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed\n"); return 2; }
            hipEventElapsedTime(&ms, e0, e1);
        }
        Sync h;
        hipMemcpy(&h, sy, sizeof(Sync), hipMemcpyDeviceToHost);
        if (h.err) { printf("barrier gave up (err flag set): result void\n"); return 3; }
        t_in[move] = 1e3 * ms / N;
        printf("(%s) one launch, %d phases, sharded arrive counter + generation word%s: %.2f us per phase\n", move ? "B" : "C", N,
               move ? ", same data movement" : ", no data movement (bare seam)", t_in[move]);
    }
    printf("seam inside the launch minus kernel boundary: %+.2f us per layer\n", t_in[1] - t_launch);
    return 0;
}
