// What does the SHAPE of a streaming access cost on HBM3E?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 -o seg_stream seg_stream.hip && ./seg_stream
// The 1x1 expansion layers write (and read, as the residual) NHWC maps of N 16-bit channels in chunks of 128 columns: one wave
// instruction covers 8 rows x 128 B (a 64-column half) -- 128-byte segments at a row stride of 2 N bytes (2 KB for res4's
// N = 1024), and a row's 2 KB are completed by 16 instructions that are microseconds apart.  A map written in row order by whole
// 1 KB wave instructions is the other extreme.  Same bytes, same grid (one 256-thread block per 128 rows, as conv1x1_stream):
//   rows   : wave instruction = 1 KB contiguous, a block writes its 128 x 2N bytes front to back
//   seg128 : wave instruction = 8 rows x 128 B, chunk by chunk (the kernel's order)
//   seg256 : wave instruction = 4 rows x 256 B, chunk by chunk
// for stores (nt) and for loads.  M = 33 600 rows (fp16 batch 8, res4), N = 256 / 512 / 1024 / 2048.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool STORE>
__global__ __launch_bounds__(256) void seg_k(char* buf, int M, int N, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t rowb = (size_t)N * 2;
    u32x4 acc = {0, 0, 0, 0};
    const u32x4 val = {1u, 2u, 3u, (unsigned)threadIdx.x};
    auto touch = [&](size_t off) {
        if (off + 16 > (size_t)M * rowb) return;
        if (STORE) __builtin_nontemporal_store(val, reinterpret_cast<u32x4*>(buf + off));
        else { const u32x4 v = *reinterpret_cast<const u32x4*>(buf + off); acc ^= v; }
    };
    const int m0 = blockIdx.x * 128 + wave * 32;
    if (MODE == 0) {
        // the wave's 32 rows front to back, 1 KB per instruction
        const size_t base = (size_t)m0 * rowb;
#pragma unroll 8
        for (size_t o = 0; o < 32 * rowb; o += 1024) touch(base + o + lane * 16);
    } else if (MODE == 1) {
        for (int ch = 0; ch < N / 128; ++ch)
            for (int hh = 0; hh < 2; ++hh)
                for (int i = 0; i < 4; ++i) touch((size_t)(m0 + (lane >> 3) + 8 * i) * rowb + ch * 256 + hh * 128 + (lane & 7) * 16);
    } else {
        for (int ch = 0; ch < N / 128; ++ch)
            for (int i = 0; i < 8; ++i) touch((size_t)(m0 + (lane >> 4) + 4 * i) * rowb + ch * 256 + (lane & 15) * 16);
    }
    if (!STORE && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u) sink[0] = 1;
}

typedef void (*kern_t)(char*, int, int, unsigned*);

int main() {
    hipStream_t s;
    hipStreamCreate(&s);
    const int M = 33600;
    const size_t ring = 8, maxb = (size_t)M * 2048 * 2;
    char* buf;
    unsigned* sink;
    if (hipMalloc(&buf, ring * maxb) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, ring * maxb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct K { const char* name; kern_t f; };
    K ks[] = {{"store rows", seg_k<0, true>},   {"store seg128", seg_k<1, true>}, {"store seg256", seg_k<2, true>},
              {"load  rows", seg_k<0, false>},  {"load  seg128", seg_k<1, false>}, {"load  seg256", seg_k<2, false>}};
    printf("%-8s %-14s %10s %10s\n", "N", "access", "us", "TB/s");
    for (int N : {256, 512, 1024, 2048})
        for (auto& k : ks) {
            const size_t bytes = (size_t)M * N * 2;
            const size_t nring = (ring * maxb) / bytes;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0, s);
                for (int i = 0; i < 16; ++i)      // a ring of maps: every launch touches memory that is in no cache
                    hipLaunchKernelGGL(k.f, dim3((M + 127) / 128), dim3(256), 0, s, buf + (size_t)((i * 5) % nring) * bytes, M, N, sink);
                hipEventRecord(e1, s);
                hipEventSynchronize(e1);
                float ms = 0.f;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms / 16 < best) best = ms / 16;
            }
            printf("%-8d %-14s %10.1f %10.2f\n", N, k.name, best * 1e3, bytes / (best * 1e-3) / 1e12);
            fflush(stdout);
        }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
