// Cost of a kernel boundary on this GPU: N dependent launches on one stream of kernels that do (almost) nothing.
//   hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip && ./launch_floor
// empty<<<1, 64>>>: pure launch + completion; touch<<<256, 256>>>: one block per CU, each thread reads and writes 16 bytes (a
// kernel's minimal memory round trip: its end-of-kernel release makes the writes visible to the next launch).
// The per-launch time of such a chain is what a frame of ~175 dependent launches pays before any useful work.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void empty_k() {}
__global__ __launch_bounds__(256) void touch_k(const float4* __restrict__ x, float4* __restrict__ y) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float4 v = x[i];
    v.x += 1.f;
    y[i] = v;
}

int main() {
    hipStream_t s;
    hipStreamCreate(&s);
    float4 *a, *b;
    hipMalloc(&a, 256 * 256 * sizeof(float4));
    hipMalloc(&b, 256 * 256 * sizeof(float4));
    hipMemset(a, 0, 256 * 256 * sizeof(float4));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int N = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, s);
            for (int i = 0; i < N; ++i) {
                if (mode == 0) hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, s);
                else if (mode == 1) hipLaunchKernelGGL(touch_k, dim3(256), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b);
                else hipLaunchKernelGGL(touch_k, dim3(64), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b);
            }
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2)
                printf("%s: %.2f us per dependent launch (%d launches)\n",
                       mode == 0 ? "empty<<<1,64>>>" : (mode == 1 ? "touch<<<256,256>>> (read 16 B, write 16 B per thread)" : "touch<<<64,256>>>"),
                       1e3 * ms / N, N);
        }
    }
    return 0;
}
