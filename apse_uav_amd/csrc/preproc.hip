// Optional frame pre-processing (gfx950): undistort + Lab-L gamma, one kernel, u8 BGR in -> u8 BGR out.
// Replaces preprocess_img of /root/reference/dcnn/scripts/tests/visualize_uav.py:56-71 (cv2.undistort,
// cvtColor RGB2LAB, LUT on L, cvtColor LAB2RGB; same maths in aruco_detect.py:250-259).
// The per-pixel arithmetic lives in preproc_pixel.h; inside a context the same function feeds the horizontal resize pass
// directly (apse_set_camera -> pil_resize_h<true>), so this stand-alone form (24.9 MB read + 24.9 MB written per 4K frame)
// is the stateless operator of the tests / FramePreprocessor only.
#include "apse_common.h"
#include "preproc_pixel.h"

__global__ __launch_bounds__(256) void undistort_gamma(const UndistortParams p, const uint8_t* __restrict__ src,
                                                       uint8_t* __restrict__ dst, const LabTables* __restrict__ lab) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int b = blockIdx.z;
    if (x >= p.W) return;
    const uint8_t* s = src + (size_t)b * p.H * p.W * 3;
    int c0, c1, c2;
    undistort_gamma_pixel(p, s, lab, x, y, c0, c1, c2);
    uint8_t* o = dst + (((size_t)b * p.H + y) * p.W + x) * 3;
    o[0] = (uint8_t)c0; o[1] = (uint8_t)c1; o[2] = (uint8_t)c2;
}

// Built once per camera (apse_set_camera): the remap table of the fused path.
// The compact form (4 bytes per pixel, preproc_pixel.h); *overflow counts pixels inside the frame whose displacement does not fit.
__global__ __launch_bounds__(256) void undistort_build_map_compact(const UndistortParams p, uint32_t* __restrict__ map, int* __restrict__ overflow) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= p.W) return;
    map[(size_t)y * p.W + x] = undistort_map_compact(p, x, y, overflow);
}
extern "C" int apse_k_undistort_build_map_compact(const UndistortParams* p, void* map, int* overflow_dev, hipStream_t s) {
    hipLaunchKernelGGL(undistort_build_map_compact, dim3((p->W + 255) / 256, p->H), dim3(256), 0, s, *p, reinterpret_cast<uint32_t*>(map), overflow_dev);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}

extern "C" int apse_k_undistort_gamma(const UndistortParams* p, const uint8_t* src, uint8_t* dst, const LabTables* lab, int B,
                                      hipStream_t s) {
    hipLaunchKernelGGL(undistort_gamma, dim3((p->W + 255) / 256, p->H, B), dim3(256), 0, s, *p, src, dst, lab);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
