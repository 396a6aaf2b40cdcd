// Camera-ray generation for a whole image and the reference's debug colour.
#include "kernels.h"
#include "pt_camera_math.h"

namespace mipt {

namespace {

__global__ __launch_bounds__(256) void k_raygen(pt_camera cam, uint32_t width, uint32_t height,
                                                pt_ray *rays) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (uint64_t)width * height) return;
    const uint32_t px = (uint32_t)(i % width), py = (uint32_t)(i / width);
    float4 o, d;
    make_camera_ray(cam, width, height, (float)px + 0.5f, (float)py + 0.5f, o, d);
    float4 *p = reinterpret_cast<float4 *>(rays + i);
    p[0] = o;
    p[1] = d;
}

// shaders/depth.frag:90-107
__global__ __launch_bounds__(256) void k_depth_shade(const pt_hit *hits, uint64_t n, float *rgba) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    reinterpret_cast<float4 *>(rgba)[i] = depth_grey(hits[i].shapeid, hits[i].uvwt[3]);
}

} // namespace

void launch_raygen(hipStream_t st, const pt_camera &cam, uint32_t width, uint32_t height,
                   pt_ray *rays) {
    const uint64_t n = (uint64_t)width * height;
    if (n == 0) return;
    hipLaunchKernelGGL(k_raygen, dim3((unsigned int)((n + 255) / 256)), dim3(256), 0, st, cam, width,
                       height, rays);
}

void launch_depth_shade(hipStream_t st, const pt_hit *hits, uint64_t n, float *rgba) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_depth_shade, dim3((unsigned int)((n + 255) / 256)), dim3(256), 0, st, hits,
                       n, rgba);
}

} // namespace mipt
