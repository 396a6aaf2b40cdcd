// Exploration tool (not part of the library): dependent random 64-byte fetches from a table laid
// out as several equal regions, in one allocation or in several -- what an allocation's size and
// the number of allocations do to the fetch rate beyond the caches.
// build: hipcc -O3 --offload-arch=gfx950 -o gather_regions gather_regions.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mixu(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ void k_fill(uint4 *t, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        uint32_t b = (uint32_t)i * 4u + (uint32_t)(i >> 30);
        t[i] = make_uint4(mixu(b), mixu(b + 1), mixu(b + 2), mixu(b + 3));
    }
}

struct Regions { const uint4 *base[16]; };

__global__ __launch_bounds__(256) void k_chase(Regions r, uint32_t n_regions, uint32_t len, int iters, uint32_t *out) {
    uint32_t h = mixu(blockIdx.x * 256 + threadIdx.x);
    uint32_t acc = 0;
    const uint32_t n_rec = n_regions * len;
    for (int i = 0; i < iters; ++i) {
        const uint32_t idx = (uint32_t)(((uint64_t)h * n_rec) >> 32);
        const uint32_t reg = idx / len, off = idx - reg * len;
        const uint4 *p = r.base[reg] + 4 * (size_t)off;
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { uint4 q = p[k]; v += q.x + q.w; }
        acc += v;
        h = mixu(v + i);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static hipEvent_t e0, e1;
static uint32_t *out;
static int blocks;

static double rate(const Regions &r, uint32_t n_regions, size_t region_mb) {
    const int iters = 200;
    const uint32_t len = (uint32_t)((region_mb << 20) / 64);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        k_chase<<<blocks, 256>>>(r, n_regions, len, iters, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    return (double)blocks * 256 * iters / best * 1e-6;
}

static uint4 *alloc_mb(size_t mb) {
    uint4 *p; CK(hipMalloc(&p, mb << 20));
    k_fill<<<4096, 256>>>(p, (mb << 20) / 16);
    CK(hipDeviceSynchronize());
    return p;
}

// n allocations of alloc_mb each, region_mb of each used
static void several(int n, size_t a_mb, size_t region_mb) {
    std::vector<uint4 *> ps;
    Regions r{};
    for (int i = 0; i < n; ++i) { ps.push_back(alloc_mb(a_mb)); r.base[i] = ps.back(); }
    printf("%2d allocations of %5zu MB, %5zu MB of each used (%5zu MB in all) : %.1f G fetches/s   [first at %p]\n", n, a_mb, region_mb,
           n * region_mb, rate(r, n, region_mb), (void *)ps[0]);
    fflush(stdout);
    for (uint4 *p : ps) CK(hipFree(p));
}

int main() {
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    blocks = cus * 5;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    {
        uint4 *t = alloc_mb(16384);
        Regions r{};
        for (int i = 0; i < 8; ++i) r.base[i] = t + ((size_t)i * 2048 << 20) / 16;
        printf("one 16 GB allocation, 8 regions of 448 MB at 2 GB intervals (3584 MB) : %.1f\n", rate(r, 8, 448));
        printf("one 16 GB allocation, 8 regions of 512 MB at 2 GB intervals (4096 MB) : %.1f\n", rate(r, 8, 512));
        for (int i = 0; i < 16; ++i) r.base[i] = t + ((size_t)i * 1024 << 20) / 16;
        printf("one 16 GB allocation, 16 regions of 224 MB at 1 GB intervals (3584 MB) : %.1f\n", rate(r, 16, 224));
        printf("one 16 GB allocation, 16 regions of 128 MB at 1 GB intervals (2048 MB) : %.1f\n", rate(r, 16, 128));
        fflush(stdout);
        CK(hipFree(t));
    }
    several(1, 3584, 3584);
    several(1, 4096, 3584);
    several(1, 1792, 1792);
    several(2, 1792, 1792);
    several(2, 2048, 1792);
    several(2, 2048, 1024);
    several(2, 1024, 1024);
    several(3, 1024, 1024);
    several(4, 1024, 896);
    several(4, 896, 896);
    several(7, 512, 512);
    several(14, 256, 256);
    several(8, 256, 256);
    several(16, 128, 128);
    several(16, 64, 64);
    return 0;
}
