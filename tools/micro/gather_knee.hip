// Exploration tool (not part of the library): rate of dependent random 64-byte fetches against the
// size of the table, in fine steps between 256 MB and 8 GB (where gather_sweep.hip shows the rate
// falling by more than half), and what the fall goes with: the bytes fetched from, or the span of
// addresses they lie in (records at a stride; a table in two halves far apart; a table made of
// separate allocations).
// build: hipcc -O3 --offload-arch=gfx950 -o gather_knee gather_knee.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mixu(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ void k_fill(uint4 *t, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        uint32_t b = (uint32_t)i * 4u + (uint32_t)(i >> 30);
        t[i] = make_uint4(mixu(b), mixu(b + 1), mixu(b + 2), mixu(b + 3));
    }
}

// One 64-byte record per step, uniformly random among n_rec records; record r lies at 64-byte
// position r * stride, and the records from `half` on lie `gap` positions further up.
__global__ __launch_bounds__(256) void k_chase(const uint4 *t, uint32_t n_rec, uint32_t stride, uint32_t half,
                                               uint64_t gap, int iters, uint32_t *out) {
    uint32_t h = mixu(blockIdx.x * 256 + threadIdx.x);
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint32_t idx = (uint32_t)(((uint64_t)h * n_rec) >> 32);
        const uint64_t pos = (uint64_t)idx * stride + (idx >= half ? gap : 0);
        const uint4 *p = t + 4 * pos;
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

static double rate(const uint4 *t, uint32_t n_rec, uint32_t stride, uint32_t half, uint64_t gap) {
    const int iters = 200;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        k_chase<<<blocks, 256>>>(t, n_rec, stride, half, gap, iters, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    return (double)blocks * 256 * iters / best * 1e-6;
}

int main() {
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    blocks = cus * 5;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t max_bytes = (size_t)16 << 30;
    uint4 *t; CK(hipMalloc(&t, max_bytes));
    k_fill<<<4096, 256>>>(t, max_bytes / 16);
    CK(hipDeviceSynchronize());
    for (size_t mb : {256, 1024, 2048, 3072, 3584, 3712, 3840, 3968, 4096, 4352, 5120, 6144, 8192, 16384}) {
        const uint32_t n_rec = (uint32_t)((mb << 20) / 64);
        printf("one allocation, first %5zu MB : %.1f G fetches/s\n", mb, rate(t, n_rec, 1, n_rec, 0));
        fflush(stdout);
    }
    for (size_t mb : {1024, 1792, 2048, 3072, 3584}) {
        const uint32_t n_rec = (uint32_t)((mb << 20) / 64);
        printf("%5zu MB of records, every other 64 bytes of %5zu MB : %.1f   every fourth of %5zu MB : %.1f\n", mb, 2 * mb,
               rate(t, n_rec, 2, n_rec, 0), 4 * mb, rate(t, n_rec, 4, n_rec, 0));
        fflush(stdout);
    }
    for (size_t mb : {2048, 3072, 3584}) {
        const uint32_t n_rec = (uint32_t)((mb << 20) / 64);
        const uint64_t gap = (((size_t)12 << 30) - (mb << 19)) / 64;      // second half starts at 12 GB
        printf("%5zu MB in two halves, at 0 and at 12 GB : %.1f\n", mb, rate(t, n_rec, 1, n_rec / 2, gap));
        fflush(stdout);
    }
    CK(hipFree(t));
    // the same 3.5 GB as 7 separate allocations of 512 MB: positions are mapped through a table
    // of bases by the host choosing the gap -- simplest: two allocations, measure their distance
    uint4 *a, *b; CK(hipMalloc(&a, (size_t)1792 << 20)); CK(hipMalloc(&b, (size_t)1792 << 20));
    k_fill<<<4096, 256>>>(a, ((size_t)1792 << 20) / 16);
    k_fill<<<4096, 256>>>(b, ((size_t)1792 << 20) / 16);
    CK(hipDeviceSynchronize());
    const uint4 *lo = a < b ? a : b, *hi = a < b ? b : a;
    const uint64_t dist = (uint64_t)(hi - lo) / 4;                        // in 64-byte positions
    const uint32_t n_half = (uint32_t)(((size_t)1792 << 20) / 64);
    printf("two allocations of 1792 MB, %.2f GB apart : %.1f\n", (double)dist * 64 / (1 << 30),
           rate(lo, 2 * n_half, 1, n_half, dist - n_half));
    return 0;
}
