// Exploration tool (not part of the library): rate of dependent random 64-byte fetches against the
// size of the table, in fine steps between 256 MB and 8 GB (where gather_sweep.hip shows the rate
// falling by more than half), from one allocation and from a table cut into separate allocations.
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

// one 64-byte record per step, uniformly random among n_rec records
__global__ __launch_bounds__(256) void k_chase(const uint4 *t, uint32_t n_rec, int iters, uint32_t *out) {
    uint32_t h = mixu(blockIdx.x * 256 + threadIdx.x);
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint32_t idx = (uint32_t)(((uint64_t)h * n_rec) >> 32);
        const uint4 *p = t + 4 * (size_t)idx;
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { uint4 q = p[k]; v += q.x + q.w; }
        acc += v;
        h = mixu(v + i);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    uint32_t *out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    const size_t max_bytes = (size_t)8 << 30;
    uint4 *t; CK(hipMalloc(&t, max_bytes));
    k_fill<<<4096, 256>>>(t, max_bytes / 16);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 200, blocks = cus * 5;
    for (size_t mb : {256, 512, 768, 1024, 1280, 1536, 1792, 2048, 2304, 2560, 3072, 3584, 4096, 5120, 6144, 8192}) {
        const uint32_t n_rec = (uint32_t)((mb << 20) / 64);
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            k_chase<<<blocks, 256>>>(t, n_rec, iters, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("table %5zu MB : %.1f G fetches/s\n", mb, (double)blocks * 256 * iters / best * 1e-6);
        fflush(stdout);
    }
    return 0;
}
