// Exploration tool (not part of the library): what bounds dependent random record fetches beyond
// the caches on this GPU -- table size, resident waves, line neighbours, page locality.
// build: hipcc -O3 --offload-arch=gfx950 -o gather_sweep gather_sweep.hip
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

// MODE 0: one 64 B record per step, uniformly random.
// MODE 1: one 128 B record per step (8 loads), uniformly random 128 B lines.
// MODE 2: 64 B record, then its neighbour in the same 128 B line as the next dependent step.
// MODE 3: 64 B record; `local` consecutive steps stay inside one 2 MB window, then jump.
// MODE 4: 64 B record; `local` consecutive steps stay inside one 4 KB window, then jump.
template <int MODE>
__global__ __launch_bounds__(256) void k_chase(const uint4 *t, uint32_t mask, int iters, int local, uint32_t *out) {
    uint32_t idx = mixu(blockIdx.x * 256 + threadIdx.x) & mask;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t v = 0;
        if (MODE == 1) {
            const uint4 *p = t + 8 * (size_t)(idx >> 1);
#pragma unroll
            for (int k = 0; k < 8; ++k) { uint4 q = p[k]; v += q.x + q.w; }
        } else {
            const uint4 *p = t + 4 * (size_t)idx;
#pragma unroll
            for (int k = 0; k < 4; ++k) { uint4 q = p[k]; v += q.x + q.w; }
        }
        acc += v;
        uint32_t nxt = mixu(v + i) & mask;
        if (MODE == 2 && !(i & 1)) nxt = idx ^ 1u;
        if (MODE == 3 && (i % local)) nxt = (idx & ~0x7FFFu) | (nxt & 0x7FFFu);   // 2 MB = 32768 records
        if (MODE == 4 && (i % local)) nxt = (idx & ~0x3Fu) | (nxt & 0x3Fu);       // 4 KB = 64 records
        idx = nxt;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
double run(const uint4 *t, uint32_t mask, int blocks, int iters, int local, uint32_t *out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        k_chase<MODE><<<blocks, 256>>>(t, mask, iters, local, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    return (double)blocks * 256 * iters / best * 1e-6;   // G steps / s
}

int main() {
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    uint32_t *out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    const int iters = 200;
    for (int lg = 19; lg <= 28; lg += (lg < 22 ? 3 : 2)) {        // 32 MB .. 16 GB
        const size_t n_rec = (size_t)1 << lg;
        uint4 *t; CK(hipMalloc(&t, n_rec * 64));
        k_fill<<<4096, 256>>>(t, n_rec * 4);
        CK(hipDeviceSynchronize());
        for (int bpc : {2, 5, 8}) {
            const int blocks = cus * bpc;
            printf("table %6zu MB  blocks/CU %d : 64B %.1f  128B %.1f  pair-in-line %.1f  2MB-local(8) %.1f  4KB-local(8) %.1f  G steps/s\n",
                   n_rec * 64 >> 20, bpc,
                   run<0>(t, (uint32_t)(n_rec - 1), blocks, iters, 1, out),
                   run<1>(t, (uint32_t)(n_rec - 1), blocks, iters, 1, out),
                   run<2>(t, (uint32_t)(n_rec - 1), blocks, iters, 1, out),
                   run<3>(t, (uint32_t)(n_rec - 1), blocks, iters, 8, out),
                   run<4>(t, (uint32_t)(n_rec - 1), blocks, iters, 8, out));
            fflush(stdout);
        }
        CK(hipFree(t));
    }
    return 0;
}
