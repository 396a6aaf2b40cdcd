// Microbenchmark: dependent random fetches of 64-byte records by every lane of resident
// waves, three ways.  A: each lane loads its own record with four 16-byte loads.  B: the
// four lanes of a quad load one record per instruction (16 bytes each), no exchange (the
// ceiling of what coalescing buys).  C: as B, but straight into LDS (global_load_lds), then
// each lane reads its own record back with four ds_read_b128.  D: as A with 32-byte records
// (two loads per lane, the same number of records: what a node of half the size would buy).
// build: hipcc -O3 --offload-arch=gfx950 -o gather_nodes gather_nodes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mixu(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ __launch_bounds__(256) void k_a(const uint4 *nodes, uint32_t mask, int iters, uint32_t *out) {
    uint32_t idx = mixu(blockIdx.x * 256 + threadIdx.x) & mask;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint4 *p = nodes + 4 * (size_t)idx;
        uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        uint32_t v = a.x + b.y + c.z + d.w;
        acc += v;
        idx = mixu(v + i) & mask;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_b(const uint4 *nodes, uint32_t mask, int iters, uint32_t *out) {
    uint32_t idx = mixu(blockIdx.x * 256 + threadIdx.x) & mask;
    const int q = threadIdx.x & 3;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t v = 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t src = __shfl(idx, (threadIdx.x & 60) | s, 64);
            const uint4 a = nodes[4 * (size_t)src + q];
            v += a.x + a.w;
        }
        acc += v;
        idx = mixu(v + i) & mask;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_d(const uint4 *nodes, uint32_t mask, int iters, uint32_t *out) {
    uint32_t idx = mixu(blockIdx.x * 256 + threadIdx.x) & mask;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint4 *p = nodes + 2 * (size_t)idx;
        uint4 a = p[0], b = p[1];
        uint32_t v = a.x + b.y + a.z + b.w;
        acc += v;
        idx = mixu(v + i) & mask;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

constexpr int kPart = 1040;   // bytes between the four instruction images of one wave (padded)
__global__ __launch_bounds__(256) void k_c(const uint4 *nodes, uint32_t mask, int iters, uint32_t *out) {
    __shared__ __attribute__((aligned(16))) unsigned char stage[4 * 4 * kPart];
    uint32_t idx = mixu(blockIdx.x * 256 + threadIdx.x) & mask;
    const int lane = threadIdx.x & 63, q = lane & 3, wave = threadIdx.x >> 6;
    unsigned char *wbase = stage + wave * 4 * kPart;
    const uint4 *mine = reinterpret_cast<const uint4 *>(wbase + q * kPart + (lane >> 2) * 64);
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t src = __shfl(idx, (lane & 60) | s, 64);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(nodes + 4 * (size_t)src + q),
                (__attribute__((address_space(3))) void *)(wbase + s * kPart), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) lgkmcnt(0) expcnt(0)
        __builtin_amdgcn_wave_barrier();
        const uint4 a = mine[0], b = mine[1], c = mine[2], d = mine[3];
        uint32_t v = a.x + b.y + c.z + d.w;
        acc += v;
        idx = mixu(v + i) & mask;
        __builtin_amdgcn_wave_barrier();
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const uint32_t n_nodes = 1u << (argc > 1 ? atoi(argv[1]) : 18);   // 2^18 x 64 B = 16 MB
    const int iters = 400, blocks = 256 * 5;
    std::vector<uint32_t> h((size_t)n_nodes * 16);
    uint32_t s = 12345;
    for (auto &x : h) { s = s * 1664525u + 1013904223u; x = s; }
    uint4 *d_nodes; uint32_t *d_out;
    CK(hipMalloc(&d_nodes, h.size() * 4));
    CK(hipMalloc(&d_out, blocks * 256 * 4));
    CK(hipMemcpy(d_nodes, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int which = 0; which < 4; ++which) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            if (which == 0) k_a<<<blocks, 256>>>(d_nodes, n_nodes - 1, iters, d_out);
            if (which == 1) k_b<<<blocks, 256>>>(d_nodes, n_nodes - 1, iters, d_out);
            if (which == 2) k_c<<<blocks, 256>>>(d_nodes, n_nodes - 1, iters, d_out);
            if (which == 3) k_d<<<blocks, 256>>>(d_nodes, n_nodes - 1, iters, d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        std::vector<uint32_t> o(blocks * 256);
        CK(hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost));
        uint32_t chk = 0; for (auto x : o) chk ^= x;
        printf("%c: %.3f ms  %.1f G records/s  check %08x\n", "ABCD"[which], best,
               (double)blocks * 256 * iters / best * 1e-6, chk);
    }
    return 0;
}
