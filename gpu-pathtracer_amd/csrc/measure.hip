// Measurement aid (no counterpart in the reference): the rate at which this GPU serves
// dependent random gathers of fixed-size records -- the access pattern of a BVH traversal
// (a 64-byte node or 48-byte triangle record per step, its address known only once the
// previous record has arrived).  A benchmark runs it beside the traversal kernels, with the
// table sized to fall into the level of the memory hierarchy in question, to get the ceiling
// such a kernel has on the machine it is running on rather than a number from a data sheet.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "../../include/mipt.h"
#include "internal.h"

using namespace mipt;

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(kBlock) void k_fill(uint4 *table, size_t n16) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += stride) {
        const uint32_t b = (uint32_t)i * 4u + (uint32_t)(i >> 30);
        table[i] = make_uint4(mix32(b), mix32(b + 1), mix32(b + 2), mix32(b + 3));
    }
}

// Every lane chases `iters` records: the next index is a hash of what the last fetch
// returned, so no fetch can be issued before the previous one has come back.  PIECES 16-byte
// loads per record, as the traversal kernels fetch theirs.
template <int PIECES>
__global__ __launch_bounds__(kBlock) void k_gather(const uint4 *table, uint32_t mask, int iters,
                                                   uint32_t *out) {
    uint32_t idx = mix32(blockIdx.x * kBlock + threadIdx.x) & mask;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint4 *p = table + (size_t)PIECES * idx;
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < PIECES; ++k) {
            const uint4 q = p[k];
            v += (k & 1) ? q.y + q.w : q.x + q.z;
        }
        acc += v;
        idx = mix32(v + (uint32_t)i) & mask;
    }
    out[blockIdx.x * kBlock + threadIdx.x] = acc;
}

} // namespace

extern "C" int pt_measure_gather(pt_context *ctx, uint64_t table_bytes, uint32_t record_bytes,
                                 uint32_t iterations, double *records_per_s) {
    if (!ctx || !records_per_s) return fail(PT_ERR_INVALID, "pt_measure_gather: null argument");
    if (record_bytes != 16 && record_bytes != 32 && record_bytes != 48 && record_bytes != 64)
        return fail(PT_ERR_INVALID, "pt_measure_gather: record_bytes must be 16, 32, 48 or 64");
    if (iterations == 0 || iterations > (1u << 20))
        return fail(PT_ERR_INVALID, "pt_measure_gather: iterations must be in [1, 2^20]");
    uint64_t n_rec = 64;                       // largest power of two that fits the table
    while (n_rec * 2 * record_bytes <= table_bytes && n_rec < (1ull << 31)) n_rec *= 2;
    if (n_rec * record_bytes > table_bytes)
        return fail(PT_ERR_INVALID, "pt_measure_gather: table smaller than 64 records");
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    const unsigned int grid = (unsigned int)(ctx->num_cus * (ctx->extend_blocks_per_cu > 0 ? ctx->extend_blocks_per_cu : 5));
    uint4 *table = nullptr;
    uint32_t *out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = PT_OK;
    double best_ms = 1e30;
    const size_t n16 = (size_t)n_rec * (record_bytes / 16);
    if ((e = hipMalloc((void **)&table, n16 * 16)) != hipSuccess ||
        (e = hipMalloc((void **)&out, (size_t)grid * kBlock * 4)) != hipSuccess ||
        (e = hipEventCreate(&e0)) != hipSuccess || (e = hipEventCreate(&e1)) != hipSuccess) {
        rc = hip_fail(e, "pt_measure_gather setup");
    } else {
        hipStream_t st = ctx->stream;
        hipLaunchKernelGGL(k_fill, dim3(std::min<size_t>((n16 + kBlock - 1) / kBlock, 65536)),
                           dim3(kBlock), 0, st, table, n16);
        for (int rep = 0; rep < 4 && rc == PT_OK; ++rep) {      // the first run warms caches and TLBs
            (void)hipEventRecord(e0, st);
            const uint32_t mask = (uint32_t)(n_rec - 1);
            switch (record_bytes / 16) {
            case 1: hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(kBlock), 0, st, table, mask, (int)iterations, out); break;
            case 2: hipLaunchKernelGGL(k_gather<2>, dim3(grid), dim3(kBlock), 0, st, table, mask, (int)iterations, out); break;
            case 3: hipLaunchKernelGGL(k_gather<3>, dim3(grid), dim3(kBlock), 0, st, table, mask, (int)iterations, out); break;
            default: hipLaunchKernelGGL(k_gather<4>, dim3(grid), dim3(kBlock), 0, st, table, mask, (int)iterations, out); break;
            }
            (void)hipEventRecord(e1, st);
            if ((e = hipEventSynchronize(e1)) != hipSuccess || (e = hipGetLastError()) != hipSuccess) {
                rc = hip_fail(e, "pt_measure_gather launch");
                break;
            }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0) best_ms = std::min(best_ms, (double)ms);
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (out) (void)hipFree(out);
    if (table) (void)hipFree(table);
    if (rc != PT_OK) return rc;
    *records_per_s = (double)grid * kBlock * iterations / (best_ms * 1e-3);
    return PT_OK;
}
