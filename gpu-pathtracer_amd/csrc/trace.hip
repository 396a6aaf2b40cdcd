// Stand-alone ray/scene intersection kernels (pt_trace_closest*, pt_trace_any*).
// The per-ray traversal code lives in pt_traverse.h.
#include "pt_traverse.h"

#include <algorithm>

namespace mipt {

namespace {

constexpr int kBlock = kTraceBlock;

template <bool MT, bool COUNT>
__global__ __launch_bounds__(kBlock) void k_trace_skiplink(DeviceScene s, const pt_ray *rays,
                                                          pt_hit *hits, uint64_t n,
                                                          DeviceTraceCounters *counters) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    unsigned long long nb = 0, nt = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float4 ro = reinterpret_cast<const float4 *>(rays + i)[0];
        const float4 rd = reinterpret_cast<const float4 *>(rays + i)[1];
        float b0, b1, b2, tbest;
        int32_t shapeid, primid;
        traverse_skiplink<MT, COUNT>(s, ro, rd, shapeid, primid, b0, b1, b2, tbest, nb, nt);
        store_hit(hits, i, shapeid, primid, b0, b1, b2, tbest);
    }
    if (COUNT) {
        nb = wave_sum(nb);
        nt = wave_sum(nt);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&counters->box_tests, nb);
            atomicAdd(&counters->tri_tests, nt);
        }
    }
}

template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_trace_ordered(DeviceScene s, const pt_ray *rays,
                                                         pt_hit *hits, uint64_t n, bool fixup,
                                                         DeviceTraceCounters *counters) {
    __shared__ int32_t s_stack[kTraceStack * kBlock];
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    unsigned long long nb = 0, nt = 0;
    bool tie = false;
    if (i < n) {
        const float4 ro = reinterpret_cast<const float4 *>(rays + i)[0];
        const float4 rd = reinterpret_cast<const float4 *>(rays + i)[1];
        float b0, b1, b2, t;
        int32_t shapeid, primid;
        tie = closest_hit<COUNT>(s, ro, rd, s_stack + threadIdx.x, fixup, shapeid, primid, b0, b1,
                                 b2, t, nb, nt);
        store_hit(hits, i, shapeid, primid, b0, b1, b2, t);
    }
    count_ties(tie, &counters->tie_count);
    if (COUNT) {
        nb = wave_sum(nb);
        nt = wave_sum(nt);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&counters->box_tests, nb);
            atomicAdd(&counters->tri_tests, nt);
        }
    }
}

template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_trace_ordered_persistent(DeviceScene s,
                                                                    const pt_ray *rays,
                                                                    pt_hit *hits, uint32_t n,
                                                                    bool fixup, int refill, int leaf_lanes,
                                                                    DeviceTraceCounters *counters) {
    __shared__ int32_t s_stack[(kTraceStack + 1) * kBlock];   // + the bottom sentinel
    persistent_closest_hits<COUNT>(s, rays, hits, n, fixup, refill, leaf_lanes, s_stack + threadIdx.x,
                                   &counters->tie_count, counters);
}

// IntersectSceneAny, shaders/RadeonRays/bvh.glslh:378-414 (leaf test :350-375, :289-310).
// wide: walk the two-child nodes nearer child first (same answer, see occluded_wide).
__global__ __launch_bounds__(kBlock) void k_trace_any(DeviceScene s, const pt_ray *rays,
                                                     int32_t *occluded, uint64_t n, bool wide) {
    __shared__ int32_t s_stack[kTraceStack * kBlock];
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 ro = reinterpret_cast<const float4 *>(rays + i)[0];
    const float4 rd = reinterpret_cast<const float4 *>(rays + i)[1];
    const bool occ = wide ? occluded_wide(s, ro, rd, s_stack + threadIdx.x)
                          : occluded_skiplink(s, ro, rd);
    occluded[i] = occ ? 1 : 0;
}

inline unsigned int grid_for(uint64_t n) { return (unsigned int)((n + kBlock - 1) / kBlock); }

} // namespace

unsigned int persistent_grid(const void *kernel, int block, const ExtendConfig &cfg, int *cache) {
    // Never launch more persistent workgroups than can be resident at once: the surplus
    // would only start after the first ones have finished their whole streams.
    if (*cache == 0) {
        int nb = 0;
        const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, 0);
        *cache = (e == hipSuccess && nb > 0) ? nb : 4;
    }
    return (unsigned int)(cfg.num_cus * std::min(*cache, cfg.blocks_per_cu));
}

void launch_trace_skiplink(hipStream_t st, const DeviceScene &s, const pt_ray *rays, pt_hit *hits,
                           uint64_t n, bool moller_trumbore, DeviceTraceCounters *counters) {
    if (n == 0) return;
    const unsigned int grid = (unsigned int)std::min<uint64_t>(grid_for(n), 1u << 20);
    const bool count = counters != nullptr;
    if (moller_trumbore) {
        if (count)
            hipLaunchKernelGGL((k_trace_skiplink<true, true>), dim3(grid), dim3(kBlock), 0, st, s,
                               rays, hits, n, counters);
        else
            hipLaunchKernelGGL((k_trace_skiplink<true, false>), dim3(grid), dim3(kBlock), 0, st, s,
                               rays, hits, n, counters);
    } else {
        if (count)
            hipLaunchKernelGGL((k_trace_skiplink<false, true>), dim3(grid), dim3(kBlock), 0, st, s,
                               rays, hits, n, counters);
        else
            hipLaunchKernelGGL((k_trace_skiplink<false, false>), dim3(grid), dim3(kBlock), 0, st,
                               s, rays, hits, n, counters);
    }
}

void launch_trace_ordered(hipStream_t st, const DeviceScene &s, const pt_ray *rays, pt_hit *hits,
                          uint64_t n, bool fixup, DeviceTraceCounters *counters,
                          bool count_work, const ExtendConfig &cfg) {
    if (n == 0) return;
    if (cfg.persistent_blocks > 0) {
        static int occ[2] = {0, 0};
        const void *k = count_work ? (const void *)k_trace_ordered_persistent<true>
                                   : (const void *)k_trace_ordered_persistent<false>;
        const unsigned int grid = (unsigned int)std::min<uint64_t>(
            grid_for(n), persistent_grid(k, kBlock, cfg, &occ[count_work ? 1 : 0]));
        if (count_work)
            hipLaunchKernelGGL((k_trace_ordered_persistent<true>), dim3(grid), dim3(kBlock), 0, st,
                               s, rays, hits, (uint32_t)n, fixup, cfg.refill, cfg.leaf_lanes, counters);
        else
            hipLaunchKernelGGL((k_trace_ordered_persistent<false>), dim3(grid), dim3(kBlock), 0, st,
                               s, rays, hits, (uint32_t)n, fixup, cfg.refill, cfg.leaf_lanes, counters);
        return;
    }
    const unsigned int grid = grid_for(n);
    if (count_work)
        hipLaunchKernelGGL((k_trace_ordered<true>), dim3(grid), dim3(kBlock), 0, st, s, rays, hits,
                           n, fixup, counters);
    else
        hipLaunchKernelGGL((k_trace_ordered<false>), dim3(grid), dim3(kBlock), 0, st, s, rays,
                           hits, n, fixup, counters);
}

void launch_trace_any(hipStream_t st, const DeviceScene &s, const pt_ray *rays, int32_t *occluded,
                      uint64_t n, bool exact_order) {
    if (n == 0) return;
    const bool wide = !exact_order && s.n_tri >= 2;
    hipLaunchKernelGGL(k_trace_any, dim3(grid_for(n)), dim3(kBlock), 0, st, s, rays, occluded, n,
                       wide);
}

} // namespace mipt
