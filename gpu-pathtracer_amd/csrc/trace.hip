// Stand-alone ray/scene intersection kernels (pt_trace_closest*, pt_trace_any*).
// The per-ray traversal code lives in pt_traverse.h.
#include "pt_traverse.h"
#include "internal.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <utility>

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

template <bool COUNT, bool CACHE, bool BIG, bool QUAD = false>
__global__ __launch_bounds__(kBlock, PT_EXTEND_MIN_WAVES) void k_trace_ordered_persistent(
    DeviceScene s, const pt_ray *rays, pt_hit *hits, uint32_t n, bool fixup, PersistentParams pp,
    DeviceTraceCounters *counters, const uint32_t *perm) {
    persistent_closest_hits<COUNT, CACHE, BIG, QUAD>(s, rays, hits, n, fixup, pp, &counters->tie_count,
                                                     counters, perm);
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

// The same query by persistent waves (lane refill, step voting), as the closest-hit kernel.
struct OccludedSink {
    int32_t *out;
    __device__ __forceinline__ void operator()(uint32_t ray, bool occluded) const {
        out[ray] = occluded ? 1 : 0;
    }
};
template <bool CACHE, bool BIG, bool QUAD = false>
__global__ __launch_bounds__(kBlock, PT_EXTEND_MIN_WAVES) void k_trace_any_persistent(
    DeviceScene s, const pt_ray *rays, int32_t *occluded, uint32_t n, PersistentParams pp) {
    persistent_traverse<false, true, CACHE, BIG, OccludedSink, QUAD>(s, rays, n, true, pp, nullptr, nullptr,
                                                                    OccludedSink{occluded});
}

// Two-level scenes: one thread per ray (see closest_hit_two_level).
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_trace_two_level(DeviceScene s, const pt_ray *rays,
                                                           pt_hit *hits, uint64_t n, bool fixup,
                                                           bool exact,
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
        tie = closest_hit_two_level<COUNT>(s, ro, rd, s_stack + threadIdx.x, fixup, exact, shapeid,
                                           primid, b0, b1, b2, t, nb, nt);
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

__global__ __launch_bounds__(kBlock) void k_trace_any_two_level(DeviceScene s, const pt_ray *rays,
                                                               int32_t *occluded, uint64_t n,
                                                               bool exact) {
    __shared__ int32_t s_stack[kTraceStack * kBlock];
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 ro = reinterpret_cast<const float4 *>(rays + i)[0];
    const float4 rd = reinterpret_cast<const float4 *>(rays + i)[1];
    occluded[i] = occluded_two_level(s, ro, rd, s_stack + threadIdx.x, exact) ? 1 : 0;
}

// Two-level scenes by persistent waves (see persistent_traverse_two_level).
// (QUAD: four-child nodes inside the instances; those kernels keep no LDS copy of the top and,
// like the single-level four-child kernels, are not held to five waves per SIMD.)
template <bool COUNT, bool BIG, bool CACHE, bool QUAD = false>
__global__ __launch_bounds__(kBlock, QUAD ? PT_EXTEND_MIN_WAVES : PT_TWO_LEVEL_MIN_WAVES)
void k_trace_two_level_persistent(
    DeviceScene s, const pt_ray *rays, pt_hit *hits, uint32_t n, bool fixup, PersistentParams pp,
    DeviceTraceCounters *counters) {
    persistent_traverse_two_level<COUNT, false, BIG, HitSink, CACHE, QUAD>(s, rays, n, fixup, pp,
                                                                          &counters->tie_count, counters,
                                                                          HitSink{hits});
}
template <bool BIG, bool CACHE, bool QUAD = false>
__global__ __launch_bounds__(kBlock, PT_EXTEND_MIN_WAVES) void k_trace_any_two_level_persistent(
    DeviceScene s, const pt_ray *rays, int32_t *occluded, uint32_t n, PersistentParams pp) {
    persistent_traverse_two_level<false, true, BIG, OccludedSink, CACHE, QUAD>(s, rays, n, true, pp, nullptr,
                                                                              nullptr, OccludedSink{occluded});
}

// The rays the persistent closest-hit kernel handed over, one wave each (long_ray_search).  The
// last wave to finish empties the list for the next launch: by then every wave has read the count.
template <bool BIG>
__global__ __launch_bounds__(256) void k_trace_long_rays(DeviceScene s, const pt_ray *rays, pt_hit *hits,
                                                         bool fixup, PersistentParams pp,
                                                         unsigned int *tie_count,
                                                         DeviceTraceCounters *work) {
    __shared__ int32_t s_stk[4][kLongStack];
    uint32_t *const list = pp.long_list;
    const uint32_t listed = list[0];
    const uint32_t count = listed < pp.long_cap ? listed : pp.long_cap;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
    for (uint32_t i = wave; i < count; i += n_waves)
        long_ray_search<BIG>(s, rays, list[kLongListHeader + i], s_stk[threadIdx.x >> 6], fixup, tie_count,
                             work, HitSink{hits});
    if ((threadIdx.x & 63u) == 0u) {
        __threadfence();
        if (atomicAdd(&list[1], 1u) == n_waves - 1u) {
            if (work && count) atomicAdd(&work->long_rays, count);
            list[0] = 0u;
            list[1] = 0u;
            __threadfence();
        }
    }
}

// Sort key of a ray: 30-bit Morton code of its origin's cell in a 1024^3 grid over the scene's
// root box.  Rays that start close together walk the same part of the tree; on scenes much
// larger than the caches that is worth more than the sort costs.
__device__ __forceinline__ unsigned int spread10(unsigned int v) {
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
// ob: bits per axis of the origin cell (<= 10); db: bits per axis of the direction, as a point of
// the octahedral map of the sphere (0: origin only).  The origin cell is the major part of the
// key: rays of one cell come out grouped by direction.
__global__ __launch_bounds__(kBlock) void k_ray_keys(DeviceScene s, const pt_ray *rays, uint32_t n,
                                                     unsigned int *keys, uint32_t *vals, int ob, int db) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 lo = s.nodes[0], hi = s.nodes[1];
    const float4 o = reinterpret_cast<const float4 *>(rays + i)[0];
    const float p[3] = {o.x, o.y, o.z}, a[3] = {lo.x, lo.y, lo.z}, b[3] = {hi.x, hi.y, hi.z};
    unsigned int q[3];
    const float cells = (float)(1u << ob);
    for (int k = 0; k < 3; ++k) {
        const float ext = b[k] - a[k];
        float x = ext > 0.f ? (p[k] - a[k]) / ext : 0.f;
        x = fminf(fmaxf(x * cells, 0.f), cells - 1.f);      // NaN -> 0
        q[k] = (unsigned int)x;
    }
    unsigned int key = ((spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2])) & ((1u << (3 * ob)) - 1u);
    if (db > 0) {
        const float4 d = reinterpret_cast<const float4 *>(rays + i)[1];
        const float l1 = fabsf(d.x) + fabsf(d.y) + fabsf(d.z);
        float u = l1 > 0.f ? d.x / l1 : 0.f, v = l1 > 0.f ? d.y / l1 : 0.f;
        if (d.z < 0.f) {                                     // fold the lower hemisphere outwards
            const float fu = (1.f - fabsf(v)) * (u >= 0.f ? 1.f : -1.f);
            const float fv = (1.f - fabsf(u)) * (v >= 0.f ? 1.f : -1.f);
            u = fu; v = fv;
        }
        const float bins = (float)(1u << db);
        const unsigned int qu = (unsigned int)fminf(fmaxf((u * 0.5f + 0.5f) * bins, 0.f), bins - 1.f);
        const unsigned int qv = (unsigned int)fminf(fmaxf((v * 0.5f + 0.5f) * bins, 0.f), bins - 1.f);
        // the two direction coordinates interleaved, so that neighbours in the key are neighbours
        // on the sphere
        unsigned int dm = 0;
        for (int k = 0; k < db; ++k) dm |= (((qu >> k) & 1u) << (2 * k + 1)) | (((qv >> k) & 1u) << (2 * k));
        key = (key << (2 * db)) | dm;
    }
    keys[i] = key;
    vals[i] = i;
}

// Breadth-first from the root: the node taken from the queue at position i becomes cache node i,
// and a child that is itself queued (while there is room) is renamed kLdsNodeBase + its queue
// position.  The queue is kept: it names each cache node's place in wnodes.  One thread: a few
// hundred dependent fetches, once per scene.  (Taking the pending node with the largest box
// first instead -- the nodes a ray is most likely to meet -- was measured on the benchmark
// scene: 31.5 instead of 31.7 of a ray's 67.4 slab tests served from LDS, launch time within
// 0.5 %: breadth first already holds the nodes that matter.)
__global__ void k_build_top_cache(const float4 *wnodes, uint32_t n_wnodes, float4 *cache,
                                  int32_t *queue, uint32_t *n_out, int capacity, uint32_t root) {
    if (blockIdx.x || threadIdx.x) return;
    int count = 0;
    if (n_wnodes > 1 && capacity > 0) {
        queue[0] = (int32_t)root;
        count = 1;
        for (int i = 0; i < count; ++i) {
            const float4 *np = wnodes + 4 * (size_t)queue[i];
            int4 ch = reinterpret_cast<const int4 *>(np)[3];
            int32_t c[2] = {ch.x, ch.y};
            for (int k = 0; k < 2; ++k)
                if (c[k] >= 0 && count < capacity) {
                    queue[count] = c[k];
                    c[k] = kLdsNodeBase + count;
                    ++count;
                }
            cache[4 * i + 0] = np[0];
            cache[4 * i + 1] = np[1];
            cache[4 * i + 2] = np[2];
            reinterpret_cast<int4 *>(cache)[4 * i + 3] = make_int4(c[0], c[1], ch.z, ch.w);
        }
    }
    *n_out = (uint32_t)count;
}

// Four-child nodes from the two-child ones, one thread per two-child node i: the children of
// record i are i's grandchildren (a child of i that is a leaf stays a child), so a traversal
// from record 0 only ever reaches the records of even depth; the others are built all the same,
// which keeps this a single independent pass with the numbering -- and so the locality in
// memory -- of the two-child array.  A child's box is stored as 8-bit steps of a per-node grid,
// rounded outwards with the very fused multiply-add the traversal decodes it with, so the
// decoded box never lies inside the exact one.
__device__ __forceinline__ int32_t quad_leaf_ref(int32_t wide_leaf) {      // ~(16-byte position)
    return ~(int32_t)(4u * ((uint32_t)(~wide_leaf) / (uint32_t)PT_TRI_RECORD_16B));
}
__global__ __launch_bounds__(256) void k_build_quad(const float4 *wnodes, uint32_t n_wnodes,
                                                    const float4 *wtris, float4 *qnodes,
                                                    float4 *qtris, uint32_t *bad) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_wnodes) return;
    float lo[4][3], hi[4][3];
    int32_t ref[4] = {kNoChild, kNoChild, kNoChild, kNoChild};
    int m = 0;
    auto child_box = [](const float4 *np, int c, float *l, float *h) {
        const float4 xy = np[c], z = np[2];
        l[0] = xy.x; h[0] = xy.y; l[1] = xy.z; h[1] = xy.w;
        l[2] = c ? z.z : z.x; h[2] = c ? z.w : z.y;
    };
    const float4 *np = wnodes + 4 * (size_t)i;
    const int4 ch = reinterpret_cast<const int4 *>(np)[3];
    const int32_t c2[2] = {ch.x, ch.y};
    for (int c = 0; c < 2; ++c) {
        if (c2[c] < 0) {                               // a leaf: stays a child, and its record is made
            child_box(np, c, lo[m], hi[m]);
            ref[m] = quad_leaf_ref(c2[c]);
            const float4 *src = wtris + (size_t)(uint32_t)(~c2[c]);
            float4 *dst = qtris + (size_t)(uint32_t)(~ref[m]);
            const float4 s2 = src[2];
            dst[0] = src[0];
            dst[1] = src[1];
            dst[2] = make_float4(s2.x, s2.y, lo[m][0], lo[m][1]);
            dst[3] = make_float4(lo[m][2], hi[m][0], hi[m][1], hi[m][2]);
            ++m;
        } else {
            const float4 *gp = wnodes + 4 * (size_t)c2[c];
            const int4 gch = reinterpret_cast<const int4 *>(gp)[3];
            const int32_t g2[2] = {gch.x, gch.y};
            for (int g = 0; g < 2; ++g) {
                child_box(gp, g, lo[m], hi[m]);
                ref[m] = g2[g] >= 0 ? g2[g] : quad_leaf_ref(g2[g]);
                ++m;
            }
        }
    }
    float org[3], scale[3];
    uint32_t ql[3] = {0, 0, 0}, qh[3] = {0, 0, 0};
    bool ok = true;
    for (int a = 0; a < 3; ++a) {
        float mn = lo[0][a], mx = hi[0][a];
        for (int k = 1; k < m; ++k) {
            mn = fminf(mn, lo[k][a]);
            mx = fmaxf(mx, hi[k][a]);
        }
        org[a] = mn;
        float sc = (mx - mn) / 255.0f;
        ok = ok && fabsf(mn) < INFINITY && fabsf(mx) < INFINITY && sc == sc && sc < INFINITY;
        if (ok)
            for (int guard = 0; guard < 64 && __builtin_fmaf(255.0f, sc, mn) < mx; ++guard)
                sc = __uint_as_float(__float_as_uint(sc) + 1u);        // next float up (sc >= 0)
        ok &= !(__builtin_fmaf(255.0f, sc, mn) < mx);
        scale[a] = sc;
        for (int k = 0; k < m && ok; ++k) {
            int q = sc > 0.f ? (int)fminf(fmaxf(floorf((lo[k][a] - mn) / sc), 0.f), 255.f) : 0;
            while (q > 0 && __builtin_fmaf((float)q, sc, mn) > lo[k][a]) --q;
            while (q < 255 && __builtin_fmaf((float)(q + 1), sc, mn) <= lo[k][a]) ++q;
            ok &= !(__builtin_fmaf((float)q, sc, mn) > lo[k][a]);
            ql[a] |= (uint32_t)q << (8 * k);
            int r = sc > 0.f ? (int)fminf(fmaxf(ceilf((hi[k][a] - mn) / sc), 0.f), 255.f) : 0;
            while (r < 255 && __builtin_fmaf((float)r, sc, mn) < hi[k][a]) ++r;
            while (r > 0 && __builtin_fmaf((float)(r - 1), sc, mn) >= hi[k][a]) --r;
            ok &= !(__builtin_fmaf((float)r, sc, mn) < hi[k][a]);
            qh[a] |= (uint32_t)r << (8 * k);
        }
    }
    if (!ok) atomicOr(bad, 1u);
    float4 *q = qnodes + 4 * (size_t)i;
    q[0] = make_float4(org[0], org[1], org[2], scale[0]);
    q[1] = make_float4(scale[1], scale[2], __uint_as_float(ql[0]), __uint_as_float(ql[1]));
    q[2] = make_float4(__uint_as_float(ql[2]), __uint_as_float(qh[0]), __uint_as_float(qh[1]),
                       __uint_as_float(qh[2]));
    reinterpret_cast<int4 *>(q)[3] = make_int4(ref[0], ref[1], ref[2], ref[3]);
}

// k_build_top_cache over the four-child nodes.
__global__ void k_build_quad_top_cache(const float4 *qnodes, uint32_t n_nodes, float4 *cache,
                                       int32_t *queue, uint32_t *n_out, int capacity) {
    if (blockIdx.x || threadIdx.x) return;
    int count = 0;
    if (n_nodes > 0 && capacity > 0) {
        queue[0] = 0;
        count = 1;
        for (int i = 0; i < count; ++i) {
            const float4 *np = qnodes + 4 * (size_t)queue[i];
            const int4 ch = reinterpret_cast<const int4 *>(np)[3];
            int32_t c[4] = {ch.x, ch.y, ch.z, ch.w};
            for (int k = 0; k < 4; ++k)
                if (c[k] >= 0 && count < capacity) {
                    queue[count] = c[k];
                    c[k] = kLdsNodeBase + count;
                    ++count;
                }
            cache[4 * i + 0] = np[0];
            cache[4 * i + 1] = np[1];
            cache[4 * i + 2] = np[2];
            reinterpret_cast<int4 *>(cache)[4 * i + 3] = make_int4(c[0], c[1], c[2], c[3]);
        }
    }
    *n_out = (uint32_t)count;
}

inline unsigned int grid_for(uint64_t n) { return (unsigned int)((n + kBlock - 1) / kBlock); }

} // namespace

void launch_build_quad(hipStream_t st, const float4 *wnodes, uint32_t n_wnodes, const float4 *wtris,
                       uint32_t n_tri, float4 *qnodes, float4 *qtris, uint32_t *bad) {
    (void)n_tri;
    hipLaunchKernelGGL(k_build_quad, dim3((n_wnodes + 255u) / 256u), dim3(256), 0, st, wnodes, n_wnodes,
                       wtris, qnodes, qtris, bad);
}

void launch_build_quad_top_cache(hipStream_t st, const float4 *qnodes, uint32_t n_wnodes, float4 *cache,
                                 int32_t *global_index, uint32_t *n_out) {
    hipLaunchKernelGGL(k_build_quad_top_cache, dim3(1), dim3(1), 0, st, qnodes, n_wnodes, cache,
                       global_index, n_out, (int)kNodeCacheMax);
}

void launch_build_top_cache(hipStream_t st, const float4 *wnodes, uint32_t n_wnodes, float4 *cache,
                            int32_t *global_index, uint32_t *n_out, uint32_t root) {
    hipLaunchKernelGGL(k_build_top_cache, dim3(1), dim3(1), 0, st, wnodes, n_wnodes, cache,
                       global_index, n_out, (int)kNodeCacheMax, root);
}

namespace {
__global__ void k_build_tri_leaf(const float4 *nodes, uint32_t n_nodes, uint32_t link_enc,
                                 uint32_t n_tri, int32_t *tri_leaf, uint32_t *multi) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_nodes) return;
    const int32_t payload = decode_link(nodes[2 * (size_t)i].w, link_enc);
    if (payload < 0 || (uint32_t)payload >= n_tri) return;
    if (atomicCAS(&tri_leaf[payload], -1, (int32_t)i) != -1) atomicOr(multi, 1u);
}
// keys[i] = triangle of leaf node i, or 0xFFFFFFFF for an inner node; vals[i] = i
__global__ void k_leaf_keys(const float4 *nodes, uint32_t n_nodes, uint32_t link_enc, uint32_t n_tri,
                            uint32_t *keys, uint32_t *vals) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_nodes) return;
    const int32_t payload = decode_link(nodes[2 * (size_t)i].w, link_enc);
    keys[i] = (payload >= 0 && (uint32_t)payload < n_tri) ? (uint32_t)payload : 0xFFFFFFFFu;
    vals[i] = i;
}
// sorted by triangle (stable, so each triangle's leaves ascend): where each triangle's run starts
__global__ void k_leaf_runs(const uint32_t *keys, uint32_t n_nodes, uint32_t n_tri, uint32_t *first) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= n_nodes) return;
    const uint32_t k = keys[j];
    if (k == 0xFFFFFFFFu) {
        if (j == 0 || keys[j - 1] != 0xFFFFFFFFu) first[n_tri] = j;      // end of the last run
        return;
    }
    if (j == 0 || keys[j - 1] != k) first[k] = j;
    if (j + 1 == n_nodes) first[n_tri] = n_nodes;                          // no inner node: one leaf
}
__global__ void k_map_face_ids(pt_hit *hits, uint64_t n, const int2 *face_ids) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    int2 *h = reinterpret_cast<int2 *>(hits + i);      // shapeid, primid
    const int32_t tri = h->y;
    if (tri >= 0) *h = face_ids[tri];
}
} // namespace

void launch_map_face_ids(hipStream_t st, pt_hit *hits, uint64_t n, const int2 *face_ids) {
    if (!n || !face_ids) return;
    hipLaunchKernelGGL(k_map_face_ids, dim3((unsigned)((n + 255u) / 256u)), dim3(256), 0, st, hits, n,
                       face_ids);
}

void launch_build_tri_leaf(hipStream_t st, const float4 *nodes, uint32_t n_nodes, uint32_t link_enc,
                           uint32_t n_tri, int32_t *tri_leaf, uint32_t *multi) {
    hipLaunchKernelGGL(k_build_tri_leaf, dim3((n_nodes + 255u) / 256u), dim3(256), 0, st, nodes,
                       n_nodes, link_enc, n_tri, tri_leaf, multi);
}

hipError_t build_tri_leaf_lists(hipStream_t st, const float4 *nodes, uint32_t n_nodes, uint32_t link_enc,
                                uint32_t n_tri, uint32_t *first, int32_t *list, DeviceBuffer &scratch) {
    size_t sort_bytes = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                                      (uint32_t *)nullptr, (uint32_t *)nullptr, (int)n_nodes, 0, 32, st);
    if (e != hipSuccess) return e;
    const size_t arr = (4 * (size_t)n_nodes + 255) / 256 * 256;
    if (scratch.reserve(3 * arr + sort_bytes) != PT_OK) return hipErrorOutOfMemory;
    char *base = (char *)scratch.ptr;
    uint32_t *k0 = (uint32_t *)base, *k1 = (uint32_t *)(base + arr), *v0 = (uint32_t *)(base + 2 * arr);
    const unsigned int grid = (n_nodes + 255u) / 256u;
    hipLaunchKernelGGL(k_leaf_keys, dim3(grid), dim3(256), 0, st, nodes, n_nodes, link_enc, n_tri, k0, v0);
    e = hipcub::DeviceRadixSort::SortPairs(base + 3 * arr, sort_bytes, k0, k1, v0, (uint32_t *)list, (int)n_nodes,
                                           0, 32, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_leaf_runs, dim3(grid), dim3(256), 0, st, k1, n_nodes, n_tri, first);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    return e;
}

unsigned int persistent_grid(const void *kernel, int block, const ExtendConfig &cfg,
                             unsigned int lds_bytes) {
    // Never launch more persistent workgroups than can be resident at once: the surplus would
    // only start after the first ones have finished their whole streams.  The answer depends
    // on the kernel and on its LDS size, not on the GPU (one model per process).
    static std::mutex mu;
    static std::map<std::pair<const void *, unsigned int>, int> resident;
    static std::set<std::pair<const void *, int>> granted;
    int per_cu;
    {
        std::lock_guard<std::mutex> lock(mu);
        // more than 64 KB of LDS per workgroup must be granted per kernel and device
        if (lds_bytes > 64u * 1024u && granted.insert({kernel, cfg.device}).second)
            (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        int &slot = resident[{kernel, lds_bytes}];
        if (slot == 0) {
            int nb = 0;
            const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, lds_bytes);
            slot = (e == hipSuccess && nb > 0) ? nb : 1;
            // The query admits workgroups whose LDS adds up to the full 160 KB of a compute unit;
            // in practice the last one then does not become resident and runs as a second round
            // (measured: 5 x 32 KB and 3 x 53 KB take 1.4-2x as long as 5 x 31 KB).
            if (lds_bytes) slot = std::max(1, std::min(slot, (int)(kLdsBudgetPerCu / lds_bytes)));
        }
        per_cu = slot;
    }
    const unsigned int grid = (unsigned int)(cfg.num_cus * std::min(per_cu, cfg.blocks_per_cu));
    return cfg.max_grid ? std::min(grid, cfg.max_grid) : grid;
}

void launch_long_rays(hipStream_t st, const DeviceScene &s, const pt_ray *rays, pt_hit *hits, bool fixup,
                      const PersistentParams &pp, unsigned int *tie_count, DeviceTraceCounters *work) {
    if (!pp.long_list || pp.tail_budget <= 0 || s.tlas) return;
    constexpr unsigned int kLongGrid = 64;               // 256 waves: the lists are short
    if (s.big)
        hipLaunchKernelGGL(k_trace_long_rays<true>, dim3(kLongGrid), dim3(256), 0, st, s, rays, hits, fixup, pp,
                           tie_count, work);
    else
        hipLaunchKernelGGL(k_trace_long_rays<false>, dim3(kLongGrid), dim3(256), 0, st, s, rays, hits, fixup, pp,
                           tie_count, work);
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

void launch_trace_two_level(hipStream_t st, const DeviceScene &s, const pt_ray *rays, pt_hit *hits,
                            uint64_t n, bool fixup, bool exact, DeviceTraceCounters *counters,
                            bool count_work, const ExtendConfig &cfg) {
    if (n == 0) return;
    if (!exact && cfg.persistent_blocks > 0) {
        const PersistentParams pp = persistent_params(cfg, s);
        using K = void (*)(DeviceScene, const pt_ray *, pt_hit *, uint32_t, bool, PersistentParams, DeviceTraceCounters *);
        // template arguments: counting, 64-bit record addresses, cached top of the top-level tree
        static const K table[8] = {
            k_trace_two_level_persistent<false, false, false>, k_trace_two_level_persistent<false, false, true>,
            k_trace_two_level_persistent<false, true, false>,  k_trace_two_level_persistent<false, true, true>,
            k_trace_two_level_persistent<true, false, false>,  k_trace_two_level_persistent<true, false, true>,
            k_trace_two_level_persistent<true, true, false>,   k_trace_two_level_persistent<true, true, true>};
        static const K quad_table[4] = {
            k_trace_two_level_persistent<false, false, false, true>, k_trace_two_level_persistent<false, true, false, true>,
            k_trace_two_level_persistent<true, false, false, true>,  k_trace_two_level_persistent<true, true, false, true>};
        const K kernel = cfg.quad && s.qnodes
                             ? quad_table[(count_work ? 2 : 0) | (s.big ? 1 : 0)]
                             : table[(count_work ? 4 : 0) | (s.big ? 2 : 0) | (pp.cache_nodes > 0 ? 1 : 0)];
        launch_persistent(kernel, st, grid_for(n), cfg, pp, s, rays, hits, (uint32_t)n, fixup, pp, counters);
        return;
    }
    if (count_work)
        hipLaunchKernelGGL((k_trace_two_level<true>), dim3(grid_for(n)), dim3(kBlock), 0, st, s,
                           rays, hits, n, fixup, exact, counters);
    else
        hipLaunchKernelGGL((k_trace_two_level<false>), dim3(grid_for(n)), dim3(kBlock), 0, st, s,
                           rays, hits, n, fixup, exact, counters);
}

void launch_trace_ordered(hipStream_t st, const DeviceScene &s, const pt_ray *rays, pt_hit *hits,
                          uint64_t n, bool fixup, DeviceTraceCounters *counters,
                          bool count_work, const ExtendConfig &cfg, DeviceBuffer *sort_scratch,
                          hipEvent_t *bracket) {
    if (n == 0) return;
    if (s.tlas) {
        if (bracket) (void)hipEventRecord(bracket[0], st);
        launch_trace_two_level(st, s, rays, hits, n, fixup, false, counters, count_work, cfg);
        if (bracket) (void)hipEventRecord(bracket[1], st);
        return;
    }
    if (cfg.persistent_blocks > 0) {
        const uint32_t *perm = nullptr;
        if (cfg.sort_rays && sort_scratch) {
            size_t sort_bytes = 0;
            const int ob = std::min(10, std::max(1, cfg.sort_origin_bits));
            const int db = std::min((32 - 3 * ob) / 2, std::max(0, cfg.sort_dir_bits));
            const int key_bits = 3 * ob + 2 * db;
            hipError_t e = hipcub::DeviceRadixSort::SortPairs(
                nullptr, sort_bytes, (unsigned int *)nullptr, (unsigned int *)nullptr,
                (uint32_t *)nullptr, (uint32_t *)nullptr, (int)n, 0, key_bits, st);
            const size_t arr = (4 * (size_t)n + 255) / 256 * 256;
            if (e == hipSuccess && sort_scratch->reserve(4 * arr + sort_bytes) == PT_OK) {
                char *base = (char *)sort_scratch->ptr;
                unsigned int *k0 = (unsigned int *)base, *k1 = (unsigned int *)(base + arr);
                uint32_t *v0 = (uint32_t *)(base + 2 * arr), *v1 = (uint32_t *)(base + 3 * arr);
                hipLaunchKernelGGL(k_ray_keys, dim3(grid_for(n)), dim3(kBlock), 0, st, s, rays,
                                   (uint32_t)n, k0, v0, ob, db);
                e = hipcub::DeviceRadixSort::SortPairs(base + 4 * arr, sort_bytes, k0, k1, v0, v1,
                                                       (int)n, 0, key_bits, st);
                if (e == hipSuccess) perm = v1;
            }
            // on any failure the rays are simply taken in queue order
        }
        const PersistentParams pp = persistent_params(cfg, s);
        const bool cached = pp.cache_nodes > 0;
        // template arguments: counting, cached top of the tree, 64-bit record addresses
        using K = void (*)(DeviceScene, const pt_ray *, pt_hit *, uint32_t, bool, PersistentParams,
                           DeviceTraceCounters *, const uint32_t *);
        static const K table[8] = {
            k_trace_ordered_persistent<false, false, false>, k_trace_ordered_persistent<false, false, true>,
            k_trace_ordered_persistent<false, true, false>,  k_trace_ordered_persistent<false, true, true>,
            k_trace_ordered_persistent<true, false, false>,  k_trace_ordered_persistent<true, false, true>,
            k_trace_ordered_persistent<true, true, false>,   k_trace_ordered_persistent<true, true, true>};
        // the same over the four-child nodes (scenes that have them)
        static const K quad_table[8] = {
            k_trace_ordered_persistent<false, false, false, true>, k_trace_ordered_persistent<false, false, true, true>,
            k_trace_ordered_persistent<false, true, false, true>,  k_trace_ordered_persistent<false, true, true, true>,
            k_trace_ordered_persistent<true, false, false, true>,  k_trace_ordered_persistent<true, false, true, true>,
            k_trace_ordered_persistent<true, true, false, true>,   k_trace_ordered_persistent<true, true, true, true>};
        const bool quad = cfg.quad && s.qnodes;
        const K k = (quad ? quad_table : table)[(count_work ? 4 : 0) | (cached ? 2 : 0) | (s.big ? 1 : 0)];
        if (bracket) (void)hipEventRecord(bracket[0], st);
        launch_persistent(k, st, grid_for(n), cfg, pp, s, rays, hits, (uint32_t)n, fixup, pp, counters, perm);
        launch_long_rays(st, s, rays, hits, fixup, pp, &counters->tie_count, counters);
        if (bracket) (void)hipEventRecord(bracket[1], st);
        return;
    }
    const unsigned int grid = grid_for(n);
    if (bracket) (void)hipEventRecord(bracket[0], st);
    if (count_work)
        hipLaunchKernelGGL((k_trace_ordered<true>), dim3(grid), dim3(kBlock), 0, st, s, rays, hits,
                           n, fixup, counters);
    else
        hipLaunchKernelGGL((k_trace_ordered<false>), dim3(grid), dim3(kBlock), 0, st, s, rays,
                           hits, n, fixup, counters);
    if (bracket) (void)hipEventRecord(bracket[1], st);
}

void launch_trace_any(hipStream_t st, const DeviceScene &s, const pt_ray *rays, int32_t *occluded,
                      uint64_t n, bool exact_order, const ExtendConfig &cfg) {
    if (n == 0) return;
    if (s.tlas) {
        if (!exact_order && cfg.persistent_blocks > 0) {
            const PersistentParams pp = persistent_params(cfg, s);
            using K = void (*)(DeviceScene, const pt_ray *, int32_t *, uint32_t, PersistentParams);
            static const K table[4] = {
                k_trace_any_two_level_persistent<false, false>, k_trace_any_two_level_persistent<false, true>,
                k_trace_any_two_level_persistent<true, false>,  k_trace_any_two_level_persistent<true, true>};
            static const K quad_table[2] = {k_trace_any_two_level_persistent<false, false, true>,
                                            k_trace_any_two_level_persistent<true, false, true>};
            const K kernel = cfg.quad && s.qnodes ? quad_table[s.big ? 1 : 0]
                                                  : table[(s.big ? 2 : 0) | (pp.cache_nodes > 0 ? 1 : 0)];
            launch_persistent(kernel, st, grid_for(n), cfg, pp, s, rays, occluded, (uint32_t)n, pp);
            return;
        }
        hipLaunchKernelGGL(k_trace_any_two_level, dim3(grid_for(n)), dim3(kBlock), 0, st, s, rays,
                           occluded, n, exact_order);
        return;
    }
    const bool wide = !exact_order && s.n_tri >= 2;
    if (wide && cfg.persistent_blocks > 0) {
        const PersistentParams pp = persistent_params(cfg, s);
        using K = void (*)(DeviceScene, const pt_ray *, int32_t *, uint32_t, PersistentParams);
        static const K table[8] = {
            k_trace_any_persistent<false, false>, k_trace_any_persistent<false, true>,
            k_trace_any_persistent<true, false>, k_trace_any_persistent<true, true>,
            k_trace_any_persistent<false, false, true>, k_trace_any_persistent<false, true, true>,
            k_trace_any_persistent<true, false, true>, k_trace_any_persistent<true, true, true>};
        const bool quad = cfg.quad && s.qnodes;
        launch_persistent(table[(quad ? 4 : 0) | (pp.cache_nodes > 0 ? 2 : 0) | (s.big ? 1 : 0)], st, grid_for(n), cfg,
                          pp, s, rays, occluded, (uint32_t)n, pp);
        return;
    }
    hipLaunchKernelGGL(k_trace_any, dim3(grid_for(n)), dim3(kBlock), 0, st, s, rays, occluded, n,
                       wide);
}

// ---- the closest-hit query without any allocation ----------------------------------------
// Everything the kernels above need beyond the caller's rays and hits -- the counter block and
// the global slots behind the LDS stacks -- lies in this array, which every GPU the library
// runs on holds for the life of the process.  In host code `scratch` is only a handle for
// hipGetSymbolAddress; the address on each device is looked up once (static_scratch_init) and
// that pointer is what memsets and kernel arguments use.
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {
constexpr int kMaxDevices = 64;
void *g_scratch_ptr[kMaxDevices];       // per device: that device's copy of `scratch`
std::atomic<bool> g_scratch_known[kMaxDevices];
}

int static_scratch_init(int device) {
    if (device < 0 || device >= kMaxDevices)
        return fail(PT_ERR_UNSUPPORTED, "static scratch: device index beyond " + std::to_string(kMaxDevices));
    if (g_scratch_known[device].load(std::memory_order_acquire)) return PT_OK;
    void *ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) return hip_fail(e, "hipGetSymbolAddress(scratch)");
    g_scratch_ptr[device] = ptr;        // contexts racing here store the same value
    g_scratch_known[device].store(true, std::memory_order_release);
    return PT_OK;
}

int launch_trace_static(hipStream_t st, int device, const DeviceScene &s, const pt_ray *rays,
                        pt_hit *hits, uint64_t n, uint32_t flags, ExtendConfig cfg) {
    if (device < 0 || device >= kMaxDevices || !g_scratch_known[device].load(std::memory_order_acquire))
        return fail(PT_ERR_DEVICE, "static scratch of this device was not initialised");
    unsigned char *base = (unsigned char *)g_scratch_ptr[device];
    DeviceTraceCounters *counters = (DeviceTraceCounters *)base;
    static_assert(sizeof(DeviceTraceCounters) <= kStaticCounterBytes, "counter block");
    // the array keeps the last call's contents: the counters must start at zero, the stack
    // slots are written before they are read
    const hipError_t e = hipMemsetAsync(counters, 0, sizeof(DeviceTraceCounters), st);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(static counters)");
    cfg.spill = (int32_t *)(base + kStaticCounterBytes);
    cfg.spill_entries = kStaticSpillEntries;
    cfg.max_grid = kStaticMaxGrid;
    cfg.sort_rays = 0;                  // a sorted call needs room in proportion to n
    cfg.long_list = nullptr;            // one kernel: every wave finishes its own rays
    const bool exact = (flags & (PT_TRACE_EXACT_ORDER | PT_TRACE_MOLLER_TRUMBORE)) != 0;
    const bool fixup = (flags & PT_TRACE_NO_TIE_FIXUP) == 0;
    if (s.tlas)
        launch_trace_two_level(st, s, rays, hits, n, fixup, exact, counters, false, cfg);
    else if (exact)
        launch_trace_skiplink(st, s, rays, hits, n, (flags & PT_TRACE_MOLLER_TRUMBORE) != 0, nullptr);
    else
        launch_trace_ordered(st, s, rays, hits, n, fixup, counters, false, cfg, nullptr);
    launch_map_face_ids(st, hits, n, s.face_ids);
    return PT_OK;
}

} // namespace mipt
