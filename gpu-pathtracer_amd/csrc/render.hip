// Wavefront path tracer for gfx950: raygen -> (extend -> shade)* -> accumulate.
//
// Replaces the fragment-shader "megakernel" of the reference (shaders/depth.frag, drawn
// once per frame at src/main_granite.cpp:419-445) with the loop its author sketched at
// src/main_granite.cpp:403-415: seed camera rays, intersect, terminate on miss / Russian
// roulette, accumulate, compact the survivors into the next ray queue.
//
// Data layout in HBM: rays and hits are 32-byte records indexed by queue slot (one
// coalesced 2 KiB run per wavefront); path state (throughput, radiance, key) is indexed by
// path id and reached through the queue's path-id array.  Queues are compacted with one
// atomic per workgroup (ballot + prefix count per wave, prefix over the waves in LDS).
#include "render.h"

#include <algorithm>
#include <atomic>

#include "pt_camera_math.h"
#include "pt_shade_math.h"
#include "pt_traverse.h"

namespace mipt {

namespace {

constexpr int kBlock = kTraceBlock;

constexpr int kWideBlock = 1024;   // raygen / shade: 16 waves share one queue atomic

__global__ __launch_bounds__(kWideBlock) void k_pt_raygen(RenderLaunch rl, RenderWorkspace ws,
                                                          unsigned int *count_out) {
    __shared__ unsigned int s_q[17];
    const uint64_t i = (uint64_t)blockIdx.x * kWideBlock + threadIdx.x;
    const uint64_t total = (uint64_t)rl.n_samples * rl.map.n_local;
    bool valid = false;
    float4 o = make_float4(0, 0, 0, 0), d = make_float4(0, 0, 0, 0);
    if (i < total) {
        const uint32_t j = (uint32_t)(i / rl.map.n_local), lp = (uint32_t)(i % rl.map.n_local);
        uint32_t x, y;
        valid = local_to_pixel(rl.map, lp, x, y);
        if (valid) {
            const uint32_t key = path_key(rl.p.seed, y * rl.map.width + x, rl.sample0 + j);
            float jx = 0.5f, jy = 0.5f;
            if (!(rl.p.flags & (PT_RENDER_PIXEL_CENTRE | PT_RENDER_DEPTH_GREY))) {
                jx = rnd(key, 0);
                jy = rnd(key, 1);
            }
            make_camera_ray(rl.cam, rl.map.width, rl.map.height, (float)x + jx, (float)y + jy, o, d);
            ws.key[i] = key;
            ws.throughput[i] = make_float4(1.f, 1.f, 1.f, -1.f);   // w: density of the last sampled direction
        }
        ws.radiance[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const uint32_t slot = block_queue_append(valid, count_out, s_q);
    if (valid) {
        float4 *r = reinterpret_cast<float4 *>(ws.rays[0] + slot);
        r[0] = o;
        r[1] = d;
        ws.queue[0][slot] = (uint32_t)i;
    }
}

// extend: closest hit of every queued ray.
template <bool COUNT>
__global__ __launch_bounds__(kBlock, PT_EXTEND_MIN_WAVES) void k_pt_extend(DeviceScene s, const pt_ray *rays,
                                                     pt_hit *hits, const unsigned int *count,
                                                     unsigned int *tie_count,
                                                     DeviceTraceCounters *work) {
    __shared__ int32_t s_stack[kTraceStack * kBlock];
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t n = *count;
    if ((uint64_t)blockIdx.x * kBlock >= n) return;      // whole workgroup past the queue end
    bool tie = false;
    unsigned long long nb = 0, nt = 0;
    if (i < n) {
        const float4 ro = reinterpret_cast<const float4 *>(rays + i)[0];
        const float4 rd = reinterpret_cast<const float4 *>(rays + i)[1];
        float b0, b1, b2, t;
        int32_t shapeid, primid;
        tie = closest_hit<COUNT>(s, ro, rd, s_stack + threadIdx.x, true, shapeid, primid, b0, b1,
                                 b2, t, nb, nt);
        store_hit(hits, i, shapeid, primid, b0, b1, b2, t);
    }
    count_ties(tie, tie_count);
    if (COUNT) {
        nb = wave_sum(nb);
        nt = wave_sum(nt);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&work->box_tests, nb);
            atomicAdd(&work->tri_tests, nt);
        }
    }
}

// extend / occlude on a two-level scene: one thread per ray.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_pt_extend_two_level(DeviceScene s, const pt_ray *rays,
                                                               pt_hit *hits,
                                                               const unsigned int *count,
                                                               unsigned int *tie_count,
                                                               DeviceTraceCounters *work) {
    __shared__ int32_t s_stack[kTraceStack * kBlock];
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t n = *count;
    if ((uint64_t)blockIdx.x * kBlock >= n) return;
    bool tie = false;
    unsigned long long nb = 0, nt = 0;
    if (i < n) {
        const float4 ro = reinterpret_cast<const float4 *>(rays + i)[0];
        const float4 rd = reinterpret_cast<const float4 *>(rays + i)[1];
        float b0, b1, b2, t;
        int32_t shapeid, primid;
        tie = closest_hit_two_level<COUNT>(s, ro, rd, s_stack + threadIdx.x, true, false, shapeid,
                                           primid, b0, b1, b2, t, nb, nt);
        store_hit(hits, i, shapeid, primid, b0, b1, b2, t);
    }
    count_ties(tie, tie_count);
    if (COUNT) {
        nb = wave_sum(nb);
        nt = wave_sum(nt);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&work->box_tests, nb);
            atomicAdd(&work->tri_tests, nt);
        }
    }
}

template <bool COUNT, bool BIG, bool CACHE, bool QUAD = false>
__global__ __launch_bounds__(kBlock, QUAD ? PT_EXTEND_MIN_WAVES : PT_TWO_LEVEL_MIN_WAVES)
void k_pt_extend_two_level_persistent(
    DeviceScene s, const pt_ray *rays, pt_hit *hits, const unsigned int *count,
    unsigned int *tie_count, DeviceTraceCounters *work, PersistentParams pp) {
    persistent_traverse_two_level<COUNT, false, BIG, HitSink, CACHE, QUAD>(s, rays, *count, true, pp, tie_count,
                                                                          work, HitSink{hits});
}

// extend, persistent form: the grid is sized to the machine, not to the queue.  Queue
// chunks of 64 rays are dealt round-robin to the waves (chunk c belongs to wave c % n_waves:
// static, so no work-distribution atomics), and a wave tops its idle lanes up from its own
// stream whenever `refill` or more of them have finished, instead of idling them until its
// slowest ray ends.
template <bool COUNT, bool CACHE, bool BIG, bool QUAD = false>
__global__ __launch_bounds__(kBlock, PT_EXTEND_MIN_WAVES) void k_pt_extend_persistent(
    DeviceScene s, const pt_ray *rays, pt_hit *hits, const unsigned int *count,
    unsigned int *tie_count, DeviceTraceCounters *work, PersistentParams pp) {
    persistent_closest_hits<COUNT, CACHE, BIG, QUAD>(s, rays, hits, *count, true, pp, tie_count, work);
}

// shade: one path vertex per queued hit; survivors are appended to the next queue.
__global__ __launch_bounds__(kWideBlock) void k_pt_shade(DeviceScene s, RenderLaunch rl,
                                                         RenderWorkspace ws, int buf_in,
                                                         const unsigned int *count_in,
                                                         unsigned int *count_out,
                                                         unsigned int *shadow_count,
                                                         uint32_t bounce) {
    __shared__ unsigned int s_q[17];
    const bool nee = (rl.p.flags & PT_RENDER_NEE) != 0 && s.n_lights > 0;
    const uint64_t i = (uint64_t)blockIdx.x * kWideBlock + threadIdx.x;
    const uint64_t n = *count_in;
    if ((uint64_t)blockIdx.x * kWideBlock >= n) return;
    bool alive = false, shadow = false;
    uint32_t pid = 0;
    V3 no = v3(0, 0, 0), nd = v3(0, 0, 0), so = v3(0, 0, 0), sd = v3(0, 0, 0), carry = v3(0, 0, 0);
    float smax = 0.f;
    if (i < n) {
        pid = ws.queue[buf_in][i];
        const int4 hi = reinterpret_cast<const int4 *>(ws.hits + i)[0];
        const float4 hu = reinterpret_cast<const float4 *>(ws.hits + i)[1];
        const float4 thr4 = ws.throughput[pid];
        const V3 thr = v3(thr4.x, thr4.y, thr4.z);
        // Radiance this vertex adds to its path, in two ordered terms; the path's slot is only
        // touched when one is non-zero (adding zeros would not change it).
        V3 add0 = v3(0, 0, 0), add1 = v3(0, 0, 0);
        if (hi.y < 0) {                                     // left the scene
            add0 = v3(thr.x * rl.p.background[0], thr.y * rl.p.background[1],
                      thr.z * rl.p.background[2]);
        } else {
            const float4 rd4 = reinterpret_cast<const float4 *>(ws.rays[buf_in] + i)[1];
            // two-level scenes: hi.x names the instance, hi.y the triangle of its mesh; the
            // shading records are per instanced triangle, in world space
            const int32_t tri = s.tlas ? (int32_t)(s.instances[hi.x].first_tri + (uint32_t)hi.y) : hi.y;
            const TriVerts tv = fetch_tri(s.stris, tri);
            const pt_material m = s.materials[tv.material];
            LightList lights;
            lights.tri = s.light_tri;
            lights.cdf = s.light_cdf;
            lights.count = s.n_lights;
            lights.area = s.light_area;
            const ShadeResult r = shade_vertex(
                m, tv.v0, tv.v1, tv.v2, hu.x, hu.y, hu.z, v3(rd4.x, rd4.y, rd4.z), thr, ws.key[pid],
                bounce, rl.p.max_depth, rl.p.rr_start_depth, rl.p.ray_offset, nee, lights, s.stris,
                s.materials, thr4.w, hu.w);
            add0 = r.emitted;
            alive = r.alive;
            if (alive) {
                ws.throughput[pid] = make_float4(r.throughput.x, r.throughput.y, r.throughput.z, r.q);
                no = r.o;
                nd = r.d;
            }
            if (r.shadow) {
                if (r.smax > 0.0f) {
                    shadow = true;
                    so = r.so; sd = r.sd; carry = r.direct; smax = r.smax;
                } else {                                    // no room for an occluder
                    add1 = r.direct;
                }
            }
        }
        if (add0.x != 0.f || add0.y != 0.f || add0.z != 0.f || add1.x != 0.f || add1.y != 0.f ||
            add1.z != 0.f) {
            float4 L = ws.radiance[pid];
            L.x = (L.x + add0.x) + add1.x;
            L.y = (L.y + add0.y) + add1.y;
            L.z = (L.z + add0.z) + add1.z;
            ws.radiance[pid] = L;
        }
    }
    const uint32_t slot = block_queue_append(alive, count_out, s_q);
    if (alive) {
        float4 *r = reinterpret_cast<float4 *>(ws.rays[buf_in ^ 1] + slot);
        r[0] = make_float4(no.x, no.y, no.z, 1e30f);
        r[1] = make_float4(nd.x, nd.y, nd.z, 0.f);
        ws.queue[buf_in ^ 1][slot] = pid;
    }
    if (nee) {
        const uint32_t ss = block_queue_append(shadow, shadow_count, s_q);
        if (shadow) {
            float4 *r = reinterpret_cast<float4 *>(ws.shadow_rays + ss);
            r[0] = make_float4(so.x, so.y, so.z, smax);
            r[1] = make_float4(sd.x, sd.y, sd.z, 0.f);
            ws.shadow_carry[ss] = make_float4(carry.x, carry.y, carry.z, __uint_as_float(pid));
        }
    }
}

// occlude: IntersectSceneAny (shaders/RadeonRays/bvh.glslh:378-414) for the shadow rays of one
// bounce; a ray that reaches its light adds the radiance it carries to its path.  A path has
// at most one shadow ray per bounce, so the additions to a path are ordered by the launches.
struct ShadowSink {
    const float4 *carry;
    float4 *radiance;
    __device__ __forceinline__ void operator()(uint32_t ray, bool occluded) const {
        if (occluded) return;
        const float4 c = carry[ray];
        const uint32_t pid = __float_as_uint(c.w);
        float4 L = radiance[pid];
        L.x += c.x;
        L.y += c.y;
        L.z += c.z;
        radiance[pid] = L;
    }
};

// One thread per shadow ray.
__global__ __launch_bounds__(kBlock) void k_pt_occlude(DeviceScene s, RenderWorkspace ws,
                                                       const unsigned int *count, bool wide) {
    __shared__ int32_t s_stack[kTraceStack * kBlock];
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t n = *count;
    if (i >= n) return;
    const float4 ro = reinterpret_cast<const float4 *>(ws.shadow_rays + i)[0];
    const float4 rd = reinterpret_cast<const float4 *>(ws.shadow_rays + i)[1];
    const bool occ = wide ? occluded_wide(s, ro, rd, s_stack + threadIdx.x)
                          : occluded_skiplink(s, ro, rd);
    ShadowSink{ws.shadow_carry, ws.radiance}((uint32_t)i, occ);
}

__global__ __launch_bounds__(kBlock) void k_pt_occlude_two_level(DeviceScene s, RenderWorkspace ws,
                                                                 const unsigned int *count) {
    __shared__ int32_t s_stack[kTraceStack * kBlock];
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t n = *count;
    if (i >= n) return;
    const float4 ro = reinterpret_cast<const float4 *>(ws.shadow_rays + i)[0];
    const float4 rd = reinterpret_cast<const float4 *>(ws.shadow_rays + i)[1];
    const bool occ = occluded_two_level(s, ro, rd, s_stack + threadIdx.x, false);
    ShadowSink{ws.shadow_carry, ws.radiance}((uint32_t)i, occ);
}

template <bool BIG, bool CACHE, bool QUAD = false>
__global__ __launch_bounds__(kBlock, PT_EXTEND_MIN_WAVES) void k_pt_occlude_two_level_persistent(
    DeviceScene s, RenderWorkspace ws, const unsigned int *count, PersistentParams pp) {
    persistent_traverse_two_level<false, true, BIG, ShadowSink, CACHE, QUAD>(
        s, ws.shadow_rays, *count, true, pp, nullptr, nullptr, ShadowSink{ws.shadow_carry, ws.radiance});
}

// Persistent waves with lane refill and step voting, as the closest-hit kernel.
template <bool CACHE, bool BIG, bool QUAD = false>
__global__ __launch_bounds__(kBlock, PT_EXTEND_MIN_WAVES) void k_pt_occlude_persistent(
    DeviceScene s, RenderWorkspace ws, const unsigned int *count, PersistentParams pp) {
    persistent_traverse<false, true, CACHE, BIG, ShadowSink, QUAD>(s, ws.shadow_rays, *count, true, pp, nullptr,
                                                                  nullptr, ShadowSink{ws.shadow_carry, ws.radiance});
}

// PT_RENDER_DEPTH_GREY: the reference's per-pixel output (shaders/depth.frag:90-107).
__global__ __launch_bounds__(256) void k_pt_shade_depth(RenderWorkspace ws,
                                                        const unsigned int *count) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= *count) return;
    const uint32_t pid = ws.queue[0][i];
    ws.radiance[pid] = depth_grey(ws.hits[i].shapeid, ws.hits[i].uvwt[3]);
}

// accumulate: film[pixel] += radiance of each of the pass's samples, in sample order, so
// the sum does not depend on how paths were scheduled.
__global__ __launch_bounds__(256) void k_pt_accumulate(RenderLaunch rl, RenderWorkspace ws,
                                                       float4 *film, bool overwrite) {
    const uint32_t lp = blockIdx.x * 256 + threadIdx.x;
    if (lp >= rl.map.n_local) return;
    uint32_t x, y;
    if (!local_to_pixel(rl.map, lp, x, y)) return;
    const uint64_t px = (uint64_t)y * rl.map.width + x;
    if (overwrite) {
        film[px] = ws.radiance[lp];
        return;
    }
    float4 f = film[px];
    for (uint32_t j = 0; j < rl.n_samples; ++j) {
        const float4 L = ws.radiance[(uint64_t)j * rl.map.n_local + lp];
        f.x += L.x;
        f.y += L.y;
        f.z += L.z;
        f.w += 1.0f;
    }
    film[px] = f;
}

// Copy the shard's tiles between the film and a dense buffer (for the RCCL gather).
__global__ __launch_bounds__(256) void k_film_pack(PixelMap map, float4 *film, float4 *packed,
                                                   bool unpack) {
    const uint32_t lp = blockIdx.x * 256 + threadIdx.x;
    if (lp >= map.n_local) return;
    uint32_t x, y;
    if (!local_to_pixel(map, lp, x, y)) {
        if (!unpack) packed[lp] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const uint64_t px = (uint64_t)y * map.width + x;
    if (unpack) film[px] = packed[lp];
    else packed[lp] = film[px];
}

inline unsigned int blocks(uint64_t n, unsigned int b) { return (unsigned int)((n + b - 1) / b); }

} // namespace

void launch_pt_raygen(hipStream_t st, const RenderLaunch &rl, const RenderWorkspace &ws,
                      unsigned int *count_out) {
    const uint64_t total = (uint64_t)rl.n_samples * rl.map.n_local;
    if (!total) return;
    hipLaunchKernelGGL(k_pt_raygen, dim3(blocks(total, kWideBlock)), dim3(kWideBlock), 0, st, rl, ws,
                       count_out);
}

void launch_pt_extend(hipStream_t st, const DeviceScene &s, const RenderWorkspace &ws, int buf,
                      const unsigned int *count, uint64_t max_rays, unsigned int *tie_count,
                      DeviceTraceCounters *work, const ExtendConfig &cfg) {
    if (!max_rays) return;
    const PersistentParams pp = persistent_params(cfg, s);
    if (s.tlas && cfg.persistent_blocks > 0) {
        using K = void (*)(DeviceScene, const pt_ray *, pt_hit *, const unsigned int *, unsigned int *,
                           DeviceTraceCounters *, PersistentParams);
        // template arguments: counting, 64-bit record addresses, cached top of the top-level tree
        static const K table[8] = {
            k_pt_extend_two_level_persistent<false, false, false>, k_pt_extend_two_level_persistent<false, false, true>,
            k_pt_extend_two_level_persistent<false, true, false>,  k_pt_extend_two_level_persistent<false, true, true>,
            k_pt_extend_two_level_persistent<true, false, false>,  k_pt_extend_two_level_persistent<true, false, true>,
            k_pt_extend_two_level_persistent<true, true, false>,   k_pt_extend_two_level_persistent<true, true, true>};
        static const K quad_table[4] = {
            k_pt_extend_two_level_persistent<false, false, false, true>, k_pt_extend_two_level_persistent<false, true, false, true>,
            k_pt_extend_two_level_persistent<true, false, false, true>,  k_pt_extend_two_level_persistent<true, true, false, true>};
        const K kernel = cfg.quad && s.qnodes
                             ? quad_table[(work ? 2 : 0) | (s.big ? 1 : 0)]
                             : table[(work ? 4 : 0) | (s.big ? 2 : 0) | (pp.cache_nodes > 0 ? 1 : 0)];
        launch_persistent(kernel, st, blocks(max_rays, kBlock), cfg, pp, s, (const pt_ray *)ws.rays[buf], ws.hits,
                          count, tie_count, work, pp);
        return;
    }
    if (s.tlas) {
        if (work)
            hipLaunchKernelGGL(k_pt_extend_two_level<true>, dim3(blocks(max_rays, kBlock)),
                               dim3(kBlock), 0, st, s, ws.rays[buf], ws.hits, count, tie_count, work);
        else
            hipLaunchKernelGGL(k_pt_extend_two_level<false>, dim3(blocks(max_rays, kBlock)),
                               dim3(kBlock), 0, st, s, ws.rays[buf], ws.hits, count, tie_count, work);
        return;
    }
    if (cfg.persistent_blocks > 0) {
        // template arguments: counting, cached top of the tree, 64-bit record addresses
        using K = void (*)(DeviceScene, const pt_ray *, pt_hit *, const unsigned int *, unsigned int *,
                           DeviceTraceCounters *, PersistentParams);
        static const K table[8] = {
            k_pt_extend_persistent<false, false, false>, k_pt_extend_persistent<false, false, true>,
            k_pt_extend_persistent<false, true, false>,  k_pt_extend_persistent<false, true, true>,
            k_pt_extend_persistent<true, false, false>,  k_pt_extend_persistent<true, false, true>,
            k_pt_extend_persistent<true, true, false>,   k_pt_extend_persistent<true, true, true>};
        // the same over the four-child nodes (scenes that have them)
        static const K quad_table[8] = {
            k_pt_extend_persistent<false, false, false, true>, k_pt_extend_persistent<false, false, true, true>,
            k_pt_extend_persistent<false, true, false, true>,  k_pt_extend_persistent<false, true, true, true>,
            k_pt_extend_persistent<true, false, false, true>,  k_pt_extend_persistent<true, false, true, true>,
            k_pt_extend_persistent<true, true, false, true>,   k_pt_extend_persistent<true, true, true, true>};
        const bool quad = cfg.quad && s.qnodes;
        const K k = (quad ? quad_table : table)[(work ? 4 : 0) | (pp.cache_nodes > 0 ? 2 : 0) | (s.big ? 1 : 0)];
        launch_persistent(k, st, blocks(max_rays, kBlock), cfg, pp, s, (const pt_ray *)ws.rays[buf], ws.hits, count,
                          tie_count, work, pp);
        launch_long_rays(st, s, (const pt_ray *)ws.rays[buf], ws.hits, true, pp, tie_count, work);
        return;
    }
    if (work)
        hipLaunchKernelGGL(k_pt_extend<true>, dim3(blocks(max_rays, kBlock)), dim3(kBlock), 0, st,
                           s, ws.rays[buf], ws.hits, count, tie_count, work);
    else
        hipLaunchKernelGGL(k_pt_extend<false>, dim3(blocks(max_rays, kBlock)), dim3(kBlock), 0, st,
                           s, ws.rays[buf], ws.hits, count, tie_count, work);
}

void launch_pt_shade(hipStream_t st, const DeviceScene &s, const RenderLaunch &rl,
                     const RenderWorkspace &ws, int buf_in, const unsigned int *count_in,
                     unsigned int *count_out, unsigned int *shadow_count, uint64_t max_rays,
                     uint32_t bounce) {
    if (!max_rays) return;
    hipLaunchKernelGGL(k_pt_shade, dim3(blocks(max_rays, kWideBlock)), dim3(kWideBlock), 0, st, s,
                       rl, ws, buf_in, count_in, count_out, shadow_count, bounce);
}

void launch_pt_occlude(hipStream_t st, const DeviceScene &s, const RenderWorkspace &ws,
                       const unsigned int *shadow_count, uint64_t max_rays,
                       const ExtendConfig &cfg) {
    if (!max_rays) return;
    const PersistentParams pp = persistent_params(cfg, s);
    if (s.tlas && cfg.persistent_blocks > 0) {
        using K = void (*)(DeviceScene, RenderWorkspace, const unsigned int *, PersistentParams);
        static const K table[4] = {
            k_pt_occlude_two_level_persistent<false, false>, k_pt_occlude_two_level_persistent<false, true>,
            k_pt_occlude_two_level_persistent<true, false>,  k_pt_occlude_two_level_persistent<true, true>};
        static const K quad_table[2] = {k_pt_occlude_two_level_persistent<false, false, true>,
                                        k_pt_occlude_two_level_persistent<true, false, true>};
        const K kernel = cfg.quad && s.qnodes ? quad_table[s.big ? 1 : 0]
                                              : table[(s.big ? 2 : 0) | (pp.cache_nodes > 0 ? 1 : 0)];
        launch_persistent(kernel, st, blocks(max_rays, kBlock), cfg, pp, s, ws, shadow_count, pp);
        return;
    }
    if (s.tlas) {
        hipLaunchKernelGGL(k_pt_occlude_two_level, dim3(blocks(max_rays, kBlock)), dim3(kBlock), 0,
                           st, s, ws, shadow_count);
        return;
    }
    if (cfg.persistent_blocks > 0 && s.n_tri >= 2) {
        using K = void (*)(DeviceScene, RenderWorkspace, const unsigned int *, PersistentParams);
        static const K table[8] = {
            k_pt_occlude_persistent<false, false>, k_pt_occlude_persistent<false, true>,
            k_pt_occlude_persistent<true, false>, k_pt_occlude_persistent<true, true>,
            k_pt_occlude_persistent<false, false, true>, k_pt_occlude_persistent<false, true, true>,
            k_pt_occlude_persistent<true, false, true>, k_pt_occlude_persistent<true, true, true>};
        const bool quad = cfg.quad && s.qnodes;
        launch_persistent(table[(quad ? 4 : 0) | (pp.cache_nodes > 0 ? 2 : 0) | (s.big ? 1 : 0)], st,
                          blocks(max_rays, kBlock), cfg, pp, s, ws, shadow_count, pp);
        return;
    }
    hipLaunchKernelGGL(k_pt_occlude, dim3(blocks(max_rays, kBlock)), dim3(kBlock), 0, st, s, ws,
                       shadow_count, s.n_tri >= 2);
}

void launch_pt_shade_depth(hipStream_t st, const RenderWorkspace &ws, const unsigned int *count,
                           uint64_t max_rays) {
    if (!max_rays) return;
    hipLaunchKernelGGL(k_pt_shade_depth, dim3(blocks(max_rays, 256)), dim3(256), 0, st, ws, count);
}

void launch_pt_accumulate(hipStream_t st, const RenderLaunch &rl, const RenderWorkspace &ws,
                          float4 *film, bool overwrite) {
    if (!rl.map.n_local) return;
    hipLaunchKernelGGL(k_pt_accumulate, dim3(blocks(rl.map.n_local, 256)), dim3(256), 0, st, rl, ws,
                       film, overwrite);
}

void launch_film_pack(hipStream_t st, const PixelMap &map, const float4 *film, float4 *packed,
                      bool unpack) {
    if (!map.n_local) return;
    hipLaunchKernelGGL(k_film_pack, dim3(blocks(map.n_local, 256)), dim3(256), 0, st, map,
                       const_cast<float4 *>(film), packed, unpack);
}

} // namespace mipt
