// C ABI: film objects and pt_render (the host loop of the wavefront path tracer).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mipt.h"
#include "internal.h"
#include "render.h"

using namespace mipt;

#define HIP_CHECK(expr)                                                  \
    do {                                                                 \
        hipError_t _e = (expr);                                          \
        if (_e != hipSuccess) return hip_fail(_e, #expr);                \
    } while (0)

namespace {

int make_pixel_map(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t shard_index,
                   uint32_t shard_count, PixelMap &m) {
    if (width == 0 || height == 0 || (uint64_t)width * height > 0x7FFFFFFFull)
        return fail(PT_ERR_INVALID, "image size out of range");
    if (tile_size == 0) tile_size = 64;
    if (tile_size % 8 || tile_size > 4096)
        return fail(PT_ERR_INVALID, "tile_size must be a multiple of 8 and at most 4096");
    if (shard_count == 0) shard_count = 1;
    if (shard_index >= shard_count)
        return fail(PT_ERR_INVALID, "shard_index must be below shard_count");
    m.width = width;
    m.height = height;
    m.ts = tile_size;
    m.tiles_x = (width + tile_size - 1) / tile_size;
    const uint32_t tiles_y = (height + tile_size - 1) / tile_size;
    m.n_tiles = m.tiles_x * tiles_y;
    m.shard_index = shard_index;
    m.shard_count = shard_count;
    m.n_owned = m.n_tiles > shard_index ? (m.n_tiles - shard_index + shard_count - 1) / shard_count : 0;
    const uint64_t nl = (uint64_t)m.n_owned * tile_size * tile_size;
    if (nl > 0x7FFFFFFFull) return fail(PT_ERR_INVALID, "too many pixels in one shard");
    m.n_local = (uint32_t)nl;
    return PT_OK;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

} // namespace

extern "C" {

void pt_render_params_default(pt_render_params *p, uint32_t width, uint32_t height) {
    if (!p) return;
    p->width = width;
    p->height = height;
    p->sample_begin = 0;
    p->sample_count = 1;
    p->max_depth = 4;
    p->seed = 1;
    p->rr_start_depth = 3;
    p->flags = PT_RENDER_DEFAULT;
    p->tile_size = 64;
    p->shard_index = 0;
    p->shard_count = 1;
    p->samples_per_pass = 0;
    p->background[0] = p->background[1] = p->background[2] = 0.f;
    p->ray_offset = 1e-4f;
}

int pt_film_create(pt_context *ctx, uint32_t width, uint32_t height, pt_film **out) {
    if (!ctx || !out) return fail(PT_ERR_INVALID, "pt_film_create: null argument");
    *out = nullptr;
    if (width == 0 || height == 0 || (uint64_t)width * height > 0x7FFFFFFFull)
        return fail(PT_ERR_INVALID, "pt_film_create: image size out of range");
    HIP_CHECK(hipSetDevice(ctx->device));
    auto f = std::make_unique<pt_film>();
    f->ctx = ctx;
    f->width = width;
    f->height = height;
    f->owned = true;
    const size_t bytes = sizeof(float4) * (size_t)width * height;
    HIP_CHECK(hipMalloc((void **)&f->data, bytes));
    hipError_t e = hipMemsetAsync(f->data, 0, bytes, ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(f->data);
        return hip_fail(e, "hipMemsetAsync");
    }
    *out = f.release();
    return PT_OK;
}

int pt_film_wrap(pt_context *ctx, uint32_t width, uint32_t height, float *rgba_dev,
                 pt_film **out) {
    if (!ctx || !out || !rgba_dev) return fail(PT_ERR_INVALID, "pt_film_wrap: null argument");
    *out = nullptr;
    if (width == 0 || height == 0 || (uint64_t)width * height > 0x7FFFFFFFull)
        return fail(PT_ERR_INVALID, "pt_film_wrap: image size out of range");
    if ((uintptr_t)rgba_dev % 16) return fail(PT_ERR_INVALID, "pt_film_wrap: buffer not 16-byte aligned");
    auto f = std::make_unique<pt_film>();
    f->ctx = ctx;
    f->width = width;
    f->height = height;
    f->data = reinterpret_cast<float4 *>(rgba_dev);
    f->owned = false;
    *out = f.release();
    return PT_OK;
}

int pt_film_clear(pt_film *film) {
    if (!film) return fail(PT_ERR_INVALID, "pt_film_clear: null film");
    HIP_CHECK(hipSetDevice(film->ctx->device));
    HIP_CHECK(hipMemsetAsync(film->data, 0, sizeof(float4) * (size_t)film->width * film->height,
                             film->ctx->stream));
    return PT_OK;
}

float *pt_film_device_ptr(pt_film *film) { return film ? reinterpret_cast<float *>(film->data) : nullptr; }

int pt_film_read(pt_film *film, float *rgba) {
    if (!film || !rgba) return fail(PT_ERR_INVALID, "pt_film_read: null argument");
    HIP_CHECK(hipSetDevice(film->ctx->device));
    HIP_CHECK(hipMemcpyAsync(rgba, film->data, sizeof(float4) * (size_t)film->width * film->height,
                             hipMemcpyDeviceToHost, film->ctx->stream));
    HIP_CHECK(hipStreamSynchronize(film->ctx->stream));
    return PT_OK;
}

int pt_film_destroy(pt_film *film) {
    if (!film) return PT_OK;
    (void)hipSetDevice(film->ctx->device);
    (void)hipStreamSynchronize(film->ctx->stream);
    if (film->owned && film->data) (void)hipFree(film->data);
    delete film;
    return PT_OK;
}

int pt_shard_layout(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t shard_index,
                    uint32_t shard_count, uint64_t *n_slots, int32_t *xy) {
    if (!n_slots) return fail(PT_ERR_INVALID, "pt_shard_layout: null argument");
    PixelMap m;
    int rc = make_pixel_map(width, height, tile_size, shard_index, shard_count, m);
    if (rc != PT_OK) return rc;
    *n_slots = m.n_local;
    if (xy) {
        for (uint32_t lp = 0; lp < m.n_local; ++lp) {
            uint32_t x, y;
            const bool ok = local_to_pixel(m, lp, x, y);
            xy[2 * (size_t)lp + 0] = ok ? (int32_t)x : -1;
            xy[2 * (size_t)lp + 1] = ok ? (int32_t)y : -1;
        }
    }
    return PT_OK;
}

uint64_t pt_film_packed_size(const pt_film *film, uint32_t tile_size, uint32_t shard_index,
                             uint32_t shard_count) {
    if (!film) return 0;
    PixelMap m;
    if (make_pixel_map(film->width, film->height, tile_size, shard_index, shard_count, m) != PT_OK)
        return 0;
    return m.n_local;
}

static int film_pack_impl(pt_film *film, uint32_t tile_size, uint32_t shard_index,
                          uint32_t shard_count, float *packed_dev, bool unpack) {
    if (!film || !packed_dev) return fail(PT_ERR_INVALID, "pt_film_pack_tiles: null argument");
    PixelMap m;
    int rc = make_pixel_map(film->width, film->height, tile_size, shard_index, shard_count, m);
    if (rc != PT_OK) return rc;
    HIP_CHECK(hipSetDevice(film->ctx->device));
    launch_film_pack(film->ctx->stream, m, film->data, reinterpret_cast<float4 *>(packed_dev), unpack);
    HIP_CHECK(hipGetLastError());
    return PT_OK;
}

int pt_film_pack_tiles(pt_film *film, uint32_t tile_size, uint32_t shard_index,
                       uint32_t shard_count, float *packed_dev) {
    return film_pack_impl(film, tile_size, shard_index, shard_count, packed_dev, false);
}

int pt_film_unpack_tiles(pt_film *film, uint32_t tile_size, uint32_t shard_index,
                         uint32_t shard_count, const float *packed_dev) {
    return film_pack_impl(film, tile_size, shard_index, shard_count,
                          const_cast<float *>(packed_dev), true);
}

int pt_render(pt_context *ctx, const pt_scene *scene, const pt_camera *cam,
              const pt_render_params *params, pt_film *film, pt_render_stats *stats) {
    if (!ctx || !scene || !cam || !params || !film)
        return fail(PT_ERR_INVALID, "pt_render: null argument");
    if (scene->ctx != ctx || film->ctx != ctx)
        return fail(PT_ERR_INVALID, "pt_render: scene or film belongs to another context");
    if (cam->mode != PT_CAMERA_REFERENCE && cam->mode != PT_CAMERA_PINHOLE)
        return fail(PT_ERR_INVALID, "pt_render: unknown camera mode");
    pt_render_params p = *params;
    if (p.width != film->width || p.height != film->height)
        return fail(PT_ERR_INVALID, "pt_render: params and film disagree on the image size");
    const bool depth_mode = (p.flags & PT_RENDER_DEPTH_GREY) != 0;
    if (depth_mode) {
        p.sample_count = 1;
        p.max_depth = 1;
    }
    if (p.sample_count == 0 || p.max_depth == 0 || p.max_depth > 64)
        return fail(PT_ERR_INVALID, "pt_render: sample_count >= 1 and 1 <= max_depth <= 64 required");
    if (!(p.ray_offset >= 0.f)) return fail(PT_ERR_INVALID, "pt_render: ray_offset must be >= 0");

    RenderLaunch rl{};
    rl.p = p;
    rl.cam = *cam;
    int rc = make_pixel_map(p.width, p.height, p.tile_size, p.shard_index, p.shard_count, rl.map);
    if (rc != PT_OK) return rc;
    if (stats) *stats = pt_render_stats{};
    if (rl.map.n_local == 0) return PT_OK;      // more shards than tiles: nothing to do here

    HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    // Samples in flight together: the persistent closest-hit launches end with a tail in which
    // few waves still have work, and the longer the queue the less that tail weighs (1080p on
    // MI355X: 4 / 8 / 16 samples per pass give 4.26 / 4.52 / 4.66 Grays/s).  32 M paths are a
    // 4.5 GB workspace, little on a 288 GB device.
    uint32_t spp_pass = p.samples_per_pass;
    if (spp_pass == 0) {
        const uint64_t target_paths = 32ull << 20;
        spp_pass = (uint32_t)std::max<uint64_t>(1, target_paths / rl.map.n_local);
    }
    spp_pass = std::min(spp_pass, p.sample_count);
    const uint64_t max_paths = (uint64_t)spp_pass * rl.map.n_local;
    if (max_paths > 0xFFFFFF00ull) return fail(PT_ERR_INVALID, "pt_render: too many paths per pass");
    const uint32_t n_pass = (p.sample_count + spp_pass - 1) / spp_pass;

    // Workspace carve-up (256-byte aligned pieces of one allocation).
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    const size_t o_r0 = take(sizeof(pt_ray) * max_paths), o_r1 = take(sizeof(pt_ray) * max_paths);
    const size_t o_q0 = take(4 * max_paths), o_q1 = take(4 * max_paths);
    const size_t o_h = take(sizeof(pt_hit) * max_paths);
    const size_t o_t = take(16 * max_paths), o_L = take(16 * max_paths), o_k = take(4 * max_paths);
    const bool nee = !depth_mode && (p.flags & PT_RENDER_NEE) && scene->dev.n_lights > 0;
    const size_t o_sr = nee ? take(sizeof(pt_ray) * max_paths) : 0;
    const size_t o_sc = nee ? take(16 * max_paths) : 0;
    rc = ctx->render_ws.reserve(off);
    if (rc != PT_OK) return rc;
    char *base = (char *)ctx->render_ws.ptr;
    RenderWorkspace ws;
    ws.rays[0] = (pt_ray *)(base + o_r0);
    ws.rays[1] = (pt_ray *)(base + o_r1);
    ws.queue[0] = (uint32_t *)(base + o_q0);
    ws.queue[1] = (uint32_t *)(base + o_q1);
    ws.hits = (pt_hit *)(base + o_h);
    ws.throughput = (float4 *)(base + o_t);
    ws.radiance = (float4 *)(base + o_L);
    ws.key = (uint32_t *)(base + o_k);
    ws.shadow_rays = nee ? (pt_ray *)(base + o_sr) : nullptr;
    ws.shadow_carry = nee ? (float4 *)(base + o_sc) : nullptr;

    // Queue sizes: per pass one row of max_depth+1 ray counters followed by max_depth
    // shadow-ray counters; the tie counter comes last.
    const uint32_t row = 2 * p.max_depth + 1;
    const size_t n_counts = (size_t)n_pass * row + 1;
    rc = ctx->render_counts.reserve(sizeof(unsigned int) * n_counts);
    if (rc != PT_OK) return rc;
    unsigned int *counts = (unsigned int *)ctx->render_counts.ptr;
    unsigned int *tie_count = counts + (size_t)n_pass * row;
    HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(unsigned int) * n_counts, st));

    DeviceTraceCounters *work = nullptr;
    if (p.flags & PT_RENDER_COUNT_WORK) {
        work = ctx->counters;
        HIP_CHECK(hipMemsetAsync(work, 0, sizeof(DeviceTraceCounters), st));
    }
    // a pass's queues are in pixel order: neighbouring rays start next to each other
    const ExtendConfig ecfg = make_extend_config(ctx, scene, true);
    // Camera rays and the first bounce are coherent enough for the caches to serve the two-child
    // nodes, which cost less arithmetic; from the second bounce on the four-child nodes pay
    // where the scene has them and the policy picked them (4K, 10 M triangles, depth 8: 42.3 ms of
    // extend with two-child nodes throughout, 36.3 with four-child nodes throughout, of which the
    // first two launches take 19.0 instead of 17.8).
    ExtendConfig ecfg_first = ecfg;
    if (ctx->extend_quad < 0 && ecfg.quad) {
        ecfg_first.quad = 0;
        ecfg_first.stack_entries = ctx->extend_stack_entries > 0 ? ctx->extend_stack_entries : 24;
        ecfg_first.cache_nodes = ctx->extend_cache_nodes >= 0 ? ctx->extend_cache_nodes : 96;
    }
    // Kernel timing: one event between every two launches of the call; the span between two
    // neighbouring events is the device time of the launch recorded between them.
    const bool timing = ctx->kernel_timing;
    enum { kRaygen, kExtend, kShade, kOcclude, kAccumulate, kKinds };
    std::vector<int> span_kind;
    size_t n_events = 0;
    if (timing) {
        const size_t need = (size_t)n_pass * (3 * (size_t)p.max_depth + 2) + 1;
        while (ctx->events.size() < need) {
            hipEvent_t e;
            HIP_CHECK(hipEventCreate(&e));
            ctx->events.push_back(e);
        }
        span_kind.reserve(need);
        HIP_CHECK(hipEventRecord(ctx->events[n_events++], st));
    }
#define PT_TIMED(kind, launch)                                              \
    do {                                                                    \
        launch;                                                             \
        if (timing) {                                                       \
            HIP_CHECK(hipEventRecord(ctx->events[n_events++], st));         \
            span_kind.push_back(kind);                                      \
        }                                                                   \
    } while (0)

    for (uint32_t pass = 0; pass < n_pass; ++pass) {
        rl.sample0 = p.sample_begin + pass * spp_pass;
        rl.n_samples = std::min(spp_pass, p.sample_count - pass * spp_pass);
        const uint64_t paths = (uint64_t)rl.n_samples * rl.map.n_local;
        unsigned int *c = counts + (size_t)pass * row;
        PT_TIMED(kRaygen, launch_pt_raygen(st, rl, ws, &c[0]));
        int buf = 0;
        for (uint32_t b = 0; b < p.max_depth; ++b) {
            PT_TIMED(kExtend, launch_pt_extend(st, scene->dev, ws, buf, &c[b], paths, tie_count, work,
                                               b < 2 ? ecfg_first : ecfg));
            if (depth_mode) {
                PT_TIMED(kShade, launch_pt_shade_depth(st, ws, &c[b], paths));
            } else {
                unsigned int *sc = &c[p.max_depth + 1 + b];
                PT_TIMED(kShade, launch_pt_shade(st, scene->dev, rl, ws, buf, &c[b], &c[b + 1], sc, paths, b));
                if (nee && b + 1 < p.max_depth)
                    PT_TIMED(kOcclude, launch_pt_occlude(st, scene->dev, ws, sc, paths, ecfg));
                buf ^= 1;
            }
        }
        PT_TIMED(kAccumulate, launch_pt_accumulate(st, rl, ws, film->data, depth_mode));
    }
#undef PT_TIMED
    HIP_CHECK(hipGetLastError());

    if (stats) {
        std::vector<unsigned int> h(n_counts);
        HIP_CHECK(hipMemcpyAsync(h.data(), counts, sizeof(unsigned int) * n_counts,
                                 hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        for (uint32_t pass = 0; pass < n_pass; ++pass) {
            const unsigned int *c = h.data() + (size_t)pass * row;
            stats->paths += c[0];
            stats->rays_primary += c[0];
            for (uint32_t b = 1; b < p.max_depth; ++b) stats->rays_bounce += c[b];
            for (uint32_t b = 0; b < p.max_depth; ++b) stats->rays_shadow += c[p.max_depth + 1 + b];
        }
        stats->tie_rays = h[(size_t)n_pass * row];
        if (work) {
            DeviceTraceCounters wc;
            HIP_CHECK(hipMemcpy(&wc, work, sizeof(wc), hipMemcpyDeviceToHost));
            stats->box_tests = wc.box_tests;
            stats->tri_tests = wc.tri_tests;
            stats->lds_box_tests = wc.lds_box_tests;
        }
        stats->extend_launches = (uint64_t)n_pass * p.max_depth;
        if (timing) {
            double ms[kKinds] = {0, 0, 0, 0, 0};
            for (size_t k = 0; k + 1 < n_events; ++k) {
                float t = 0.f;
                HIP_CHECK(hipEventElapsedTime(&t, ctx->events[k], ctx->events[k + 1]));
                ms[span_kind[k]] += t;
            }
            stats->extend_ms = ms[kExtend];
            stats->raygen_ms = ms[kRaygen];
            stats->shade_ms = ms[kShade];
            stats->occlude_ms = ms[kOcclude];
            stats->accumulate_ms = ms[kAccumulate];
        }
    }
    return PT_OK;
}

} // extern "C"
