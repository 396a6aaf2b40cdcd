// Internal interface of the wavefront path tracer (render.hip <-> render_api.cpp).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace mipt {

// How the pixels a call renders are enumerated: the image is cut into square tiles of
// `ts` pixels (a multiple of 8); this shard owns tiles shard_index, shard_index +
// shard_count, ...; inside a tile pixels are walked in 8x8 blocks so that one wavefront's
// 64 camera rays form a compact bundle.  Slots that fall outside the image are skipped.
struct PixelMap {
    uint32_t width, height;
    uint32_t ts;            // tile edge in pixels
    uint32_t tiles_x;       // tiles per image row
    uint32_t n_tiles;       // tiles in the image
    uint32_t shard_index, shard_count;
    uint32_t n_owned;       // tiles owned by this shard
    uint32_t n_local;       // n_owned * ts * ts pixel slots
};

__host__ __device__ inline bool local_to_pixel(const PixelMap &m, uint32_t lp, uint32_t &x,
                                               uint32_t &y) {
    const uint32_t per_tile = m.ts * m.ts;
    const uint32_t k = lp / per_tile, inner = lp % per_tile;
    const uint32_t tile = m.shard_index + k * m.shard_count;
    const uint32_t bx = tile % m.tiles_x, by = tile / m.tiles_x;
    const uint32_t sub = inner >> 6, e = inner & 63u;
    const uint32_t nsx = m.ts >> 3;
    const uint32_t sx = sub % nsx, sy = sub / nsx;
    x = bx * m.ts + sx * 8u + (e & 7u);
    y = by * m.ts + sy * 8u + (e >> 3);
    return x < m.width && y < m.height;
}

// Device arrays of one pass.  Path id = sample_in_pass * n_local + local pixel slot.
struct RenderWorkspace {
    pt_ray *rays[2];        // ray queue, double buffered, indexed by queue slot
    uint32_t *queue[2];     // path id of each queue slot
    pt_hit *hits;           // closest hit of each queue slot
    float4 *throughput;     // per path
    float4 *radiance;       // per path: radiance gathered so far
    uint32_t *key;          // per path: random-number key
    // Next-event estimation (null when off): shadow rays of the current bounce by shadow-
    // queue slot, and per slot the radiance the ray carries (xyz) and its path id (w, bits).
    pt_ray *shadow_rays;
    float4 *shadow_carry;
};

struct RenderLaunch {
    pt_render_params p;
    pt_camera cam;
    PixelMap map;
    uint32_t sample0;       // first sample index of the pass
    uint32_t n_samples;     // samples in the pass
};

void launch_pt_raygen(hipStream_t st, const RenderLaunch &rl, const RenderWorkspace &ws,
                      unsigned int *count_out);
void launch_pt_extend(hipStream_t st, const DeviceScene &s, const RenderWorkspace &ws, int buf,
                      const unsigned int *count, uint64_t max_rays, unsigned int *tie_count,
                      DeviceTraceCounters *work, const ExtendConfig &cfg);
void launch_pt_shade(hipStream_t st, const DeviceScene &s, const RenderLaunch &rl,
                     const RenderWorkspace &ws, int buf_in, const unsigned int *count_in,
                     unsigned int *count_out, unsigned int *shadow_count, uint64_t max_rays,
                     uint32_t bounce);
// Shadow rays of one bounce: radiance[path] += carried radiance of every unoccluded ray.
void launch_pt_occlude(hipStream_t st, const DeviceScene &s, const RenderWorkspace &ws,
                       const unsigned int *shadow_count, uint64_t max_rays,
                       const ExtendConfig &cfg);
void launch_pt_shade_depth(hipStream_t st, const RenderWorkspace &ws, const unsigned int *count,
                           uint64_t max_rays);
void launch_pt_accumulate(hipStream_t st, const RenderLaunch &rl, const RenderWorkspace &ws,
                          float4 *film, bool overwrite);
void launch_film_pack(hipStream_t st, const PixelMap &map, const float4 *film, float4 *packed,
                      bool unpack);

} // namespace mipt
