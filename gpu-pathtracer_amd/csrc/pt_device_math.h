// Device-side ray / box / triangle arithmetic shared by the traversal kernels.
//
// Every function here evaluates the same IEEE binary32 expression tree, in the
// same order, as the shader function it is cited against; the translation
// unit is compiled with -ffp-contract=off so no multiply-add is fused, and HIP's
// default correctly rounded fp32 division is relied on.  min/max are
// fminf/fmaxf (v_min_f32 / v_max_f32: a NaN operand is ignored).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mipt.h"

namespace mipt {

struct RayPre {
    float ox, oy, oz;      // origin
    float ix, iy, iz;      // 1/d                      (bvh.glslh:421)
    float sx, sy, sz;      // shear                    (bvh.glslh:60)
    int axis;              // 0: .yzx, 1: .zxy, 2: identity   (bvh.glslh:55-59)
};

// shaders/RadeonRays/bvh.glslh:51-63 and :421
__device__ __forceinline__ RayPre precompute_ray(float ox, float oy, float oz, float dx, float dy,
                                                 float dz) {
    RayPre r;
    r.ox = ox; r.oy = oy; r.oz = oz;
    r.ix = 1.0f / dx; r.iy = 1.0f / dy; r.iz = 1.0f / dz;
    float px = dx, py = dy, pz = dz;
    r.axis = 2;
    if (fabsf(dx) >= fabsf(dy) && fabsf(dx) >= fabsf(dz)) {
        px = dy; py = dz; pz = dx; r.axis = 0;
    } else if (fabsf(dy) >= fabsf(dz)) {
        px = dz; py = dx; pz = dy; r.axis = 1;
    }
    r.sx = -px / pz;
    r.sy = -py / pz;
    r.sz = 1.0f / pz;
    return r;
}

// shaders/RadeonRays/bvh.glslh:165-177.  Returns hit; t_entry receives t0.
__device__ __forceinline__ bool intersect_box(const RayPre &r, float minx, float maxx, float miny,
                                              float maxy, float minz, float maxz, float maxt,
                                              float &t_entry) {
    const float fx = (maxx - r.ox) * r.ix, nx = (minx - r.ox) * r.ix;
    const float fy = (maxy - r.oy) * r.iy, ny = (miny - r.oy) * r.iy;
    const float fz = (maxz - r.oz) * r.iz, nz = (minz - r.oz) * r.iz;
    const float tmaxx = fmaxf(fx, nx), tminx = fminf(fx, nx);
    const float tmaxy = fmaxf(fy, ny), tminy = fminf(fy, ny);
    const float tmaxz = fmaxf(fz, nz), tminz = fminf(fz, nz);
    const float t1 = fminf(fminf(tmaxx, fminf(tmaxy, tmaxz)), maxt);
    const float t0 = fmaxf(fmaxf(tminx, fmaxf(tminy, tminz)), 0.0f);
    t_entry = t0;
    return t1 >= t0;
}

// shaders/RadeonRays/bvh.glslh:206-287.  tMax is the running nearest distance
// (isect.uvwt.w at :268).  On a hit writes b0,b1,b2,t.
// Written with selects and non-short-circuit predicates: the only branches left are the
// rare fp64 edge re-evaluation and the division of an accepted hit, so a divergent wave
// does not serialise the three axis permutations or the rejection ladder.
__device__ __forceinline__ bool intersect_triangle_watertight(const RayPre &r, float v0x, float v0y,
                                                              float v0z, float v1x, float v1y,
                                                              float v1z, float v2x, float v2y,
                                                              float v2z, float tMax, float &b0,
                                                              float &b1, float &b2, float &t) {
    const float ax = v0x - r.ox, ay = v0y - r.oy, az = v0z - r.oz;
    const float bx = v1x - r.ox, by = v1y - r.oy, bz = v1z - r.oz;
    const float cx = v2x - r.ox, cy = v2y - r.oy, cz = v2z - r.oz;
    const bool k0 = r.axis == 0, k1 = r.axis == 1;      // 0: .yzx   1: .zxy   else identity
    float p0x = k0 ? ay : (k1 ? az : ax), p0y = k0 ? az : (k1 ? ax : ay);
    float p0z = k0 ? ax : (k1 ? ay : az);
    float p1x = k0 ? by : (k1 ? bz : bx), p1y = k0 ? bz : (k1 ? bx : by);
    float p1z = k0 ? bx : (k1 ? by : bz);
    float p2x = k0 ? cy : (k1 ? cz : cx), p2y = k0 ? cz : (k1 ? cx : cy);
    float p2z = k0 ? cx : (k1 ? cy : cz);
    p0x += r.sx * p0z; p0y += r.sy * p0z;
    p1x += r.sx * p1z; p1y += r.sy * p1z;
    p2x += r.sx * p2z; p2y += r.sy * p2z;

    float e0 = p1x * p2y - p1y * p2x;
    float e1 = p2x * p0y - p2y * p0x;
    float e2 = p0x * p1y - p0y * p1x;

    if ((e0 == 0.0f) | (e1 == 0.0f) | (e2 == 0.0f)) {
        const double p2xp1y = (double)p2x * (double)p1y;
        const double p2yp1x = (double)p2y * (double)p1x;
        e0 = (float)(p2yp1x - p2xp1y);
        const double p0xp2y = (double)p0x * (double)p2y;
        const double p0yp2x = (double)p0y * (double)p2x;
        e1 = (float)(p0yp2x - p0xp2y);
        const double p1xp0y = (double)p1x * (double)p0y;
        const double p1yp0x = (double)p1y * (double)p0x;
        e2 = (float)(p1yp0x - p1xp0y);
    }
    const bool any_neg = (e0 < 0.0f) | (e1 < 0.0f) | (e2 < 0.0f);
    const bool any_pos = (e0 > 0.0f) | (e1 > 0.0f) | (e2 > 0.0f);
    const float det = (e0 + e1) + e2;
    p0z *= r.sz; p1z *= r.sz; p2z *= r.sz;
    const float tScaled = (e0 * p0z + e1 * p1z) + e2 * p2z;
    const float lim = tMax * det;
    const bool out_neg = (det < 0.0f) & ((tScaled >= 0.0f) | (tScaled < lim));
    const bool out_pos = (det > 0.0f) & ((tScaled <= 0.0f) | (tScaled > lim));
    if ((any_neg & any_pos) | (det == 0.0f) | out_neg | out_pos) return false;

    const float invDet = 1.0f / det;
    b0 = e0 * invDet;
    b1 = e1 * invDet;
    b2 = e2 * invDet;
    t = tScaled * invDet;
    return true;
}

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return (ax * bx + ay * by) + az * bz;
}

#define PT_MT_EPS 1e-4f /* shaders/RadeonRays/bvh.glslh:179 */

// Shared Moller-Trumbore core of shaders/RadeonRays/bvh.glslh:181-202 and :289-310.
__device__ __forceinline__ void moller_trumbore(float ox, float oy, float oz, float dx, float dy,
                                                float dz, float v1x, float v1y, float v1z,
                                                float v2x, float v2y, float v2z, float v3x,
                                                float v3y, float v3z, float &b1, float &b2,
                                                float &temp) {
    const float e1x = v2x - v1x, e1y = v2y - v1y, e1z = v2z - v1z;
    const float e2x = v3x - v1x, e2y = v3y - v1y, e2z = v3z - v1z;
    // s1 = cross(d, e2)
    const float s1x = dy * e2z - dz * e2y, s1y = dz * e2x - dx * e2z, s1z = dx * e2y - dy * e2x;
    const float invd = 1.0f / dot3(s1x, s1y, s1z, e1x, e1y, e1z);
    const float tx = ox - v1x, ty = oy - v1y, tz = oz - v1z;
    b1 = dot3(tx, ty, tz, s1x, s1y, s1z) * invd;
    // s2 = cross(t, e1)
    const float s2x = ty * e1z - tz * e1y, s2y = tz * e1x - tx * e1z, s2z = tx * e1y - ty * e1x;
    b2 = dot3(dx, dy, dz, s2x, s2y, s2z) * invd;
    temp = dot3(e2x, e2y, e2z, s2x, s2y, s2z) * invd;
}

__device__ __forceinline__ int32_t decode_link(float f, uint32_t enc) {
    return enc == PT_LINKS_INT_BITS ? __float_as_int(f) : (int32_t)f;
}

} // namespace mipt
