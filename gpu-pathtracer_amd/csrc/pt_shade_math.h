// Shading arithmetic of the wavefront path tracer: random numbers, sampling,
// Lambert and GGX lobes, Russian roulette.
//
// The reference has no shading at all (shaders/depth.frag:90-107 writes a depth
// grey; the intended algorithm is only a comment, src/main_granite.cpp:403-415),
// so this file IS the specification.  It is written so that a CPU restatement
// can reproduce it bit for bit: only +, -, *, /, sqrt, min/max, comparisons and
// integer operations, evaluated in the order written, never fused
// (-ffp-contract=off), no library transcendentals.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mipt.h"

namespace mipt {

// ---- random numbers: stateless, keyed on (seed, pixel, sample, dimension) ------------
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t path_key(uint32_t seed, uint32_t pixel,
                                                      uint32_t sample) {
    uint32_t k = mix32(seed + 0x9E3779B9U);
    k = mix32(k ^ pixel);
    k = mix32(k ^ sample);
    return k;
}
// Uniform in [0,1) with 24 random bits.
__host__ __device__ __forceinline__ float rnd(uint32_t key, uint32_t dim) {
    const uint32_t r = mix32(key ^ (dim * 0x85EBCA6BU + 1U));
    return (float)(r >> 8) * 5.9604644775390625e-08f;   // 2^-24
}
// Dimension layout: 0,1 pixel jitter; bounce b uses 2+6b .. 7+6b:
enum { DIM_LOBE = 0, DIM_U1 = 1, DIM_U2 = 2, DIM_RR = 3, DIM_L1 = 4, DIM_L2 = 5, DIMS_PER_BOUNCE = 6 };

// ---- cos/sin of 2*pi*u, u in [0,1): quadrant folding + Taylor polynomials on [0, pi/4] --
__device__ __forceinline__ float poly_sin(float y) {   // y in [0, pi/4]
    const float y2 = y * y;
    float p = 2.7557319e-06f;                 // 1/9!
    p = p * y2 + -1.9841270e-04f;             // -1/7!
    p = p * y2 + 8.3333333e-03f;              // 1/5!
    p = p * y2 + -1.6666667e-01f;             // -1/3!
    p = p * y2 + 1.0f;
    return p * y;
}
__device__ __forceinline__ float poly_cos(float y) {   // y in [0, pi/4]
    const float y2 = y * y;
    float p = 2.4801587e-05f;                 // 1/8!
    p = p * y2 + -1.3888889e-03f;             // -1/6!
    p = p * y2 + 4.1666667e-02f;              // 1/4!
    p = p * y2 + -0.5f;
    p = p * y2 + 1.0f;
    return p;
}
__device__ __forceinline__ void sincos_2pi(float u, float &s, float &c) {
    const float x = u * 4.0f;                 // exact
    int q = (int)x;                           // quadrant 0..3
    if (q > 3) q = 3;
    const float f = x - (float)q;             // exact, in [0,1)
    float sq, cq;                             // sin, cos of f*pi/2
    if (f <= 0.5f) {
        const float y = f * 1.5707963267948966f;
        sq = poly_sin(y); cq = poly_cos(y);
    } else {
        const float y = (1.0f - f) * 1.5707963267948966f;
        sq = poly_cos(y); cq = poly_sin(y);
    }
    if (q == 0) { s = sq; c = cq; }
    else if (q == 1) { s = cq; c = -sq; }
    else if (q == 2) { s = -sq; c = -cq; }
    else { s = -cq; c = sq; }
}

// ---- small vector helpers (component order and association are part of the spec) -----
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ float vdot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 vcross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 vscale(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 vmul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 vdiv(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
// a*sa + b*sb + c*sc, summed left to right
__device__ __forceinline__ V3 vcomb3(V3 a, float sa, V3 b, float sb, V3 c, float sc) {
    return v3((a.x * sa + b.x * sb) + c.x * sc, (a.y * sa + b.y * sb) + c.y * sc,
              (a.z * sa + b.z * sb) + c.z * sc);
}
__device__ __forceinline__ float vmax3(V3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }

// Orthonormal basis around unit n (Duff et al. 2017, branchless form).
__device__ __forceinline__ void basis(V3 n, V3 &t, V3 &b) {
    const float sign = copysignf(1.0f, n.z);
    const float a = -1.0f / (sign + n.z);
    const float bb = (n.x * n.y) * a;
    t = v3(1.0f + (sign * (n.x * n.x)) * a, sign * bb, (-sign) * n.x);
    b = v3(bb, sign + (n.y * n.y) * a, -n.y);
}

// Outcome of shading one path vertex.
struct ShadeResult {
    V3 emitted;      // radiance to add, already multiplied by the incoming throughput
    V3 throughput;   // throughput after this vertex (valid if alive)
    V3 o, d;         // next ray (valid if alive)
    bool alive;
};

// One path vertex.  wi is sampled from the Lambert lobe (cosine-weighted) or the GGX lobe
// (half-vector sampling of the NDF), chosen with probability proportional to the lobes'
// summed reflectances.  bounce = index of this vertex (0 = first hit).
__device__ __forceinline__ ShadeResult shade_vertex(const pt_material &m, V3 v0, V3 v1, V3 v2,
                                                    float b0, float b1, float b2, V3 rd,
                                                    V3 thr, uint32_t key, uint32_t bounce,
                                                    uint32_t max_depth, uint32_t rr_start,
                                                    float ray_offset) {
    ShadeResult r;
    r.alive = false;
    r.throughput = thr;
    r.o = v3(0, 0, 0);
    r.d = v3(0, 0, 0);
    const V3 ke = v3(m.ke[0], m.ke[1], m.ke[2]);
    r.emitted = vmul(thr, ke);
    if (bounce + 1 >= max_depth) return r;

    // geometry
    const V3 p = vcomb3(v0, b0, v1, b1, v2, b2);
    V3 ng = vcross(vsub(v1, v0), vsub(v2, v0));
    const float len2 = vdot(ng, ng);
    if (!(len2 > 0.0f)) return r;                       // degenerate triangle: absorb
    ng = vdiv(ng, sqrtf(len2));
    const V3 n = vdot(ng, rd) > 0.0f ? v3(-ng.x, -ng.y, -ng.z) : ng;
    const V3 wo = v3(-rd.x, -rd.y, -rd.z);
    const float n_wo = vdot(n, wo);
    if (!(n_wo > 0.0f)) return r;                       // grazing to rounding: absorb

    // lobe choice
    const V3 kd = v3(m.kd[0], m.kd[1], m.kd[2]);
    const V3 ks = v3(m.ks[0], m.ks[1], m.ks[2]);
    const float wd = (kd.x + kd.y) + kd.z;
    const float ws = (ks.x + ks.y) + ks.z;
    const float wsum = wd + ws;
    if (!(wsum > 0.0f)) return r;                       // black surface
    const float p_spec = ws / wsum;
    const uint32_t dim0 = 2u + bounce * (uint32_t)DIMS_PER_BOUNCE;
    const float u_lobe = rnd(key, dim0 + DIM_LOBE);
    const float u1 = rnd(key, dim0 + DIM_U1);
    const float u2 = rnd(key, dim0 + DIM_U2);
    float sphi, cphi;
    sincos_2pi(u2, sphi, cphi);
    V3 t, b;
    basis(n, t, b);

    V3 wi, weight;
    if (u_lobe < p_spec) {
        // GGX: sample the half vector from D(h) (n.h), reflect wo about it.
        const float rough = fminf(fmaxf(m.roughness, 0.03f), 1.0f);
        const float alpha = rough * rough;
        const float a2 = alpha * alpha;
        const float cos2 = (1.0f - u1) / (1.0f + (a2 - 1.0f) * u1);
        const float cos_h = sqrtf(cos2);
        const float sin_h = sqrtf(fmaxf(0.0f, 1.0f - cos2));
        const V3 h = vcomb3(t, sin_h * cphi, b, sin_h * sphi, n, cos_h);
        const float wo_h = vdot(wo, h);
        if (!(wo_h > 0.0f)) return r;
        wi = vsub(vscale(h, 2.0f * wo_h), wo);
        const float n_wi = vdot(n, wi);
        if (!(n_wi > 0.0f)) return r;                   // reflected below the surface
        const float n_h = cos_h;
        const float om = 1.0f - wo_h;
        const float om2 = om * om;
        const float om5 = (om2 * om2) * om;
        const V3 F = v3(ks.x + (1.0f - ks.x) * om5, ks.y + (1.0f - ks.y) * om5,
                        ks.z + (1.0f - ks.z) * om5);
        const float g1o = (2.0f * n_wo) / (n_wo + sqrtf(a2 + (1.0f - a2) * (n_wo * n_wo)));
        const float g1i = (2.0f * n_wi) / (n_wi + sqrtf(a2 + (1.0f - a2) * (n_wi * n_wi)));
        const float w = ((g1o * g1i) * wo_h) / ((n_wo * n_h) * p_spec);
        weight = vscale(F, w);
    } else {
        const float rr = sqrtf(u1);
        const float z = sqrtf(fmaxf(0.0f, 1.0f - u1));
        wi = vcomb3(t, rr * cphi, b, rr * sphi, n, z);
        weight = vdiv(kd, 1.0f - p_spec);
    }
    V3 nthr = vmul(thr, weight);

    // Russian roulette
    if (bounce >= rr_start) {
        const float q = fminf(vmax3(nthr), 0.95f);
        const float u_rr = rnd(key, dim0 + DIM_RR);
        if (!(u_rr < q)) return r;
        nthr = vdiv(nthr, q);
    }
    if (!(vmax3(nthr) > 0.0f)) return r;

    const float wl = sqrtf(vdot(wi, wi));
    if (!(wl > 0.0f)) return r;
    const float big = fmaxf(1.0f, fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z))));
    const float offs = ray_offset * big;
    r.o = v3(p.x + n.x * offs, p.y + n.y * offs, p.z + n.z * offs);
    r.d = vdiv(wi, wl);
    r.throughput = nthr;
    r.alive = true;
    return r;
}

} // namespace mipt
