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
// Dimension layout: 0,1 pixel jitter; bounce b uses 2+8b .. 9+8b:
enum { DIM_LOBE = 0, DIM_U1 = 1, DIM_U2 = 2, DIM_RR = 3, DIM_LSEL = 4, DIM_L1 = 5, DIM_L2 = 6,
       DIMS_PER_BOUNCE = 8 };

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

// Emissive triangles of a scene, for next-event estimation: triangle indices in ascending
// order and the running sum of their areas (binary32, summed in that order).
struct LightList {
    const int32_t *tri;
    const float *cdf;
    uint32_t count;
    float area;        // cdf[count - 1]
};

#define PT_INV_PI 0.31830988618379067f

// Area of a triangle from its un-normalised normal c = (v1-v0) x (v2-v0).
__host__ __device__ __forceinline__ float tri_area_from_len2(float len2) { return 0.5f * sqrtf(len2); }

// Index of the first entry of cdf[0..n) that exceeds x (n - 1 if none does).
__host__ __device__ __forceinline__ uint32_t cdf_find(const float *cdf, uint32_t n, float x) {
    uint32_t lo = 0, hi = n - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cdf[mid] > x) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

// Power-heuristic weight of a sample drawn with density `mine` against a competing
// technique of density `other`, written so that an infinite density gives 0 or 1, not NaN.
__device__ __forceinline__ float mis_weight(float mine, float other) {
    const float r = other / mine;
    return 1.0f / (1.0f + r * r);
}

// Outcome of shading one path vertex.
struct ShadeResult {
    V3 emitted;      // radiance to add, already multiplied by the incoming throughput
    V3 throughput;   // throughput after this vertex (valid if alive)
    V3 o, d;         // next ray (valid if alive)
    float q;         // density the next direction was drawn with (lobe probability included)
    bool alive;
    bool shadow;     // a shadow ray was generated:
    V3 so, sd;       //   its origin and unit direction
    float smax;      //   its extent (<= 0: nothing can lie in between, count it unoccluded)
    V3 direct;       //   radiance to add if it reaches the light
};

// Vertices and material id of a triangle from its 48-byte shading record (one contiguous
// fetch instead of an index triple and three vertex gathers).
struct TriVerts { V3 v0, v1, v2; int32_t material; };
__device__ __forceinline__ TriVerts fetch_tri(const float4 *stris, int32_t tri) {
    const float4 *p = stris + 3 * (int64_t)tri;
    const float4 q0 = p[0], q1 = p[1], q2 = p[2];
    TriVerts r;
    r.v0 = v3(q0.x, q0.y, q0.z);
    r.v1 = v3(q0.w, q1.x, q1.y);
    r.v2 = v3(q1.z, q1.w, q2.x);
    r.material = __float_as_int(q2.y);
    return r;
}

// One path vertex.  wi is sampled from the Lambert lobe (cosine-weighted) or the GGX lobe
// (half-vector sampling of the NDF), chosen with probability proportional to the lobes'
// summed reflectances.  bounce = index of this vertex (0 = first hit).
//
// With next-event estimation (nee, and the scene has emissive triangles) the vertex also
// aims one shadow ray at a point drawn uniformly by area from the emissive triangles, and
// light reached by a sampled direction is weighted against that technique: `q_in` is the
// density the incoming direction was drawn with (<= 0: a camera ray, weight 1), t_hit the
// distance it travelled.  Each lobe is weighted separately, with the lobe's selection
// probability folded into its density.
__device__ __forceinline__ ShadeResult shade_vertex(const pt_material &m, V3 v0, V3 v1, V3 v2,
                                                    float b0, float b1, float b2, V3 rd,
                                                    V3 thr, uint32_t key, uint32_t bounce,
                                                    uint32_t max_depth, uint32_t rr_start,
                                                    float ray_offset, bool nee,
                                                    const LightList &lights, const float4 *stris,
                                                    const pt_material *materials, float q_in,
                                                    float t_hit) {
    ShadeResult r;
    r.alive = false;
    r.shadow = false;
    r.throughput = thr;
    r.o = v3(0, 0, 0);
    r.d = v3(0, 0, 0);
    r.so = v3(0, 0, 0);
    r.sd = v3(0, 0, 0);
    r.direct = v3(0, 0, 0);
    r.smax = 0.0f;
    r.q = -1.0f;
    const V3 ke = v3(m.ke[0], m.ke[1], m.ke[2]);
    r.emitted = vmul(thr, ke);
    V3 ng = vcross(vsub(v1, v0), vsub(v2, v0));
    const float len2 = vdot(ng, ng);
    const bool use_nee = nee && lights.count > 0;
    if (use_nee && q_in > 0.0f && (ke.x + ke.y) + ke.z > 0.0f && len2 > 0.0f) {
        // This triangle is in the light list: weight the hit against having sampled it.
        const float cos_l = fabsf(vdot(ng, rd)) / sqrtf(len2);
        if (cos_l > 0.0f) {
            const float pl = (t_hit * t_hit) / (cos_l * lights.area);
            r.emitted = vscale(r.emitted, mis_weight(q_in, pl));
        }
    }
    if (bounce + 1 >= max_depth) return r;

    // geometry
    const V3 p = vcomb3(v0, b0, v1, b1, v2, b2);
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
    const float rough = fminf(fmaxf(m.roughness, 0.03f), 1.0f);
    const float alpha = rough * rough;
    const float a2 = alpha * alpha;
    const float big = fmaxf(1.0f, fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z))));
    const float offs = ray_offset * big;
    const V3 org = v3(p.x + n.x * offs, p.y + n.y * offs, p.z + n.z * offs);

    // next-event estimation
    if (use_nee) {
        const float u_sel = rnd(key, dim0 + DIM_LSEL);
        const float ua = rnd(key, dim0 + DIM_L1);
        const float ub = rnd(key, dim0 + DIM_L2);
        const uint32_t li = cdf_find(lights.cdf, lights.count, u_sel * lights.area);
        const int32_t lt = lights.tri[li];
        const TriVerts L = fetch_tri(stris, lt);
        const float su = sqrtf(ua);
        const float c0 = 1.0f - su, c1 = ub * su;
        const float c2 = (1.0f - c0) - c1;
        const V3 y = vcomb3(L.v0, c0, L.v1, c1, L.v2, c2);
        const V3 dv = vsub(y, org);
        const float dist2 = vdot(dv, dv);
        const V3 lc = vcross(vsub(L.v1, L.v0), vsub(L.v2, L.v0));
        const float ll2 = vdot(lc, lc);
        if (dist2 > 0.0f && ll2 > 0.0f) {
            const float dist = sqrtf(dist2);
            const V3 wi = vdiv(dv, dist);
            const float n_wi = vdot(n, wi);
            const float cos_l = fabsf(vdot(lc, wi)) / sqrtf(ll2);
            const float pl = dist2 / (cos_l * lights.area);
            if (n_wi > 0.0f && cos_l > 0.0f && pl > 0.0f) {
                const pt_material lm = materials[L.material];
                // Lambert lobe
                const float qd = (1.0f - p_spec) * (n_wi * PT_INV_PI);
                V3 f = vscale(kd, (n_wi * PT_INV_PI) * mis_weight(pl, qd));
                // GGX lobe
                const V3 hv = v3(wo.x + wi.x, wo.y + wi.y, wo.z + wi.z);
                const float hl2 = vdot(hv, hv);
                if (ws > 0.0f && hl2 > 0.0f) {
                    const V3 h = vdiv(hv, sqrtf(hl2));
                    const float n_h = vdot(n, h), wo_h = vdot(wo, h);
                    if (n_h > 0.0f && wo_h > 0.0f) {
                        const float tt = (n_h * n_h) * (a2 - 1.0f) + 1.0f;
                        const float D = (a2 * PT_INV_PI) / (tt * tt);
                        const float om = 1.0f - wo_h;
                        const float om2 = om * om;
                        const float om5 = (om2 * om2) * om;
                        const V3 F = v3(ks.x + (1.0f - ks.x) * om5, ks.y + (1.0f - ks.y) * om5,
                                        ks.z + (1.0f - ks.z) * om5);
                        const float g1o = (2.0f * n_wo) / (n_wo + sqrtf(a2 + (1.0f - a2) * (n_wo * n_wo)));
                        const float g1i = (2.0f * n_wi) / (n_wi + sqrtf(a2 + (1.0f - a2) * (n_wi * n_wi)));
                        const float qs = p_spec * ((D * n_h) / (4.0f * wo_h));
                        // f_s * n_wi = F D G / (4 n_wo)
                        const float fs = (((D * g1o) * g1i) / (4.0f * n_wo)) * mis_weight(pl, qs);
                        f = v3(f.x + F.x * fs, f.y + F.y * fs, f.z + F.z * fs);
                    }
                }
                const V3 le = v3(lm.ke[0], lm.ke[1], lm.ke[2]);
                r.direct = vdiv(vmul(vmul(thr, f), le), pl);
                r.shadow = vmax3(r.direct) > 0.0f;
                r.so = org;
                r.sd = wi;
                r.smax = dist * 0.999f - 2e-4f;
            }
        }
    }

    const float u_lobe = rnd(key, dim0 + DIM_LOBE);
    const float u1 = rnd(key, dim0 + DIM_U1);
    const float u2 = rnd(key, dim0 + DIM_U2);
    float sphi, cphi;
    sincos_2pi(u2, sphi, cphi);
    V3 t, b;
    basis(n, t, b);

    V3 wi, weight;
    float q;
    if (u_lobe < p_spec) {
        // GGX: sample the half vector from D(h) (n.h), reflect wo about it.
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
        const float tt = (n_h * n_h) * (a2 - 1.0f) + 1.0f;
        const float D = (a2 * PT_INV_PI) / (tt * tt);
        q = p_spec * ((D * n_h) / (4.0f * wo_h));
    } else {
        const float rr = sqrtf(u1);
        const float z = sqrtf(fmaxf(0.0f, 1.0f - u1));
        wi = vcomb3(t, rr * cphi, b, rr * sphi, n, z);
        weight = vdiv(kd, 1.0f - p_spec);
        q = (1.0f - p_spec) * (z * PT_INV_PI);
    }
    V3 nthr = vmul(thr, weight);

    // Russian roulette
    if (bounce >= rr_start) {
        const float qq = fminf(vmax3(nthr), 0.95f);
        const float u_rr = rnd(key, dim0 + DIM_RR);
        if (!(u_rr < qq)) return r;
        nthr = vdiv(nthr, qq);
    }
    if (!(vmax3(nthr) > 0.0f)) return r;

    const float wl = sqrtf(vdot(wi, wi));
    if (!(wl > 0.0f)) return r;
    r.o = org;
    r.d = vdiv(wi, wl);
    r.throughput = nthr;
    r.q = q > 0.0f ? q : -1.0f;
    r.alive = true;
    return r;
}

} // namespace mipt
