/*
 * pt_shade_oracle.c -- CPU restatement of the path tracer's shading specification.
 *
 * TEST INFRASTRUCTURE ONLY (see pt_oracle.h).
 *
 * PARITY UNPINNED: the reference contains no BRDF, random numbers, bounce loop or
 * accumulation (shaders/depth.frag:90-107 writes a depth grey; the intended algorithm is
 * a comment at src/main_granite.cpp:403-415).  The specification restated here is this
 * project's own; it is written down twice, independently, in C (this file) and in HIP,
 * and the two are required to agree.  Ray/scene intersection inside the path loop is the
 * reference's IntersectSceneClosest (pt_oracle.c).
 *
 * Arithmetic contract: IEEE binary32, no contraction, only + - * / sqrt min max, in the
 * order written; random numbers are integer hashes.  Under that contract the CPU and GPU
 * results are expected to be bit-identical, not merely close.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "pt_oracle.h"

typedef struct { float kd[3]; float roughness; float ks[3]; float pad0; float ke[3]; float pad1; } os_material;

typedef struct {
    float eye[3]; float sx;
    float c0[3];  float sy;
    float c1[3];  uint32_t mode;      /* 0: depth.frag camera, 1: pinhole */
    float c2[3];  float pad;
} os_camera;

typedef struct {
    uint32_t width, height, sample_begin, sample_count, max_depth, seed, rr_start_depth, flags;
    uint32_t tile_size, shard_index, shard_count, samples_per_pass;
    float background[3];
    float ray_offset;
} os_params;

#define OS_FLAG_DEPTH_GREY 1u
#define OS_FLAG_NEE 2u
#define OS_FLAG_PIXEL_CENTRE 4u

/* Emissive triangles with non-zero area in ascending order, and the running binary32 sum
 * of their areas. */
typedef struct { int32_t *tri; float *cdf; uint32_t count; float area; } os_lights;

/* ---- random numbers ---------------------------------------------------------------- */
static uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
static uint32_t path_key(uint32_t seed, uint32_t pixel, uint32_t sample)
{
    uint32_t k = mix32(seed + 0x9E3779B9U);
    k = mix32(k ^ pixel);
    k = mix32(k ^ sample);
    return k;
}
static float rnd(uint32_t key, uint32_t dim)
{
    const uint32_t r = mix32(key ^ (dim * 0x85EBCA6BU + 1U));
    return (float)(r >> 8) * 5.9604644775390625e-08f;
}

/* ---- sin/cos of 2*pi*u ---------------------------------------------------------------- */
static float poly_sin(float y)
{
    const float y2 = y * y;
    float p = 2.7557319e-06f;
    p = p * y2 + -1.9841270e-04f;
    p = p * y2 + 8.3333333e-03f;
    p = p * y2 + -1.6666667e-01f;
    p = p * y2 + 1.0f;
    return p * y;
}
static float poly_cos(float y)
{
    const float y2 = y * y;
    float p = 2.4801587e-05f;
    p = p * y2 + -1.3888889e-03f;
    p = p * y2 + 4.1666667e-02f;
    p = p * y2 + -0.5f;
    p = p * y2 + 1.0f;
    return p;
}
static void sincos_2pi(float u, float *s, float *c)
{
    const float x = u * 4.0f;
    int q = (int)x;
    if (q > 3) q = 3;
    const float f = x - (float)q;
    float sq, cq;
    if (f <= 0.5f) {
        const float y = f * 1.5707963267948966f;
        sq = poly_sin(y); cq = poly_cos(y);
    } else {
        const float y = (1.0f - f) * 1.5707963267948966f;
        sq = poly_cos(y); cq = poly_sin(y);
    }
    if (q == 0) { *s = sq; *c = cq; }
    else if (q == 1) { *s = cq; *c = -sq; }
    else if (q == 2) { *s = -sq; *c = -cq; }
    else { *s = -cq; *c = sq; }
}

/* ---- vectors ---------------------------------------------------------------------------- */
typedef struct { float x, y, z; } v3;
static v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static v3 cross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static v3 divs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static v3 comb3(v3 a, float sa, v3 b, float sb, v3 c, float sc)
{
    return V((a.x * sa + b.x * sb) + c.x * sc, (a.y * sa + b.y * sb) + c.y * sc,
             (a.z * sa + b.z * sb) + c.z * sc);
}
static float max3(v3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }

static void basis(v3 n, v3 *t, v3 *b)
{
    const float sign = copysignf(1.0f, n.z);
    const float a = -1.0f / (sign + n.z);
    const float bb = (n.x * n.y) * a;
    *t = V(1.0f + (sign * (n.x * n.x)) * a, sign * bb, (-sign) * n.x);
    *b = V(bb, sign + (n.y * n.y) * a, -n.y);
}

enum { DIM_LOBE = 0, DIM_U1 = 1, DIM_U2 = 2, DIM_RR = 3, DIM_LSEL = 4, DIM_L1 = 5, DIM_L2 = 6,
       DIMS_PER_BOUNCE = 8 };

#define OS_INV_PI 0.31830988618379067f

static void build_lights(const float *vtx, const int32_t *idx, uint32_t n_tri,
                         const os_material *mats, const int32_t *mat_ids, os_lights *L)
{
    L->tri = (int32_t *)malloc(sizeof(int32_t) * (n_tri ? n_tri : 1));
    L->cdf = (float *)malloc(sizeof(float) * (n_tri ? n_tri : 1));
    L->count = 0;
    float total = 0.0f;
    for (uint32_t t = 0; t < n_tri; ++t) {
        const os_material *m = &mats[mat_ids[t]];
        if (!((m->ke[0] + m->ke[1]) + m->ke[2] > 0.0f)) continue;
        const float *a = vtx + 3 * (size_t)idx[3 * (size_t)t + 0];
        const float *b = vtx + 3 * (size_t)idx[3 * (size_t)t + 1];
        const float *c = vtx + 3 * (size_t)idx[3 * (size_t)t + 2];
        const v3 n = cross(sub(V(b[0], b[1], b[2]), V(a[0], a[1], a[2])),
                           sub(V(c[0], c[1], c[2]), V(a[0], a[1], a[2])));
        const float len2 = dot(n, n);
        if (!(len2 > 0.0f)) continue;
        total += 0.5f * sqrtf(len2);
        L->tri[L->count] = (int32_t)t;
        L->cdf[L->count] = total;
        ++L->count;
    }
    L->area = total;
}

static uint32_t cdf_find(const float *cdf, uint32_t n, float x)
{
    uint32_t lo = 0, hi = n - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cdf[mid] > x) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

static float mis_weight(float mine, float other)
{
    const float r = other / mine;
    return 1.0f / (1.0f + r * r);
}

/* What next-event estimation at a vertex produced. */
typedef struct { int shadow; v3 so, sd, direct; float smax; } os_nee;

/* One path vertex; returns 1 if the path continues (then *thr, *no, *nd are updated). */
static int shade_vertex(const os_material *m, v3 v0, v3 v1, v3 v2, float b0, float b1, float b2,
                        v3 rd, v3 *thr, uint32_t key, uint32_t bounce, const os_params *P,
                        v3 *emitted, v3 *no, v3 *nd, const os_lights *lights, const float *vtx,
                        const int32_t *idx, const os_material *mats, const int32_t *mat_ids,
                        float *q_io, float t_hit, os_nee *nee)
{
    const float q_in = *q_io;
    *q_io = -1.0f;
    nee->shadow = 0;
    const v3 ke = V(m->ke[0], m->ke[1], m->ke[2]);
    *emitted = mul(*thr, ke);
    v3 ng = cross(sub(v1, v0), sub(v2, v0));
    const float len2 = dot(ng, ng);
    const int use_nee = (P->flags & OS_FLAG_NEE) && lights->count > 0;
    if (use_nee && q_in > 0.0f && (ke.x + ke.y) + ke.z > 0.0f && len2 > 0.0f) {
        const float cos_l = fabsf(dot(ng, rd)) / sqrtf(len2);
        if (cos_l > 0.0f) {
            const float pl = (t_hit * t_hit) / (cos_l * lights->area);
            *emitted = scale(*emitted, mis_weight(q_in, pl));
        }
    }
    if (bounce + 1 >= P->max_depth) return 0;

    const v3 p = comb3(v0, b0, v1, b1, v2, b2);
    if (!(len2 > 0.0f)) return 0;
    ng = divs(ng, sqrtf(len2));
    const v3 n = dot(ng, rd) > 0.0f ? V(-ng.x, -ng.y, -ng.z) : ng;
    const v3 wo = V(-rd.x, -rd.y, -rd.z);
    const float n_wo = dot(n, wo);
    if (!(n_wo > 0.0f)) return 0;

    const v3 kd = V(m->kd[0], m->kd[1], m->kd[2]);
    const v3 ks = V(m->ks[0], m->ks[1], m->ks[2]);
    const float wd = (kd.x + kd.y) + kd.z;
    const float ws = (ks.x + ks.y) + ks.z;
    const float wsum = wd + ws;
    if (!(wsum > 0.0f)) return 0;
    const float p_spec = ws / wsum;
    const uint32_t dim0 = 2u + bounce * (uint32_t)DIMS_PER_BOUNCE;
    const float rough = fminf(fmaxf(m->roughness, 0.03f), 1.0f);
    const float alpha = rough * rough;
    const float a2 = alpha * alpha;
    const float big = fmaxf(1.0f, fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z))));
    const float offs = P->ray_offset * big;
    const v3 org = V(p.x + n.x * offs, p.y + n.y * offs, p.z + n.z * offs);

    if (use_nee) {
        const float u_sel = rnd(key, dim0 + DIM_LSEL);
        const float ua = rnd(key, dim0 + DIM_L1);
        const float ub = rnd(key, dim0 + DIM_L2);
        const uint32_t li = cdf_find(lights->cdf, lights->count, u_sel * lights->area);
        const int32_t lt = lights->tri[li];
        const float *q0 = vtx + 3 * (size_t)idx[3 * (size_t)lt + 0];
        const float *q1 = vtx + 3 * (size_t)idx[3 * (size_t)lt + 1];
        const float *q2 = vtx + 3 * (size_t)idx[3 * (size_t)lt + 2];
        const v3 l0 = V(q0[0], q0[1], q0[2]), l1 = V(q1[0], q1[1], q1[2]), l2 = V(q2[0], q2[1], q2[2]);
        const float su = sqrtf(ua);
        const float c0 = 1.0f - su, c1 = ub * su;
        const float c2 = (1.0f - c0) - c1;
        const v3 y = comb3(l0, c0, l1, c1, l2, c2);
        const v3 dv = sub(y, org);
        const float dist2 = dot(dv, dv);
        const v3 lc = cross(sub(l1, l0), sub(l2, l0));
        const float ll2 = dot(lc, lc);
        if (dist2 > 0.0f && ll2 > 0.0f) {
            const float dist = sqrtf(dist2);
            const v3 wi = divs(dv, dist);
            const float n_wi = dot(n, wi);
            const float cos_l = fabsf(dot(lc, wi)) / sqrtf(ll2);
            const float pl = dist2 / (cos_l * lights->area);
            if (n_wi > 0.0f && cos_l > 0.0f && pl > 0.0f) {
                const os_material *lm = &mats[mat_ids[lt]];
                const float qd = (1.0f - p_spec) * (n_wi * OS_INV_PI);
                v3 f = scale(kd, (n_wi * OS_INV_PI) * mis_weight(pl, qd));
                const v3 hv = V(wo.x + wi.x, wo.y + wi.y, wo.z + wi.z);
                const float hl2 = dot(hv, hv);
                if (ws > 0.0f && hl2 > 0.0f) {
                    const v3 h = divs(hv, sqrtf(hl2));
                    const float n_h = dot(n, h), wo_h = dot(wo, h);
                    if (n_h > 0.0f && wo_h > 0.0f) {
                        const float tt = (n_h * n_h) * (a2 - 1.0f) + 1.0f;
                        const float D = (a2 * OS_INV_PI) / (tt * tt);
                        const float om = 1.0f - wo_h;
                        const float om2 = om * om;
                        const float om5 = (om2 * om2) * om;
                        const v3 F = V(ks.x + (1.0f - ks.x) * om5, ks.y + (1.0f - ks.y) * om5,
                                       ks.z + (1.0f - ks.z) * om5);
                        const float g1o = (2.0f * n_wo) / (n_wo + sqrtf(a2 + (1.0f - a2) * (n_wo * n_wo)));
                        const float g1i = (2.0f * n_wi) / (n_wi + sqrtf(a2 + (1.0f - a2) * (n_wi * n_wi)));
                        const float qs = p_spec * ((D * n_h) / (4.0f * wo_h));
                        const float fs = (((D * g1o) * g1i) / (4.0f * n_wo)) * mis_weight(pl, qs);
                        f = V(f.x + F.x * fs, f.y + F.y * fs, f.z + F.z * fs);
                    }
                }
                const v3 le = V(lm->ke[0], lm->ke[1], lm->ke[2]);
                nee->direct = divs(mul(mul(*thr, f), le), pl);
                nee->shadow = max3(nee->direct) > 0.0f;
                nee->so = org;
                nee->sd = wi;
                nee->smax = dist * 0.999f - 2e-4f;
            }
        }
    }

    const float u_lobe = rnd(key, dim0 + DIM_LOBE);
    const float u1 = rnd(key, dim0 + DIM_U1);
    const float u2 = rnd(key, dim0 + DIM_U2);
    float sphi, cphi;
    sincos_2pi(u2, &sphi, &cphi);
    v3 t, b;
    basis(n, &t, &b);

    v3 wi, weight;
    float q;
    if (u_lobe < p_spec) {
        const float cos2 = (1.0f - u1) / (1.0f + (a2 - 1.0f) * u1);
        const float cos_h = sqrtf(cos2);
        const float sin_h = sqrtf(fmaxf(0.0f, 1.0f - cos2));
        const v3 h = comb3(t, sin_h * cphi, b, sin_h * sphi, n, cos_h);
        const float wo_h = dot(wo, h);
        if (!(wo_h > 0.0f)) return 0;
        wi = sub(scale(h, 2.0f * wo_h), wo);
        const float n_wi = dot(n, wi);
        if (!(n_wi > 0.0f)) return 0;
        const float n_h = cos_h;
        const float om = 1.0f - wo_h;
        const float om2 = om * om;
        const float om5 = (om2 * om2) * om;
        const v3 F = V(ks.x + (1.0f - ks.x) * om5, ks.y + (1.0f - ks.y) * om5,
                       ks.z + (1.0f - ks.z) * om5);
        const float g1o = (2.0f * n_wo) / (n_wo + sqrtf(a2 + (1.0f - a2) * (n_wo * n_wo)));
        const float g1i = (2.0f * n_wi) / (n_wi + sqrtf(a2 + (1.0f - a2) * (n_wi * n_wi)));
        const float w = ((g1o * g1i) * wo_h) / ((n_wo * n_h) * p_spec);
        weight = scale(F, w);
        const float tt = (n_h * n_h) * (a2 - 1.0f) + 1.0f;
        const float D = (a2 * OS_INV_PI) / (tt * tt);
        q = p_spec * ((D * n_h) / (4.0f * wo_h));
    } else {
        const float rr = sqrtf(u1);
        const float z = sqrtf(fmaxf(0.0f, 1.0f - u1));
        wi = comb3(t, rr * cphi, b, rr * sphi, n, z);
        weight = divs(kd, 1.0f - p_spec);
        q = (1.0f - p_spec) * (z * OS_INV_PI);
    }
    v3 nthr = mul(*thr, weight);

    if (bounce >= P->rr_start_depth) {
        const float qq = fminf(max3(nthr), 0.95f);
        const float u_rr = rnd(key, dim0 + DIM_RR);
        if (!(u_rr < qq)) return 0;
        nthr = divs(nthr, qq);
    }
    if (!(max3(nthr) > 0.0f)) return 0;

    const float wl = sqrtf(dot(wi, wi));
    if (!(wl > 0.0f)) return 0;
    *no = org;
    *nd = divs(wi, wl);
    *thr = nthr;
    *q_io = q > 0.0f ? q : -1.0f;
    return 1;
}

/* Camera ray at sample position (fx, fy) in pixels.  mode 0 restates shaders/depth.frag:47-83
 * (see o_raygen_reference); mode 1 is the pinhole camera of this project. */
static void camera_ray(const os_camera *cam, uint32_t width, uint32_t height, float fx, float fy,
                       o_ray *out)
{
    if (cam->mode == 0) {
        const float u = fx / (float)width;
        const float v = fy / (float)height;
        float dxyx = u - 0.5f, dxyy = v - 0.5f;
        dxyx *= cam->sx;
        dxyy *= cam->sy;
        const float vz = -1.0f;
        float d[3];
        for (int k = 0; k < 3; ++k)
            d[k] = (cam->c0[k] * dxyx + cam->c1[k] * dxyy) + cam->c2[k] * vz;
        const float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
        const float scale = 1.0f / 20.0f;
        float o[4] = { cam->eye[0], cam->eye[1], cam->eye[2], 1000.0f };
        o[1] += 2.0f;
        for (int k = 0; k < 4; ++k) out->o[k] = o[k] / scale;
        for (int k = 0; k < 3; ++k) out->d[k] = d[k] / len;
        out->d[3] = 0.0f;
    } else {
        const float x = (2.0f * fx) / (float)width - 1.0f;
        const float y = 1.0f - (2.0f * fy) / (float)height;
        const float ax = x * cam->sx, ay = y * cam->sy;
        float d[3];
        for (int k = 0; k < 3; ++k)
            d[k] = (cam->c0[k] * ax + cam->c1[k] * ay) - cam->c2[k];
        const float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
        out->o[0] = cam->eye[0]; out->o[1] = cam->eye[1]; out->o[2] = cam->eye[2]; out->o[3] = 1e30f;
        for (int k = 0; k < 3; ++k) out->d[k] = d[k] / len;
        out->d[3] = 0.0f;
    }
}

/* What rays are traced against.  Single level: nodes / vtx / idx as the reference binds them.
 * Two levels (top != NULL): vtx / idx then hold the world-space copy of every instanced
 * triangle, instance after instance (first_tri[i] = index of instance i's first one), which is
 * what shading and light sampling read; intersection goes through the two-level walk. */
typedef struct {
    const o_bbox *nodes; const float *vtx; const int32_t *idx;
    const o_bbox *top; const o_mesh *meshes; const o_instance *instances; const uint32_t *first_tri;
} os_geom;

static void geom_closest(const os_geom *g, const o_ray *r, o_isect *h)
{
    if (g->top) {
        o_two_level_closest(g->top, g->meshes, g->instances, r, h, NULL);
        if (h->primid >= 0) h->primid += (int32_t)g->first_tri[h->shapeid];
    } else {
        o_intersect_scene_closest(g->nodes, g->vtx, g->idx, r, h, NULL);
    }
}

static int geom_any(const os_geom *g, const o_ray *r)
{
    return g->top ? o_two_level_any(g->top, g->meshes, g->instances, r)
                  : o_intersect_scene_any(g->nodes, g->vtx, g->idx, r);   /* bvh.glslh:378-414 */
}

/* Radiance of one camera sample. */
static v3 trace_path(const os_geom *g,
                     const os_material *mats, const int32_t *mat_ids, const os_camera *cam,
                     const os_params *P, uint32_t x, uint32_t y, uint32_t sample,
                     uint64_t *rays_traced, const os_lights *lights)
{
    const uint32_t key = path_key(P->seed, y * P->width + x, sample);
    float jx = 0.5f, jy = 0.5f;
    if (!(P->flags & (OS_FLAG_PIXEL_CENTRE | OS_FLAG_DEPTH_GREY))) {
        jx = rnd(key, 0);
        jy = rnd(key, 1);
    }
    o_ray ray;
    camera_ray(cam, P->width, P->height, (float)x + jx, (float)y + jy, &ray);
    v3 thr = V(1.0f, 1.0f, 1.0f), L = V(0.0f, 0.0f, 0.0f);
    float q = -1.0f;
    for (uint32_t bounce = 0; bounce < P->max_depth; ++bounce) {
        o_isect h;
        geom_closest(g, &ray, &h);
        ++*rays_traced;
        const float *vtx = g->vtx;
        const int32_t *idx = g->idx;
        if (h.primid < 0) {
            L.x += thr.x * P->background[0];
            L.y += thr.y * P->background[1];
            L.z += thr.z * P->background[2];
            break;
        }
        const int32_t *t = idx + 3 * (int64_t)h.primid;
        const float *p0 = vtx + 3 * (int64_t)t[0], *p1 = vtx + 3 * (int64_t)t[1],
                    *p2 = vtx + 3 * (int64_t)t[2];
        v3 emitted, no, nd;
        os_nee nee;
        const int alive = shade_vertex(&mats[mat_ids[h.primid]], V(p0[0], p0[1], p0[2]),
                                       V(p1[0], p1[1], p1[2]), V(p2[0], p2[1], p2[2]),
                                       h.uvwt[0], h.uvwt[1], h.uvwt[2],
                                       V(ray.d[0], ray.d[1], ray.d[2]), &thr, key, bounce, P,
                                       &emitted, &no, &nd, lights, vtx, idx, mats, mat_ids, &q,
                                       h.uvwt[3], &nee);
        L.x += emitted.x;
        L.y += emitted.y;
        L.z += emitted.z;
        if (nee.shadow) {
            int occluded = 0;
            if (nee.smax > 0.0f) {
                o_ray sr;
                sr.o[0] = nee.so.x; sr.o[1] = nee.so.y; sr.o[2] = nee.so.z; sr.o[3] = nee.smax;
                sr.d[0] = nee.sd.x; sr.d[1] = nee.sd.y; sr.d[2] = nee.sd.z; sr.d[3] = 0.0f;
                occluded = geom_any(g, &sr);
                ++*rays_traced;
            }
            if (!occluded) {
                L.x += nee.direct.x;
                L.y += nee.direct.y;
                L.z += nee.direct.z;
            }
        }
        if (!alive) break;
        ray.o[0] = no.x; ray.o[1] = no.y; ray.o[2] = no.z; ray.o[3] = 1e30f;
        ray.d[0] = nd.x; ray.d[1] = nd.y; ray.d[2] = nd.z; ray.d[3] = 0.0f;
    }
    return L;
}

/* Renders pixels [x0,x1) x [y0,y1) of the frame into film (width*height float4, row-major):
 * rgb += radiance of samples sample_begin .. sample_begin+sample_count-1 in that order,
 * a += sample_count.  With OS_FLAG_DEPTH_GREY the pixel is overwritten with the reference's
 * grey (shaders/depth.frag:90-107).  Returns the number of rays traced. */
static uint64_t render_geom(const os_geom *g, const os_material *mats, const int32_t *mat_ids,
                            const os_camera *cam, const os_params *P, uint32_t x0, uint32_t y0,
                            uint32_t x1, uint32_t y1, float *film, int nthreads, uint32_t n_tri)
{
    uint64_t rays = 0;
    if (nthreads < 1) nthreads = 1;
    os_lights lights = { NULL, NULL, 0, 0.0f };
    if (P->flags & OS_FLAG_NEE) build_lights(g->vtx, g->idx, n_tri, mats, mat_ids, &lights);
    const os_lights *LP = &lights;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+ : rays)
    for (int64_t y = y0; y < (int64_t)y1; ++y) {
        for (uint32_t x = x0; x < x1; ++x) {
            float *f = film + 4 * ((uint64_t)y * P->width + x);
            if (P->flags & OS_FLAG_DEPTH_GREY) {
                o_ray ray;
                camera_ray(cam, P->width, P->height, (float)x + 0.5f, (float)y + 0.5f, &ray);
                o_isect h;
                geom_closest(g, &ray, &h);
                ++rays;
                o_depth_shade(&h, f);
                continue;
            }
            for (uint32_t s = 0; s < P->sample_count; ++s) {
                const v3 L = trace_path(g, mats, mat_ids, cam, P, x, (uint32_t)y,
                                        P->sample_begin + s, &rays, LP);
                f[0] += L.x;
                f[1] += L.y;
                f[2] += L.z;
                f[3] += 1.0f;
            }
        }
    }
    free(lights.tri);
    free(lights.cdf);
    return rays;
}

uint64_t os_render(const o_bbox *nodes, const float *vtx, const int32_t *idx,
                   const os_material *mats, const int32_t *mat_ids, const os_camera *cam,
                   const os_params *P, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                   float *film, int nthreads, uint32_t n_tri)
{
    const os_geom g = { nodes, vtx, idx, NULL, NULL, NULL, NULL };
    return render_geom(&g, mats, mat_ids, cam, P, x0, y0, x1, y1, film, nthreads, n_tri);
}

/* The same renderer over a two-level scene.  world_vtx / world_idx / mat_ids describe the
 * n_tri instanced triangles in world space, instance after instance (see os_geom). */
uint64_t os_render_two_level(const o_bbox *top, const o_mesh *meshes, const o_instance *instances,
                             const uint32_t *first_tri, const float *world_vtx,
                             const int32_t *world_idx, const os_material *mats,
                             const int32_t *mat_ids, const os_camera *cam, const os_params *P,
                             uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, float *film,
                             int nthreads, uint32_t n_tri)
{
    const os_geom g = { NULL, world_vtx, world_idx, top, meshes, instances, first_tri };
    return render_geom(&g, mats, mat_ids, cam, P, x0, y0, x1, y1, film, nthreads, n_tri);
}

/* Exposed pieces for unit tests of the sampling routines. */
void os_sincos_2pi(float u, float *s, float *c) { sincos_2pi(u, s, c); }
float os_rnd(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t dim)
{
    return rnd(path_key(seed, pixel, sample), dim);
}
