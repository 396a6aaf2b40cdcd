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
#define OS_FLAG_PIXEL_CENTRE 4u

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

enum { DIM_LOBE = 0, DIM_U1 = 1, DIM_U2 = 2, DIM_RR = 3, DIMS_PER_BOUNCE = 6 };

/* One path vertex; returns 1 if the path continues (then *thr, *no, *nd are updated). */
static int shade_vertex(const os_material *m, v3 v0, v3 v1, v3 v2, float b0, float b1, float b2,
                        v3 rd, v3 *thr, uint32_t key, uint32_t bounce, const os_params *P,
                        v3 *emitted, v3 *no, v3 *nd)
{
    const v3 ke = V(m->ke[0], m->ke[1], m->ke[2]);
    *emitted = mul(*thr, ke);
    if (bounce + 1 >= P->max_depth) return 0;

    const v3 p = comb3(v0, b0, v1, b1, v2, b2);
    v3 ng = cross(sub(v1, v0), sub(v2, v0));
    const float len2 = dot(ng, ng);
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
    const float u_lobe = rnd(key, dim0 + DIM_LOBE);
    const float u1 = rnd(key, dim0 + DIM_U1);
    const float u2 = rnd(key, dim0 + DIM_U2);
    float sphi, cphi;
    sincos_2pi(u2, &sphi, &cphi);
    v3 t, b;
    basis(n, &t, &b);

    v3 wi, weight;
    if (u_lobe < p_spec) {
        const float rough = fminf(fmaxf(m->roughness, 0.03f), 1.0f);
        const float alpha = rough * rough;
        const float a2 = alpha * alpha;
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
    } else {
        const float rr = sqrtf(u1);
        const float z = sqrtf(fmaxf(0.0f, 1.0f - u1));
        wi = comb3(t, rr * cphi, b, rr * sphi, n, z);
        weight = divs(kd, 1.0f - p_spec);
    }
    v3 nthr = mul(*thr, weight);

    if (bounce >= P->rr_start_depth) {
        const float q = fminf(max3(nthr), 0.95f);
        const float u_rr = rnd(key, dim0 + DIM_RR);
        if (!(u_rr < q)) return 0;
        nthr = divs(nthr, q);
    }
    if (!(max3(nthr) > 0.0f)) return 0;

    const float wl = sqrtf(dot(wi, wi));
    if (!(wl > 0.0f)) return 0;
    const float big = fmaxf(1.0f, fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z))));
    const float offs = P->ray_offset * big;
    *no = V(p.x + n.x * offs, p.y + n.y * offs, p.z + n.z * offs);
    *nd = divs(wi, wl);
    *thr = nthr;
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

/* Radiance of one camera sample. */
static v3 trace_path(const o_bbox *nodes, const float *vtx, const int32_t *idx,
                     const os_material *mats, const int32_t *mat_ids, const os_camera *cam,
                     const os_params *P, uint32_t x, uint32_t y, uint32_t sample,
                     uint64_t *rays_traced)
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
    for (uint32_t bounce = 0; bounce < P->max_depth; ++bounce) {
        o_isect h;
        o_intersect_scene_closest(nodes, vtx, idx, &ray, &h, NULL);
        ++*rays_traced;
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
        const int alive = shade_vertex(&mats[mat_ids[h.primid]], V(p0[0], p0[1], p0[2]),
                                       V(p1[0], p1[1], p1[2]), V(p2[0], p2[1], p2[2]),
                                       h.uvwt[0], h.uvwt[1], h.uvwt[2],
                                       V(ray.d[0], ray.d[1], ray.d[2]), &thr, key, bounce, P,
                                       &emitted, &no, &nd);
        L.x += emitted.x;
        L.y += emitted.y;
        L.z += emitted.z;
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
uint64_t os_render(const o_bbox *nodes, const float *vtx, const int32_t *idx,
                   const os_material *mats, const int32_t *mat_ids, const os_camera *cam,
                   const os_params *P, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                   float *film, int nthreads)
{
    uint64_t rays = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+ : rays)
    for (int64_t y = y0; y < (int64_t)y1; ++y) {
        for (uint32_t x = x0; x < x1; ++x) {
            float *f = film + 4 * ((uint64_t)y * P->width + x);
            if (P->flags & OS_FLAG_DEPTH_GREY) {
                o_ray ray;
                camera_ray(cam, P->width, P->height, (float)x + 0.5f, (float)y + 0.5f, &ray);
                o_isect h;
                o_intersect_scene_closest(nodes, vtx, idx, &ray, &h, NULL);
                ++rays;
                o_depth_shade(&h, f);
                continue;
            }
            for (uint32_t s = 0; s < P->sample_count; ++s) {
                const v3 L = trace_path(nodes, vtx, idx, mats, mat_ids, cam, P, x, (uint32_t)y,
                                        P->sample_begin + s, &rays);
                f[0] += L.x;
                f[1] += L.y;
                f[2] += L.z;
                f[3] += 1.0f;
            }
        }
    }
    return rays;
}

/* Exposed pieces for unit tests of the sampling routines. */
void os_sincos_2pi(float u, float *s, float *c) { sincos_2pi(u, s, c); }
float os_rnd(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t dim)
{
    return rnd(path_key(seed, pixel, sample), dim);
}
