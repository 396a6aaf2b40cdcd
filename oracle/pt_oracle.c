/*
 * pt_oracle.c -- CPU restatement of the reference's ray/scene intersection,
 * raygen, debug shade and skip-link linearizer.  See pt_oracle.h for the
 * status of this file (test infrastructure; parity pinned only by the
 * literals the reference text holds) and for the floating-point contract.
 */
#include "pt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------------- */
/* shaders/RadeonRays/bvh.glslh:51-63                                      */
void o_precompute_ray(const o_ray *r, o_ray_internal *out)
{
    out->ray = *r;
    float dx = r->d[0], dy = r->d[1], dz = r->d[2];
    float px = dx, py = dy, pz = dz;
    if (fabsf(dx) >= fabsf(dy) && fabsf(dx) >= fabsf(dz)) {
        px = dy; py = dz; pz = dx;                 /* d.yzx */
    } else if (fabsf(dy) >= fabsf(dz)) {
        px = dz; py = dx; pz = dy;                 /* d.zxy */
    }
    out->s[0] = -px / pz;
    out->s[1] = -py / pz;
    out->s[2] = 1.0f / pz;
    out->d[0] = px; out->d[1] = py; out->d[2] = pz;
}

/* shaders/RadeonRays/bvh.glslh:165-177 */
int o_intersect_box(const o_ray *r, const float invdir[3], const o_bbox *box, float maxt)
{
    float f[3], n[3];
    for (int k = 0; k < 3; ++k) {
        f[k] = (box->pmax[k] - r->o[k]) * invdir[k];
        n[k] = (box->pmin[k] - r->o[k]) * invdir[k];
    }
    float tmaxv[3], tminv[3];
    for (int k = 0; k < 3; ++k) {
        tmaxv[k] = fmaxf(f[k], n[k]);
        tminv[k] = fminf(f[k], n[k]);
    }
    const float t1 = fminf(fminf(tmaxv[0], fminf(tmaxv[1], tmaxv[2])), maxt);
    const float t0 = fmaxf(fmaxf(tminv[0], fmaxf(tminv[1], tminv[2])), 0.0f);
    return t1 >= t0;
}

static inline void cross3(const float a[3], const float b[3], float o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static inline float dot3(const float a[3], const float b[3])
{
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}

#define O_EPS 1e-4f /* shaders/RadeonRays/bvh.glslh:179 */

/* shaders/RadeonRays/bvh.glslh:181-202 (Moller-Trumbore, closest form) */
int o_intersect_triangle_mt(const o_ray *r, const float v1[3], const float v2[3],
                            const float v3[3], o_isect *isect)
{
    float e1[3], e2[3], d[3], s1[3], s2[3];
    for (int k = 0; k < 3; ++k) { e1[k] = v2[k] - v1[k]; e2[k] = v3[k] - v1[k]; }
    cross3(r->d, e2, s1);
    const float invd = 1.0f / dot3(s1, e1);
    for (int k = 0; k < 3; ++k) d[k] = r->o[k] - v1[k];
    const float b1 = dot3(d, s1) * invd;
    cross3(d, e1, s2);
    const float b2 = dot3(r->d, s2) * invd;
    /* max(0.0, x) as GLSL defines it (y if x < y, else x): +0 for x = -0, 0 for NaN.  fmaxf
     * leaves the sign of a zero result to the implementation. */
    const float traw = dot3(e2, s2) * invd;
    const float temp = (0.0f < traw) ? traw : 0.0f;
    if (b1 < -O_EPS || b1 > 1.0f + O_EPS || b2 < -O_EPS || b1 + b2 > 1.0f + O_EPS ||
        temp < -O_EPS || temp > isect->uvwt[3] + O_EPS)
        return 0;
    isect->uvwt[0] = b1; isect->uvwt[1] = b2; isect->uvwt[2] = 0.0f; isect->uvwt[3] = temp;
    return 1;
}

/* shaders/RadeonRays/bvh.glslh:289-310 (Moller-Trumbore, occlusion form) */
int o_intersect_triangle_p(const o_ray *r, const float v1[3], const float v2[3],
                           const float v3[3])
{
    float e1[3], e2[3], d[3], s1[3], s2[3];
    for (int k = 0; k < 3; ++k) { e1[k] = v2[k] - v1[k]; e2[k] = v3[k] - v1[k]; }
    cross3(r->d, e2, s1);
    const float invd = 1.0f / dot3(s1, e1);
    for (int k = 0; k < 3; ++k) d[k] = r->o[k] - v1[k];
    const float b1 = dot3(d, s1) * invd;
    cross3(d, e1, s2);
    const float b2 = dot3(r->d, s2) * invd;
    const float temp = dot3(e2, s2) * invd;
    if (b1 < -O_EPS || b1 > 1.0f + O_EPS || b2 < -O_EPS || b1 + b2 > 1.0f + O_EPS ||
        temp < -O_EPS || temp > r->o[3] + O_EPS)
        return 0;
    return 1;
}

/* shaders/RadeonRays/bvh.glslh:206-287 */
int o_intersect_triangle_watertight(const o_ray_internal *r, const float v0[3],
                                    const float v1[3], const float v2[3], o_isect *isect)
{
    const float *o = r->ray.o;
    float p0[3], p1[3], p2[3];
    for (int k = 0; k < 3; ++k) { p0[k] = v0[k] - o[k]; p1[k] = v1[k] - o[k]; p2[k] = v2[k] - o[k]; }

    const float dx = r->ray.d[0], dy = r->ray.d[1], dz = r->ray.d[2];
    if (fabsf(dx) >= fabsf(dy) && fabsf(dx) >= fabsf(dz)) {           /* .yzx */
        float t;
        t = p0[0]; p0[0] = p0[1]; p0[1] = p0[2]; p0[2] = t;
        t = p1[0]; p1[0] = p1[1]; p1[1] = p1[2]; p1[2] = t;
        t = p2[0]; p2[0] = p2[1]; p2[1] = p2[2]; p2[2] = t;
    } else if (fabsf(dy) >= fabsf(dz)) {                              /* .zxy */
        float t;
        t = p0[2]; p0[2] = p0[1]; p0[1] = p0[0]; p0[0] = t;
        t = p1[2]; p1[2] = p1[1]; p1[1] = p1[0]; p1[0] = t;
        t = p2[2]; p2[2] = p2[1]; p2[1] = p2[0]; p2[0] = t;
    }

    const float sx = r->s[0], sy = r->s[1], sz = r->s[2];
    p0[0] += sx * p0[2]; p0[1] += sy * p0[2];
    p1[0] += sx * p1[2]; p1[1] += sy * p1[2];
    p2[0] += sx * p2[2]; p2[1] += sy * p2[2];

    float e0 = p1[0] * p2[1] - p1[1] * p2[0];
    float e1 = p2[0] * p0[1] - p2[1] * p0[0];
    float e2 = p0[0] * p1[1] - p0[1] * p1[0];

    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        const double p2xp1y = (double)p2[0] * (double)p1[1];
        const double p2yp1x = (double)p2[1] * (double)p1[0];
        e0 = (float)(p2yp1x - p2xp1y);
        const double p0xp2y = (double)p0[0] * (double)p2[1];
        const double p0yp2x = (double)p0[1] * (double)p2[0];
        e1 = (float)(p0yp2x - p0xp2y);
        const double p1xp0y = (double)p1[0] * (double)p0[1];
        const double p1yp0x = (double)p1[1] * (double)p0[0];
        e2 = (float)(p1yp0x - p1xp0y);
    }

    if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f))
        return 0;
    const float det = (e0 + e1) + e2;
    if (det == 0.0f)
        return 0;

    p0[2] *= sz; p1[2] *= sz; p2[2] *= sz;
    const float tScaled = (e0 * p0[2] + e1 * p1[2]) + e2 * p2[2];
    const float tMax = isect->uvwt[3];
    if (det < 0.0f && (tScaled >= 0.0f || tScaled < tMax * det))
        return 0;
    else if (det > 0.0f && (tScaled <= 0.0f || tScaled > tMax * det))
        return 0;

    const float invDet = 1.0f / det;
    isect->uvwt[0] = e0 * invDet;
    isect->uvwt[1] = e1 * invDet;
    isect->uvwt[2] = e2 * invDet;
    isect->uvwt[3] = tScaled * invDet;
    return 1;
}

/* shaders/RadeonRays/bvh.glslh:312-319 */
static inline void get_vertex(const float *vertices, int32_t index, float v[3])
{
    v[0] = vertices[3 * (int64_t)index + 0];
    v[1] = vertices[3 * (int64_t)index + 1];
    v[2] = vertices[3 * (int64_t)index + 2];
}

/* STARTIDX / LEAFNODE, shaders/RadeonRays/bvh.glslh:153-154 */
static inline int32_t node_startidx(const o_bbox *n) { return (int32_t)n->pmin[3]; }
static inline int node_is_leaf(const o_bbox *n) { return n->pmin[3] != -1.0f; }

/* shaders/RadeonRays/bvh.glslh:321-348 */
static inline void intersect_leaf_closest(const o_bbox *node, const float *vertices,
                                          const int32_t *indices, const o_ray_internal *ri,
                                          o_isect *isect)
{
    const int64_t start = node_startidx(node);
    float v1[3], v2[3], v3[3];
    get_vertex(vertices, indices[3 * start + 0], v1);
    get_vertex(vertices, indices[3 * start + 1], v2);
    get_vertex(vertices, indices[3 * start + 2], v3);
    if (o_intersect_triangle_watertight(ri, v1, v2, v3, isect)) {
        isect->primid = (int32_t)node->pmin[3];
        isect->shapeid = 0;
    }
}

/* shaders/RadeonRays/bvh.glslh:417-456 */
void o_intersect_scene_closest(const o_bbox *nodes, const float *vertices, const int32_t *indices,
                               const o_ray *r, o_isect *isect, o_counters *cnt)
{
    o_ray_internal ri;
    o_precompute_ray(r, &ri);
    const float invdir[3] = { 1.0f / r->d[0], 1.0f / r->d[1], 1.0f / r->d[2] };

    isect->uvwt[0] = 0.0f; isect->uvwt[1] = 0.0f; isect->uvwt[2] = 0.0f; isect->uvwt[3] = r->o[3];
    isect->shapeid = -1;
    isect->primid = -1;
    int32_t idx = 0;
    isect->pad0 = 667;
    isect->pad1 = 0;

    uint64_t nb = 0, nt = 0;
    while (idx != -1) {
        const o_bbox *node = &nodes[idx];
        ++nb;
        if (o_intersect_box(&ri.ray, invdir, node, isect->uvwt[3])) {
            if (node_is_leaf(node)) {
                ++nt;
                intersect_leaf_closest(node, vertices, indices, &ri, isect);
                idx = (int32_t)node->pmax[3];
            } else {
                ++idx;
            }
        } else {
            idx = (int32_t)node->pmax[3];
        }
    }
    if (cnt) { cnt->box_tests += nb; cnt->tri_tests += nt; }
}

/* shaders/RadeonRays/bvh.glslh:378-414 (leaf test :350-375) */
int o_intersect_scene_any(const o_bbox *nodes, const float *vertices, const int32_t *indices,
                          const o_ray *r)
{
    const float invdir[3] = { 1.0f / r->d[0], 1.0f / r->d[1], 1.0f / r->d[2] };
    int32_t idx = 0;
    while (idx != -1) {
        const o_bbox *node = &nodes[idx];
        if (o_intersect_box(r, invdir, node, r->o[3])) {
            if (node_is_leaf(node)) {
                const int64_t start = node_startidx(node);
                float v1[3], v2[3], v3[3];
                get_vertex(vertices, indices[3 * start + 0], v1);
                get_vertex(vertices, indices[3 * start + 1], v2);
                get_vertex(vertices, indices[3 * start + 2], v3);
                if (o_intersect_triangle_p(r, v1, v2, v3))
                    return 1;
                idx = (int32_t)node->pmax[3];
            } else {
                ++idx;
            }
        } else {
            idx = (int32_t)node->pmax[3];
        }
    }
    return 0;
}

/* ---- two-level scenes (specification of this project; see pt_oracle.h) ---------------- */
int o_invert_affine(const float m[12], float inv[12])
{
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9],
                 i = m[10];
    const double tx = m[3], ty = m[7], tz = m[11];
    const double A = e * i - f * h, B = f * g - d * i, C = d * h - e * g;
    const double det = (a * A + b * B) + c * C;
    if (!(fabs(det) > 0.0) || !isfinite(det)) return 0;
    const double r[3][3] = {
        { A / det, (c * h - b * i) / det, (b * f - c * e) / det },
        { B / det, (a * i - c * g) / det, (c * d - a * f) / det },
        { C / det, (b * g - a * h) / det, (a * e - b * d) / det },
    };
    for (int k = 0; k < 3; ++k) {
        inv[4 * k + 0] = (float)r[k][0];
        inv[4 * k + 1] = (float)r[k][1];
        inv[4 * k + 2] = (float)r[k][2];
        inv[4 * k + 3] = (float)(-((r[k][0] * tx + r[k][1] * ty) + r[k][2] * tz));
    }
    for (int k = 0; k < 12; ++k)
        if (!isfinite(inv[k])) return 0;
    return 1;
}

static void transform_point(const float m[12], const float p[3], float out[3])
{
    for (int k = 0; k < 3; ++k)
        out[k] = ((m[4 * k + 0] * p[0] + m[4 * k + 1] * p[1]) + m[4 * k + 2] * p[2]) + m[4 * k + 3];
}

void o_transform_points(const float m[12], const float *p, uint64_t n, float *out)
{
    for (uint64_t i = 0; i < n; ++i) {
        float w[3];
        transform_point(m, p + 3 * i, w);
        out[3 * i + 0] = w[0]; out[3 * i + 1] = w[1]; out[3 * i + 2] = w[2];
    }
}

void o_instance_bounds(const float m[12], const o_bbox *root, o_bbox *out)
{
    for (int k = 0; k < 3; ++k) { out->pmin[k] = INFINITY; out->pmax[k] = -INFINITY; }
    out->pmin[3] = out->pmax[3] = 0.0f;
    for (int corner = 0; corner < 8; ++corner) {
        const float p[3] = { (corner & 1) ? root->pmax[0] : root->pmin[0],
                             (corner & 2) ? root->pmax[1] : root->pmin[1],
                             (corner & 4) ? root->pmax[2] : root->pmin[2] };
        float w[3];
        transform_point(m, p, w);
        for (int k = 0; k < 3; ++k) {
            out->pmin[k] = fminf(out->pmin[k], w[k]);
            out->pmax[k] = fmaxf(out->pmax[k], w[k]);
        }
    }
}

static void to_object_space(const o_instance *in, const o_ray *r, o_ray *out)
{
    const float *m = in->inv;
    for (int k = 0; k < 3; ++k) {
        out->o[k] = ((m[4 * k + 0] * r->o[0] + m[4 * k + 1] * r->o[1]) + m[4 * k + 2] * r->o[2]) + m[4 * k + 3];
        out->d[k] = (m[4 * k + 0] * r->d[0] + m[4 * k + 1] * r->d[1]) + m[4 * k + 2] * r->d[2];
    }
    out->o[3] = r->o[3];
    out->d[3] = r->d[3];
}

void o_two_level_closest(const o_bbox *top, const o_mesh *meshes, const o_instance *instances,
                         const o_ray *r, o_isect *isect, o_counters *cnt)
{
    const float invdir[3] = { 1.0f / r->d[0], 1.0f / r->d[1], 1.0f / r->d[2] };
    isect->uvwt[0] = 0.0f; isect->uvwt[1] = 0.0f; isect->uvwt[2] = 0.0f; isect->uvwt[3] = r->o[3];
    isect->shapeid = -1;
    isect->primid = -1;
    isect->pad0 = 667;
    isect->pad1 = 0;
    uint64_t nb = 0;
    int32_t idx = 0;
    while (idx != -1) {                                     /* the loop of bvh.glslh:431-455 */
        const o_bbox *node = &top[idx];
        ++nb;
        if (o_intersect_box(r, invdir, node, isect->uvwt[3])) {
            if (node_is_leaf(node)) {
                const int32_t inst = (int32_t)node_startidx(node);
                const o_instance *in = &instances[inst];
                const o_mesh *mesh = &meshes[in->mesh];
                o_ray lr;
                to_object_space(in, r, &lr);
                lr.o[3] = isect->uvwt[3];                   /* running nearest distance */
                o_isect h;
                o_intersect_scene_closest(mesh->nodes, mesh->vertices, mesh->indices, &lr, &h, cnt);
                if (h.primid >= 0) {
                    *isect = h;
                    isect->shapeid = inst;
                }
                idx = (int32_t)node->pmax[3];
            } else {
                ++idx;
            }
        } else {
            idx = (int32_t)node->pmax[3];
        }
    }
    if (cnt) cnt->box_tests += nb;
}

int o_two_level_any(const o_bbox *top, const o_mesh *meshes, const o_instance *instances,
                    const o_ray *r)
{
    const float invdir[3] = { 1.0f / r->d[0], 1.0f / r->d[1], 1.0f / r->d[2] };
    int32_t idx = 0;
    while (idx != -1) {                                     /* the loop of bvh.glslh:383-411 */
        const o_bbox *node = &top[idx];
        if (o_intersect_box(r, invdir, node, r->o[3])) {
            if (node_is_leaf(node)) {
                const o_instance *in = &instances[(int32_t)node_startidx(node)];
                const o_mesh *mesh = &meshes[in->mesh];
                o_ray lr;
                to_object_space(in, r, &lr);
                if (o_intersect_scene_any(mesh->nodes, mesh->vertices, mesh->indices, &lr))
                    return 1;
                idx = (int32_t)node->pmax[3];
            } else {
                ++idx;
            }
        } else {
            idx = (int32_t)node->pmax[3];
        }
    }
    return 0;
}

void o_two_level_closest_batch(const o_bbox *top, const o_mesh *meshes,
                               const o_instance *instances, const o_ray *rays, uint64_t n,
                               o_isect *out, o_counters *cnt, int nthreads)
{
    uint64_t nb = 0, nt = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads) reduction(+ : nb, nt)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        o_counters c = { 0, 0 };
        o_two_level_closest(top, meshes, instances, &rays[i], &out[i], &c);
        nb += c.box_tests; nt += c.tri_tests;
    }
    if (cnt) { cnt->box_tests += nb; cnt->tri_tests += nt; }
}

void o_two_level_any_batch(const o_bbox *top, const o_mesh *meshes, const o_instance *instances,
                           const o_ray *rays, uint64_t n, int32_t *out, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
    for (int64_t i = 0; i < (int64_t)n; ++i)
        out[i] = o_two_level_any(top, meshes, instances, &rays[i]);
}

void o_closest_batch(const o_bbox *nodes, const float *vertices, const int32_t *indices,
                     const o_ray *rays, uint64_t n, o_isect *out, o_counters *cnt, int nthreads)
{
    uint64_t nb = 0, nt = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads) reduction(+ : nb, nt)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        o_counters c = { 0, 0 };
        o_intersect_scene_closest(nodes, vertices, indices, &rays[i], &out[i], &c);
        nb += c.box_tests; nt += c.tri_tests;
    }
    if (cnt) { cnt->box_tests += nb; cnt->tri_tests += nt; }
}

void o_any_batch(const o_bbox *nodes, const float *vertices, const int32_t *indices,
                 const o_ray *rays, uint64_t n, int32_t *out, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
    for (int64_t i = 0; i < (int64_t)n; ++i)
        out[i] = o_intersect_scene_any(nodes, vertices, indices, &rays[i]);
}

void o_brute_closest_batch(const float *vertices, const int32_t *indices, uint32_t n_tri,
                           const o_ray *rays, uint64_t n, o_isect *out, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        o_ray_internal ri;
        o_precompute_ray(&rays[i], &ri);
        o_isect is;
        is.uvwt[0] = is.uvwt[1] = is.uvwt[2] = 0.0f; is.uvwt[3] = rays[i].o[3];
        is.shapeid = -1; is.primid = -1; is.pad0 = 667; is.pad1 = 0;
        for (uint32_t t = 0; t < n_tri; ++t) {
            float v1[3], v2[3], v3[3];
            get_vertex(vertices, indices[3 * (int64_t)t + 0], v1);
            get_vertex(vertices, indices[3 * (int64_t)t + 1], v2);
            get_vertex(vertices, indices[3 * (int64_t)t + 2], v3);
            if (o_intersect_triangle_watertight(&ri, v1, v2, v3, &is)) {
                is.primid = (int32_t)t;
                is.shapeid = 0;
            }
        }
        out[i] = is;
    }
}

/* ---------------------------------------------------------------------- */
/* shaders/depth.frag:47-83; pixel -> texCoords from shaders/quad.vert:6,14
 * (texCoords = 0.5*v+0.5, sampled at the pixel centre; row 0 is the top row
 * of the framebuffer). */
void o_raygen_reference(const o_camera *cam, uint32_t width, uint32_t height,
                        uint32_t px, uint32_t py, o_ray *out)
{
    const float u = ((float)px + 0.5f) / (float)width;
    const float v = ((float)py + 0.5f) / (float)height;
    float o[4] = { 0.0f, 0.0f, 0.0f, 1000.0f };             /* :52-53 */
    float dxyx = u - 0.5f, dxyy = v - 0.5f;                  /* :55 */
    dxyx *= cam->ip00;                                       /* :59 */
    dxyy *= cam->ip11;                                       /* :60 */
    const float dv[3] = { dxyx, dxyy, -1.0f };               /* :65 */
    o[0] = cam->camera_position[0];                          /* :70 */
    o[1] = cam->camera_position[1];
    o[2] = cam->camera_position[2];
    float d[3];
    for (int k = 0; k < 3; ++k)                              /* :74, column-major mat3 * vec3 */
        d[k] = (cam->inv_view_c0[k] * dv[0] + cam->inv_view_c1[k] * dv[1]) +
               cam->inv_view_c2[k] * dv[2];
    const float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);   /* :75 */
    d[0] /= len; d[1] /= len; d[2] /= len;
    const float scale = 1.0f / 20.0f;                        /* :81 */
    o[1] += 2.0f;                                            /* :82 */
    for (int k = 0; k < 4; ++k) o[k] /= scale;               /* :83, a vec4 divide: tmax too */
    for (int k = 0; k < 4; ++k) out->o[k] = o[k];
    out->d[0] = d[0]; out->d[1] = d[1]; out->d[2] = d[2]; out->d[3] = 0.0f;   /* :66 */
}

void o_raygen_reference_image(const o_camera *cam, uint32_t width, uint32_t height, o_ray *out)
{
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x)
            o_raygen_reference(cam, width, height, x, y, &out[(uint64_t)y * width + x]);
}

static inline float o_clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
/* GLSL smoothstep(edge0, edge1, x) */
static inline float o_smoothstep(float e0, float e1, float x)
{
    const float t = o_clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return (t * t) * (3.0f - 2.0f * t);
}

/* shaders/depth.frag:90-107.  The primid colour hash at :99-104 is computed
 * and then overwritten at :105, so only the grey value survives. */
void o_depth_shade(const o_isect *isect, float rgba[4])
{
    if (isect->shapeid == -1) {
        rgba[0] = 0.0f; rgba[1] = 0.0f; rgba[2] = 0.0f; rgba[3] = 1.0f;
        return;
    }
    const float scale = 1.0f / 20.0f;
    float x = isect->uvwt[3];
    x *= scale;
    x = o_smoothstep(x, 0.0f, 3.0f);          /* argument order as written at :96 */
    rgba[0] = x; rgba[1] = x; rgba[2] = x; rgba[3] = 1.0f;
}

/* ---------------------------------------------------------------------- */
/* src/main_granite.cpp:162-172: one box per triangle, grown over its three
 * vertices from an empty box (+FLT_MAX / -FLT_MAX). */
void o_mesh_leaf_bounds(const float *vertices, const int32_t *indices, uint32_t n_tri, o_bbox *out)
{
    for (uint32_t t = 0; t < n_tri; ++t) {
        o_bbox b;
        for (int k = 0; k < 3; ++k) { b.pmin[k] = FLT_MAX; b.pmax[k] = -FLT_MAX; }
        b.pmin[3] = 0.0f; b.pmax[3] = 0.0f;
        for (int j = 0; j < 3; ++j) {
            float p[3];
            get_vertex(vertices, indices[3 * (int64_t)t + j], p);
            for (int k = 0; k < 3; ++k) {
                b.pmin[k] = fminf(b.pmin[k], p[k]);
                b.pmax[k] = fmaxf(b.pmax[k], p[k]);
            }
        }
        out[t] = b;
    }
}

/* src/bvh/bvh_builder.cpp:53-59: scale each box by 1+1e-4 about its centre
 * (centre = 0.5*(pmax+pmin)). */
void o_inflate_leaf_bounds(o_bbox *boxes, uint32_t n)
{
    const float kBoundsScale = 1.0f + 1e-4f;
    for (uint32_t i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) {
            const float c = 0.5f * (boxes[i].pmax[k] + boxes[i].pmin[k]);
            const float lo = c + (boxes[i].pmin[k] - c) * kBoundsScale;
            const float hi = c + (boxes[i].pmax[k] - c) * kBoundsScale;
            boxes[i].pmin[k] = lo;
            boxes[i].pmax[k] = hi;
        }
    }
}

/* ---------------------------------------------------------------------- */
/* Tree builder.  The reference delegates this to RadeonRays::Bvh::Build
 * (src/bvh/bvh_builder.cpp:61-62), whose sources are not in the checkout.
 * What follows is the textbook binned-SAH / spatial-median recursion with the
 * properties the linearizer and the shader rely on: binary, one primitive per
 * leaf, node bounds = union of the (inflated) leaf boxes below it, leaves
 * appended to the reordering array as they are emitted. */
typedef struct {
    const o_bbox *leaf;
    float (*centroid)[3];
    int32_t *prims;
    o_tree *t;
    int use_sah, num_bins;
    float traversal_cost;
} build_ctx;

static inline void box_empty(o_bbox *b)
{
    for (int k = 0; k < 3; ++k) { b->pmin[k] = FLT_MAX; b->pmax[k] = -FLT_MAX; }
    b->pmin[3] = 0.0f; b->pmax[3] = 0.0f;
}
static inline void box_grow_box(o_bbox *b, const o_bbox *o)
{
    for (int k = 0; k < 3; ++k) {
        b->pmin[k] = fminf(b->pmin[k], o->pmin[k]);
        b->pmax[k] = fmaxf(b->pmax[k], o->pmax[k]);
    }
}
static inline float box_area(const o_bbox *b)
{
    const float ex = b->pmax[0] - b->pmin[0], ey = b->pmax[1] - b->pmin[1],
                ez = b->pmax[2] - b->pmin[2];
    return 2.0f * ((ex * ey + ex * ez) + ey * ez);
}

static o_tree_node *build_rec(build_ctx *c, int32_t begin, int32_t end)
{
    o_tree_node *node = &c->t->pool[c->t->nodecnt++];
    box_empty(&node->bounds);
    float cmin[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, cmax[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int32_t i = begin; i < end; ++i) {
        const int32_t p = c->prims[i];
        box_grow_box(&node->bounds, &c->leaf[p]);
        for (int k = 0; k < 3; ++k) {
            cmin[k] = fminf(cmin[k], c->centroid[p][k]);
            cmax[k] = fmaxf(cmax[k], c->centroid[p][k]);
        }
    }
    const int32_t n = end - begin;
    if (n < 2) {
        node->is_leaf = 1; node->lc = node->rc = NULL;
        node->startidx = c->t->n_indices; node->numprims = n;
        for (int32_t i = begin; i < end; ++i) c->t->indices[c->t->n_indices++] = c->prims[i];
        return node;
    }
    int axis = 0;
    {
        const float e[3] = { cmax[0] - cmin[0], cmax[1] - cmin[1], cmax[2] - cmin[2] };
        if (e[1] > e[axis]) axis = 1;
        if (e[2] > e[axis]) axis = 2;
    }
    float border = 0.5f * (cmax[axis] + cmin[axis]);

    if (c->use_sah) {
        const int nb = c->num_bins;
        float best = FLT_MAX;
        const float inv_area = 1.0f / box_area(&node->bounds);
        o_bbox *bb = (o_bbox *)malloc(sizeof(o_bbox) * (size_t)nb);
        int32_t *bc = (int32_t *)malloc(sizeof(int32_t) * (size_t)nb);
        o_bbox *rb = (o_bbox *)malloc(sizeof(o_bbox) * (size_t)nb);
        for (int a = 0; a < 3; ++a) {
            const float ext = cmax[a] - cmin[a];
            if (!(ext > 0.0f)) continue;
            for (int b = 0; b < nb; ++b) { box_empty(&bb[b]); bc[b] = 0; }
            const float k1 = (float)nb / ext;
            for (int32_t i = begin; i < end; ++i) {
                const int32_t p = c->prims[i];
                int b = (int)(k1 * (c->centroid[p][a] - cmin[a]));
                if (b >= nb) b = nb - 1;
                if (b < 0) b = 0;
                bc[b]++; box_grow_box(&bb[b], &c->leaf[p]);
            }
            o_bbox acc; box_empty(&acc);
            for (int b = nb - 1; b > 0; --b) { box_grow_box(&acc, &bb[b]); rb[b] = acc; }
            box_empty(&acc);
            int32_t lc = 0;
            for (int b = 0; b < nb - 1; ++b) {
                box_grow_box(&acc, &bb[b]); lc += bc[b];
                const int32_t rc = n - lc;
                if (lc == 0 || rc == 0) continue;
                const float sah = c->traversal_cost +
                    ((float)lc * box_area(&acc) + (float)rc * box_area(&rb[b + 1])) * inv_area;
                if (sah < best) {
                    best = sah; axis = a;
                    border = cmin[a] + (float)(b + 1) * (ext / (float)nb);
                }
            }
        }
        free(bb); free(bc); free(rb);
    }

    int32_t i = begin, j = end - 1;
    while (i <= j) {
        if (c->centroid[c->prims[i]][axis] < border) ++i;
        else { int32_t tmp = c->prims[i]; c->prims[i] = c->prims[j]; c->prims[j] = tmp; --j; }
    }
    int32_t mid = i;
    if (mid == begin || mid == end) mid = begin + n / 2;

    node->is_leaf = 0; node->startidx = -1; node->numprims = -1;
    node->lc = build_rec(c, begin, mid);
    node->rc = build_rec(c, mid, end);
    return node;
}

int o_tree_build(const o_bbox *leaf_bounds, uint32_t n, int use_sah, int num_bins,
                 float traversal_cost, o_tree *out)
{
    if (n == 0 || num_bins < 2) return -1;
    memset(out, 0, sizeof(*out));
    out->pool = (o_tree_node *)calloc((size_t)2 * n, sizeof(o_tree_node));
    out->indices = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    build_ctx c;
    c.leaf = leaf_bounds;
    c.centroid = (float (*)[3])malloc(sizeof(float[3]) * (size_t)n);
    c.prims = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    c.t = out; c.use_sah = use_sah; c.num_bins = num_bins; c.traversal_cost = traversal_cost;
    for (uint32_t i = 0; i < n; ++i) {
        c.prims[i] = (int32_t)i;
        for (int k = 0; k < 3; ++k)
            c.centroid[i][k] = 0.5f * (leaf_bounds[i].pmax[k] + leaf_bounds[i].pmin[k]);
    }
    out->root = build_rec(&c, 0, (int32_t)n);
    free(c.centroid); free(c.prims);
    return 0;
}

void o_tree_free(o_tree *t)
{
    free(t->pool); free(t->indices);
    memset(t, 0, sizeof(*t));
}

int o_tree_from_arrays(const o_bbox *bounds, const int32_t *left, const int32_t *right,
                       const int32_t *prim, int32_t n_nodes, o_tree *out)
{
    if (n_nodes < 1) return -1;
    memset(out, 0, sizeof(*out));
    out->pool = (o_tree_node *)calloc((size_t)n_nodes, sizeof(o_tree_node));
    out->indices = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_nodes);
    for (int32_t i = 0; i < n_nodes; ++i) {
        o_tree_node *nd = &out->pool[i];
        nd->bounds = bounds[i];
        if (left[i] < 0) {
            nd->is_leaf = 1; nd->lc = nd->rc = NULL;
            nd->startidx = out->n_indices; nd->numprims = 1;
            out->indices[out->n_indices++] = prim[i];
        } else {
            if (left[i] >= n_nodes || right[i] < 0 || right[i] >= n_nodes) return -1;
            nd->is_leaf = 0; nd->lc = &out->pool[left[i]]; nd->rc = &out->pool[right[i]];
            nd->startidx = -1; nd->numprims = -1;
        }
    }
    out->root = &out->pool[0];
    out->nodecnt = n_nodes;
    return 0;
}

/* src/bvh/RadeonRays/plain_bvh_translator.cpp:223-246.  While numbering,
 * pmin.w holds -1 for a leaf and the right child's index for an internal
 * node; first/count remember the leaf's slice of the reordering array. */
typedef struct { o_bbox *nodes; int32_t *first; int32_t *count; int32_t nodecnt; } lin_ctx;

static int32_t process_node(lin_ctx *c, const o_tree_node *n)
{
    const int32_t idx = c->nodecnt;
    c->nodes[idx] = n->bounds;
    c->nodecnt++;
    if (n->is_leaf) {
        c->first[idx] = n->startidx;
        c->count[idx] = n->numprims;
        c->nodes[idx].pmin[3] = -1.0f;
    } else {
        process_node(c, n->lc);
        c->nodes[idx].pmin[3] = (float)process_node(c, n->rc);
        c->first[idx] = -1;
        c->count[idx] = -1;
    }
    return idx;
}

/* src/bvh/RadeonRays/plain_bvh_translator.cpp:39-96 */
int o_linearize(const o_tree *t, o_bbox *out_nodes)
{
    if (!t->root) return -1;                                  /* assert at :47 */
    lin_ctx c;
    c.nodes = out_nodes;
    c.first = (int32_t *)malloc(sizeof(int32_t) * (size_t)t->nodecnt);
    c.count = (int32_t *)malloc(sizeof(int32_t) * (size_t)t->nodecnt);
    c.nodecnt = 0;
    process_node(&c, t->root);
    int rc = 0;
    if (c.nodecnt != t->nodecnt) rc = -2;                     /* assert at :55 */
    const int32_t n = c.nodecnt;

    out_nodes[0].pmax[3] = -1.0f;                             /* :59 */
    for (int32_t i = 0; i < n; ++i) {                         /* :62-69 */
        if (out_nodes[i].pmin[3] != -1.0f) {
            out_nodes[i + 1].pmax[3] = out_nodes[i].pmin[3];
            out_nodes[(int32_t)out_nodes[i].pmin[3]].pmax[3] = out_nodes[i].pmax[3];
        }
    }
    for (int32_t i = 0; i < n; ++i) {                         /* :72-95 */
        if (out_nodes[i].pmin[3] == -1.0f) {
            const int32_t startidx = c.first[i];
            for (int32_t j = 1; j < c.count[i]; ++j)          /* assert at :85 */
                if (t->indices[startidx + j] != t->indices[startidx] + j) rc = -3;
            out_nodes[i].pmin[3] = (float)(uint32_t)t->indices[startidx];
        } else {
            out_nodes[i].pmin[3] = -1.0f;
        }
    }
    free(c.first); free(c.count);
    return rc;
}

/* src/bvh/bvh_builder.cpp:50-76 */
int o_build_bvh(const o_bbox *leaf_bounds, uint32_t n, int use_sah, int num_bins,
                float traversal_cost, o_bbox *out_nodes)
{
    o_bbox *scaled = (o_bbox *)malloc(sizeof(o_bbox) * (size_t)n);
    memcpy(scaled, leaf_bounds, sizeof(o_bbox) * (size_t)n);
    o_inflate_leaf_bounds(scaled, n);
    o_tree t;
    int rc = o_tree_build(scaled, n, use_sah, num_bins, traversal_cost, &t);
    if (rc == 0) {
        rc = o_linearize(&t, out_nodes);
        o_tree_free(&t);
    }
    free(scaled);
    return rc;
}

/* ---------------------------------------------------------------------- */
/* IntersectSceneClosest with the Moller-Trumbore leaf test: the variant the
 * reference keeps commented out at shaders/RadeonRays/bvh.glslh:339.  The
 * traversal (:417-456) is unchanged; IntersectTriangle (:181-202) takes the
 * un-permuted ray and writes uvwt = (b1, b2, 0, t). */
static void intersect_scene_closest_mt(const o_bbox *nodes, const float *vertices,
                                       const int32_t *indices, const o_ray *r, o_isect *isect)
{
    const float invdir[3] = { 1.0f / r->d[0], 1.0f / r->d[1], 1.0f / r->d[2] };
    isect->uvwt[0] = 0.0f; isect->uvwt[1] = 0.0f; isect->uvwt[2] = 0.0f; isect->uvwt[3] = r->o[3];
    isect->shapeid = -1; isect->primid = -1; isect->pad0 = 667; isect->pad1 = 0;
    int32_t idx = 0;
    while (idx != -1) {
        const o_bbox *node = &nodes[idx];
        if (o_intersect_box(r, invdir, node, isect->uvwt[3])) {
            if (node_is_leaf(node)) {
                const int64_t start = node_startidx(node);
                float v1[3], v2[3], v3[3];
                get_vertex(vertices, indices[3 * start + 0], v1);
                get_vertex(vertices, indices[3 * start + 1], v2);
                get_vertex(vertices, indices[3 * start + 2], v3);
                if (o_intersect_triangle_mt(r, v1, v2, v3, isect)) {
                    isect->primid = (int32_t)node->pmin[3];
                    isect->shapeid = 0;
                }
                idx = (int32_t)node->pmax[3];
            } else {
                ++idx;
            }
        } else {
            idx = (int32_t)node->pmax[3];
        }
    }
}

void o_closest_mt_batch(const o_bbox *nodes, const float *vertices, const int32_t *indices,
                        const o_ray *rays, uint64_t n, o_isect *out, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
    for (int64_t i = 0; i < (int64_t)n; ++i)
        intersect_scene_closest_mt(nodes, vertices, indices, &rays[i], &out[i]);
}

int o_linearize_arrays(const o_bbox *bounds, const int32_t *left, const int32_t *right,
                       const int32_t *prim, int32_t n_nodes, o_bbox *out_nodes)
{
    o_tree t;
    int rc = o_tree_from_arrays(bounds, left, right, prim, n_nodes, &t);
    if (rc == 0) rc = o_linearize(&t, out_nodes);
    o_tree_free(&t);
    return rc;
}

void o_depth_shade_batch(const o_isect *isects, uint64_t n, float *rgba)
{
    for (uint64_t i = 0; i < n; ++i) o_depth_shade(&isects[i], rgba + 4 * i);
}

/* ---------------------------------------------------------------------- */
/* IntersectSceneClosest over nodes whose pmin.w / pmax.w hold int32 bit patterns
 * (leaf payload or 0xFFFFFFFF; next-node index or 0xFFFFFFFF): the format documented at
 * src/bvh/bvh_builder.h:29-33 and left as TODOs in plain_bvh_translator.cpp:58,63,66.
 * The reference itself only emits and reads float-valued fields, which stop being exact
 * at 2^24 nodes; this variant exists so that scenes beyond that size have a checker.  The
 * loop is the one of shaders/RadeonRays/bvh.glslh:417-456 with the two decodes swapped. */
static inline int32_t bits_of(float f) { int32_t i; memcpy(&i, &f, 4); return i; }

static void intersect_scene_closest_intbits(const o_bbox *nodes, const float *vertices,
                                            const int32_t *indices, const o_ray *r, o_isect *isect)
{
    o_ray_internal ri;
    o_precompute_ray(r, &ri);
    const float invdir[3] = { 1.0f / r->d[0], 1.0f / r->d[1], 1.0f / r->d[2] };
    isect->uvwt[0] = 0.0f; isect->uvwt[1] = 0.0f; isect->uvwt[2] = 0.0f; isect->uvwt[3] = r->o[3];
    isect->shapeid = -1; isect->primid = -1; isect->pad0 = 667; isect->pad1 = 0;
    int32_t idx = 0;
    while (idx != -1) {
        const o_bbox *node = &nodes[idx];
        if (o_intersect_box(&ri.ray, invdir, node, isect->uvwt[3])) {
            const int32_t payload = bits_of(node->pmin[3]);
            if (payload != -1) {
                const int64_t start = payload;
                float v1[3], v2[3], v3[3];
                get_vertex(vertices, indices[3 * start + 0], v1);
                get_vertex(vertices, indices[3 * start + 1], v2);
                get_vertex(vertices, indices[3 * start + 2], v3);
                if (o_intersect_triangle_watertight(&ri, v1, v2, v3, isect)) {
                    isect->primid = payload;
                    isect->shapeid = 0;
                }
                idx = bits_of(node->pmax[3]);
            } else {
                ++idx;
            }
        } else {
            idx = bits_of(node->pmax[3]);
        }
    }
}

void o_closest_intbits_batch(const o_bbox *nodes, const float *vertices, const int32_t *indices,
                             const o_ray *rays, uint64_t n, o_isect *out, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
    for (int64_t i = 0; i < (int64_t)n; ++i)
        intersect_scene_closest_intbits(nodes, vertices, indices, &rays[i], &out[i]);
}
