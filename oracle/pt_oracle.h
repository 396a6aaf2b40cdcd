/*
 * pt_oracle.h -- CPU restatement of the reference's per-sample hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library may include, link
 * or call this; it is the checker the HIP path is compared against and the
 * host-core baseline that bench.py times beside the GPU numbers.
 *
 * PARITY STATUS: the reference ships no tests, golden vectors or runnable
 * build (its shaders are GLSL, its BVH builder lives in an un-vendored
 * submodule), so this restatement is pinned only by the few literals the
 * reference text holds: the fallback quad mesh (src/main.cpp:918-921), the
 * unit triangle (shaders/depth.frag:133-136), the 667 marker and the
 * shapeid 0 / -1 convention (shaders/RadeonRays/bvh.glslh:423-430,344-345).
 * Everything else is "parity unpinned": agreement is established three ways,
 * oracle <-> brute-force scan <-> HIP kernels, on seeded synthetic inputs.
 *
 * Floating-point contract (GLSL leaves these open; this file closes them):
 * IEEE binary32, round-to-nearest-even, no FMA contraction, correctly rounded
 * division and square root, min/max with fminf/fmaxf semantics (a NaN operand
 * is ignored).  Build with -ffp-contract=off -fno-fast-math.
 *
 * All file:line citations are relative to the reference checkout.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* shaders/RadeonRays/bvh.glslh:29-33 -- one 32-byte skip-link node. */
typedef struct { float pmin[4]; float pmax[4]; } o_bbox;
/* shaders/RadeonRays/bvh.glslh:37-43 -- o.w is the ray's maximum distance. */
typedef struct { float o[4]; float d[4]; } o_ray;
/* shaders/RadeonRays/bvh.glslh:94-101 */
typedef struct { int32_t shapeid, primid, pad0, pad1; float uvwt[4]; } o_isect;
/* shaders/RadeonRays/bvh.glslh:45-49 */
typedef struct { o_ray ray; float s[3]; float d[3]; } o_ray_internal;

/* Camera inputs depth.frag reads from the `global` uniform block
 * (shaders/depth.frag:59-60,70,74): inv_projection[0][0], [1][1], the three
 * columns of mat3(inv_view), camera_position. */
typedef struct {
    float ip00, ip11;
    float inv_view_c0[3], inv_view_c1[3], inv_view_c2[3];
    float camera_position[3];
} o_camera;

/* Traversal work counters, summed over a batch. */
typedef struct { uint64_t box_tests, tri_tests; } o_counters;

/* --- single-function restatements ------------------------------------- */
void o_precompute_ray(const o_ray *r, o_ray_internal *out);
int  o_intersect_box(const o_ray *r, const float invdir[3], const o_bbox *box, float maxt);
int  o_intersect_triangle_mt(const o_ray *r, const float v1[3], const float v2[3],
                             const float v3[3], o_isect *isect);
int  o_intersect_triangle_watertight(const o_ray_internal *r, const float v0[3],
                                     const float v1[3], const float v2[3], o_isect *isect);
int  o_intersect_triangle_p(const o_ray *r, const float v1[3], const float v2[3],
                            const float v3[3]);

/* --- scene queries ----------------------------------------------------- */
void o_intersect_scene_closest(const o_bbox *nodes, const float *vertices, const int32_t *indices,
                               const o_ray *r, o_isect *isect, o_counters *cnt);
int  o_intersect_scene_any(const o_bbox *nodes, const float *vertices, const int32_t *indices,
                           const o_ray *r);
/* Batch forms; nthreads <= 1 runs serially, otherwise OpenMP static chunks. */
void o_closest_batch(const o_bbox *nodes, const float *vertices, const int32_t *indices,
                     const o_ray *rays, uint64_t n, o_isect *out, o_counters *cnt, int nthreads);
void o_any_batch(const o_bbox *nodes, const float *vertices, const int32_t *indices,
                 const o_ray *rays, uint64_t n, int32_t *out, int nthreads);
/* Same traversal with the Moller-Trumbore leaf test the reference keeps commented out at
 * shaders/RadeonRays/bvh.glslh:339. */
void o_closest_mt_batch(const o_bbox *nodes, const float *vertices, const int32_t *indices,
                        const o_ray *rays, uint64_t n, o_isect *out, int nthreads);
/* IntersectSceneClosest over nodes in the integer-bit-pattern encoding documented at
 * src/bvh/bvh_builder.h:29-33 (the reference's shader only reads float-valued fields, exact
 * below 2^24 nodes; this is the checker for larger scenes). */
void o_closest_intbits_batch(const o_bbox *nodes, const float *vertices, const int32_t *indices,
                             const o_ray *rays, uint64_t n, o_isect *out, int nthreads);
/* Linear scan over every triangle in index order, same triangle test and the
 * same running tMax (pattern: shaders/depth.frag:110-129). */
void o_brute_closest_batch(const float *vertices, const int32_t *indices, uint32_t n_tri,
                           const o_ray *rays, uint64_t n, o_isect *out, int nthreads);

/* --- two-level scenes ---------------------------------------------------
 * The reference documents a top-level BVH whose leaf payload is an instance index
 * (src/bvh/bvh_builder.h:29-33) but its code for one is disabled
 * (src/bvh/RadeonRays/plain_bvh_translator.cpp:98-101,134-137) and no shader walks one, so
 * this is a specification of this project, "parity unpinned" throughout: the top level is
 * walked with the loop of IntersectSceneClosest / IntersectSceneAny on the world-space ray; a
 * top-level leaf takes the ray to the instance's object space (direction not renormalised, so
 * distances agree in both spaces) and runs the same loop on the mesh with the running nearest
 * distance as the limit.  Hits report shapeid = instance, primid = triangle of its mesh. */
typedef struct { const o_bbox *nodes; const float *vertices; const int32_t *indices; } o_mesh;
typedef struct { float inv[12]; int32_t mesh; int32_t pad[3]; } o_instance;
/* Inverse of a 3x4 transform (rows of linear part | translation): binary64 cofactors and
 * determinant, one rounding per entry.  Returns 0 if singular. */
int  o_invert_affine(const float m[12], float inv[12]);
/* out = M p per point, ((a+b)+c)+d in binary32. */
void o_transform_points(const float m[12], const float *p, uint64_t n, float *out);
/* Box of the eight transformed corners of root. */
void o_instance_bounds(const float m[12], const o_bbox *root, o_bbox *out);
void o_two_level_closest(const o_bbox *top, const o_mesh *meshes, const o_instance *instances,
                         const o_ray *r, o_isect *isect, o_counters *cnt);
int  o_two_level_any(const o_bbox *top, const o_mesh *meshes, const o_instance *instances,
                     const o_ray *r);
void o_two_level_closest_batch(const o_bbox *top, const o_mesh *meshes,
                               const o_instance *instances, const o_ray *rays, uint64_t n,
                               o_isect *out, o_counters *cnt, int nthreads);
void o_two_level_any_batch(const o_bbox *top, const o_mesh *meshes, const o_instance *instances,
                           const o_ray *rays, uint64_t n, int32_t *out, int nthreads);

/* --- raygen + debug shade ---------------------------------------------- */
void o_raygen_reference(const o_camera *cam, uint32_t width, uint32_t height,
                        uint32_t px, uint32_t py, o_ray *out);
void o_raygen_reference_image(const o_camera *cam, uint32_t width, uint32_t height, o_ray *out);
void o_depth_shade(const o_isect *isect, float rgba[4]);
void o_depth_shade_batch(const o_isect *isects, uint64_t n, float *rgba);

/* --- host BVH path ------------------------------------------------------ */
void o_mesh_leaf_bounds(const float *vertices, const int32_t *indices, uint32_t n_tri,
                        o_bbox *out);
void o_inflate_leaf_bounds(o_bbox *boxes, uint32_t n);

/* Pointer tree in the shape the linearizer consumes
 * (src/bvh/RadeonRays/plain_bvh_translator.cpp:223-246 reads bounds, type,
 * lc, rc, startidx, numprims). */
typedef struct o_tree_node {
    o_bbox bounds;
    int32_t is_leaf;
    struct o_tree_node *lc, *rc;
    int32_t startidx, numprims;
} o_tree_node;

typedef struct {
    o_tree_node *root;
    o_tree_node *pool;
    int32_t nodecnt;
    int32_t *indices;   /* primitive reordering, leaf emission order */
    int32_t n_indices;
} o_tree;

/* use_sah = 0: spatial median of the centroid bounds on their longest axis;
 * use_sah = 1: binned SAH on all three axes (num_bins, traversal_cost).
 * One primitive per leaf.  Returns 0 on success. */
int  o_tree_build(const o_bbox *leaf_bounds, uint32_t n, int use_sah, int num_bins,
                  float traversal_cost, o_tree *out);
void o_tree_free(o_tree *t);
/* Tree from explicit arrays (children < 0 mark a leaf; prim = its primitive). */
int  o_tree_from_arrays(const o_bbox *bounds, const int32_t *left, const int32_t *right,
                        const int32_t *prim, int32_t n_nodes, o_tree *out);
/* src/bvh/RadeonRays/plain_bvh_translator.cpp:39-96 */
int  o_linearize(const o_tree *t, o_bbox *out_nodes);
int  o_linearize_arrays(const o_bbox *bounds, const int32_t *left, const int32_t *right,
                        const int32_t *prim, int32_t n_nodes, o_bbox *out_nodes);
/* src/bvh/bvh_builder.cpp:50-76: inflate, build, linearize.  out_nodes holds
 * 2n-1 entries. */
int  o_build_bvh(const o_bbox *leaf_bounds, uint32_t n, int use_sah, int num_bins,
                 float traversal_cost, o_bbox *out_nodes);

#ifdef __cplusplus
}
#endif
#endif
