/*
 * mipt.h -- C ABI of the MI355X path-tracing core (libmipt.so).
 *
 * The reference (zogi/gpu-pathtracer) has no plugin or FFI layer; its boundary
 * between host C++ and GPU is the set of buffers and calls in
 * src/main_granite.cpp (build_mesh_bvh :115-178, init_device_data :180-194,
 * the "path_trace" pass callback :402-451) and src/bvh/bvh_builder.h:37.
 * Every entry point below names the reference call it stands in for.
 *
 * Conventions
 *  - Plain C types only: pointers, sizes, PODs.  No exceptions cross the ABI.
 *  - Every function returning int returns PT_OK (0) or a PT_ERR_* code; the
 *    message of the calling thread's last failure is pt_last_error().
 *    (Reference behaviour: assert/abort, src/main.cpp:26-33.)
 *  - Handles are opaque and owned by the caller until the matching *_destroy.
 *    The library never frees or keeps caller memory; host arrays passed in are
 *    copied before the call returns.
 *  - A pt_context belongs to one GPU and is not re-entrant; different contexts
 *    may be driven from different threads or processes.
 *  - Pointers named *_dev are device pointers valid on the context's GPU and
 *    the call is asynchronous on the context's stream; all others are host
 *    pointers and the call returns after the result is in host memory.
 */
#ifndef MIPT_H
#define MIPT_H

#include <stddef.h>
#include <stdint.h>

#include "pt_config.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PT_OK               0
#define PT_ERR_INVALID      1   /* bad argument or malformed input */
#define PT_ERR_DEVICE       2   /* HIP runtime failure, or no GPU */
#define PT_ERR_UNSUPPORTED  3
#define PT_ERR_IO           4
#define PT_ERR_NOMEM        5
#define PT_ERR_STALE        6   /* a cache file that was built from other inputs */

/* How leaf payloads and skip links are stored in pmin.w / pmax.w. */
#define PT_LINKS_FLOAT_VALUE 0u /* float(index), -1.f: what the reference emits
                                   (plain_bvh_translator.cpp:59,65-67,89,240); exact below 2^24 */
#define PT_LINKS_INT_BITS    1u /* int32 bit pattern, 0xFFFFFFFF: the format documented at
                                   src/bvh/bvh_builder.h:29-33 and left as a TODO in the translator */

typedef struct pt_context pt_context;
typedef struct pt_bvh pt_bvh;
typedef struct pt_mesh pt_mesh;
typedef struct pt_model pt_model;
typedef struct pt_scene pt_scene;
typedef struct pt_film pt_film;

/* shaders/RadeonRays/bvh.glslh:29-33; RadeonRays::bbox as uploaded byte-for-byte at
 * src/main_granite.cpp:191. */
typedef struct pt_bbox { float pmin[4]; float pmax[4]; } pt_bbox;
/* shaders/RadeonRays/bvh.glslh:37-43.  o[3] = maximum distance, d[3] = time (unused). */
typedef struct pt_ray { float o[4]; float d[4]; } pt_ray;
/* shaders/RadeonRays/bvh.glslh:94-101.  Miss: shapeid = primid = -1, uvwt = (0,0,0,tmax).
 * Hit: shapeid = 0, primid = index of the triangle in the caller's index buffer,
 * uvwt = (b0,b1,b2,t).  padding = (667, 0) always (:429-430). */
typedef struct pt_hit { int32_t shapeid, primid, padding[2]; float uvwt[4]; } pt_hit;

/* Builder options with the reference's keys and defaults
 * (src/bvh/bvh_builder.cpp:18-32). */
typedef struct pt_bvh_options {
    int32_t use_sah;            /* "bvh.builder" == "sah"                    default 0 */
    int32_t use_splits;         /* "bvh.sah.use_splits" > 0                  default 0 */
    int32_t max_split_depth;    /* "bvh.sah.max_split_depth"                 default 10 */
    int32_t num_bins;           /* "bvh.sah.num_bins"                        default 64 */
    float   min_overlap;        /* "bvh.sah.min_overlap"                     default 0.05 */
    float   traversal_cost;     /* "bvh.sah.traversal_cost"                  default 10 */
    float   extra_node_budget;  /* "bvh.sah.extra_node_budget"               default 0.5 */
    uint32_t link_encoding;     /* PT_LINKS_*                                default FLOAT_VALUE */
    int32_t num_threads;        /* host threads for the build, 0 = all cores */
    int32_t optimize_passes;    /* "mipt.optimize_passes": rounds of insertion-based optimisation
                                   of the finished tree (not in the reference)   default 0 */
} pt_bvh_options;

/* Work counters of one trace call, summed over its rays. */
typedef struct pt_trace_stats {
    uint64_t rays;
    uint64_t box_tests;         /* AABB slab tests */
    uint64_t tri_tests;         /* ray/triangle tests */
    uint64_t tie_rays;          /* rays re-resolved in skip-link order */
    uint64_t lds_box_tests;     /* of box_tests: on nodes served from the LDS copy of the top of the tree */
    uint64_t long_rays;         /* rays finished by a whole wave each (see "extend.tail_budget") */
} pt_trace_stats;

/* Counters of one pt_render call. */
typedef struct pt_render_stats {
    uint64_t paths;             /* camera samples started */
    uint64_t rays_primary;
    uint64_t rays_bounce;
    uint64_t rays_shadow;
    uint64_t tie_rays;
    uint64_t extend_launches;
    double   extend_ms;         /* summed device time of the closest-hit kernels, if timing is on */
    uint64_t box_tests;         /* with PT_RENDER_COUNT_WORK: AABB slab tests ... */
    uint64_t tri_tests;         /* ... and ray/triangle tests of those kernels */
    double   raygen_ms;         /* if timing is on: summed device time of the other kernels, */
    double   shade_ms;          /* by kind */
    double   occlude_ms;
    double   accumulate_ms;
    uint64_t lds_box_tests;     /* with PT_RENDER_COUNT_WORK: slab tests on nodes served from LDS */
} pt_render_stats;

/* ------------------------------------------------------------------------ */
/* Library                                                                  */
const char *pt_last_error(void);
const char *pt_version(void);
/* Number of GPUs visible to the process; 0 without a GPU (never fails). */
int pt_device_count(void);

/* ------------------------------------------------------------------------ */
/* Host BVH path                                                            */

void pt_bvh_options_default(pt_bvh_options *opt);
/* RadeonRays::Options::SetValue as used at src/main_granite.cpp:175-176; keys as read at
 * src/bvh/bvh_builder.cpp:18-24.  Numeric values are parsed from the string. */
int pt_bvh_options_set(pt_bvh_options *opt, const char *key, const char *value);

/* The per-triangle box recipe of build_mesh_bvh, src/main_granite.cpp:162-172.
 * vtx: xyz packed (n_vtx*3 floats); idx: n_tri*3 vertex indices; out: n_tri boxes. */
int pt_mesh_leaf_bounds(const float *vtx, uint32_t n_vtx, const int32_t *idx, uint32_t n_tri,
                        pt_bbox *out);

/* bvh::build_bvh, src/bvh/bvh_builder.h:37 (bvh_builder.cpp:50-76): inflates each leaf box
 * by 1+1e-4 about its centre, builds a binary tree with one primitive per leaf and
 * linearizes it into 2n-1 skip-link nodes (format: src/bvh/bvh_builder.h:26-34). */
int pt_bvh_build(const pt_bbox *leaf_bounds, uint32_t n_leaves, const pt_bvh_options *opt,
                 pt_bvh **out);
/* The insertion-based optimisation of "mipt.optimize_passes" applied to any skip-link tree (one
 * built elsewhere, or read back from pt_scene_build): same leaves, rearranged above them.  Not
 * in the reference. */
int pt_bvh_optimize(const pt_bbox *nodes, uint32_t n_nodes, uint32_t link_encoding, int32_t passes,
                    pt_bvh **out);
uint32_t pt_bvh_node_count(const pt_bvh *bvh);
/* vector<bbox> returned by build_bvh; valid until pt_bvh_destroy. */
const pt_bbox *pt_bvh_nodes(const pt_bvh *bvh);
uint32_t pt_bvh_link_encoding(const pt_bvh *bvh);
int pt_bvh_destroy(pt_bvh *bvh);

/* PlainBvhTranslator::Process, src/bvh/RadeonRays/plain_bvh_translator.cpp:39-96, for a tree
 * given as arrays: node i is a leaf when left[i] < 0 (then prim[i] is its primitive),
 * otherwise left[i]/right[i] are its children.  Node 0 is the root.  out_nodes: n_nodes. */
int pt_bvh_linearize(const pt_bbox *bounds, const int32_t *left, const int32_t *right,
                     const int32_t *prim, uint32_t n_nodes, uint32_t link_encoding,
                     pt_bbox *out_nodes);

/* Checks every structural property of a skip-link array (pre-order, left child adjacent,
 * links, one triangle per leaf and every triangle in at least one leaf -- several with
 * spatial splits; child boxes inside parents' boxes is NOT required). */
int pt_bvh_validate(const pt_bbox *nodes, uint32_t n_nodes, uint32_t link_encoding,
                    uint32_t n_tri);

/* ------------------------------------------------------------------------ */
/* OBJ ingest: the fields the reference takes from tinyobj::LoadObj at
 * src/main.cpp:860-899 (attrib.vertices, mesh.indices[].vertex_index of every shape,
 * triangulated), plus per-face material ids and the MTL Kd/Ks/Ke/Ns entries. */
int pt_obj_load(const char *path, pt_mesh **out);
/* glTF 2.0 (.gltf / .glb), the live application's scene format
 * (scene_loader_.load_scene, src/main_granite.cpp:265-268).  PT_GLTF_FIRST_MESH_ONLY gives what
 * that application uses: the first triangle primitive's positions and indices, untransformed
 * (:286-298).  Without it every triangle primitive of the default scene is merged, positions
 * in world space, with base colour / metallic / roughness / emissive factors as materials. */
#define PT_GLTF_FIRST_MESH_ONLY 1u
int pt_gltf_load(const char *path, uint32_t flags, pt_mesh **out);
uint32_t pt_mesh_vertex_count(const pt_mesh *m);
uint32_t pt_mesh_triangle_count(const pt_mesh *m);
uint32_t pt_mesh_material_count(const pt_mesh *m);
const float *pt_mesh_vertices(const pt_mesh *m);        /* xyz packed */
const int32_t *pt_mesh_indices(const pt_mesh *m);       /* 3 per triangle */
const int32_t *pt_mesh_material_ids(const pt_mesh *m);  /* 1 per triangle */
const pt_material *pt_mesh_materials(const pt_mesh *m);
/* OBJ files only (0 / NULL otherwise): the shapes tinyobj::LoadObj would cut the file into
 * (src/main.cpp:860), as one shape index and one index within the shape per triangle -- the
 * shape_id / prim_id of the Face records the legacy builder fills
 * (src/bvh/RadeonRays/intersector_skip_links.cpp:236-264).  See pt_scene_set_face_ids. */
uint32_t pt_mesh_shape_count(const pt_mesh *m);
const int32_t *pt_mesh_shape_ids(const pt_mesh *m);       /* 1 per triangle */
const int32_t *pt_mesh_shape_prim_ids(const pt_mesh *m);  /* 1 per triangle */
/* "Scale model into unit cube", src/main.cpp:865-880. */
int pt_mesh_normalize_unit_cube(pt_mesh *m);
int pt_mesh_destroy(pt_mesh *m);

/* ------------------------------------------------------------------------ */
/* Device side                                                              */

/* One context per GPU (the reference's Vulkan device, src/main_granite.cpp:311-315).
 * Fails with PT_ERR_DEVICE when the process has no such GPU: there is no CPU fallback. */
int pt_context_create(int device, pt_context **out);
/* Run all work of this context on an existing HIP stream (hipStream_t) instead of the
 * context's own; pass NULL to return to the own stream. */
int pt_context_set_stream(pt_context *ctx, void *hip_stream);
void *pt_context_stream(pt_context *ctx);
int pt_context_synchronize(pt_context *ctx);
/* When on, every kernel launch of pt_render is bracketed by HIP events on the context's
 * stream and its device time summed into pt_render_stats.{extend,raygen,shade,occlude,
 * accumulate}_ms. */
int pt_context_set_kernel_timing(pt_context *ctx, int on);
/* With timing on: device time in milliseconds of the traversal kernel of the context's last
 * pt_trace_closest* call (the kernel alone: not the copies of the host-pointer forms, nor the
 * sorting of the rays).  Waits for that kernel. */
int pt_context_last_trace_ms(pt_context *ctx, double *ms);
/* Tuning knobs (integers as strings).  "extend.persistent" 1|0: closest-hit kernel of
 * pt_render as persistent waves that refill idle lanes (default 1) or one thread per ray;
 * "extend.refill" 1..64: idle lanes that trigger a refill (default 24);
 * "extend.leaf_lanes" 1..64: lanes waiting at a triangle that trigger a triangle step of the
 * wave (default 16); "extend.blocks_per_cu" 1..8: persistent workgroups per compute unit
 * (default 5); "extend.stack_entries" -1|9..30 and "extend.cache_nodes" -1|0..448: how a
 * persistent workgroup's LDS is split between the lanes' traversal stacks (entries per lane;
 * rays that need more go on in global memory) and a copy of the first nodes of the tree,
 * breadth first from the root, which the steps every ray takes first then read from LDS
 * instead of through the vector memory pipeline (default -1: 16 entries + 224 nodes for trees
 * up to 40 levels deep that fit the 256 MB Infinity Cache, 24 + 96 for deeper or larger ones;
 * single-level scenes only);
 * "extend.quad" -1|0|1: closest hits and occlusion queries walk the tree (of a two-level scene:
 * the meshes' trees) as 64-byte four-child nodes with 8-bit child boxes and exact leaf boxes
 * beside the triangles -- half the records per ray for more arithmetic per record (default -1:
 * trace calls on scenes beyond the Infinity Cache, render passes on such scenes from the
 * second bounce on); "scene.quad" 1|0: scenes created from now
 * on get that form built (64 bytes per node and per triangle; default 1);
 * "extend.tail_budget" 0..1000000 (default 64): closest hits on single-level scenes walked as
 * four-child nodes (see "extend.quad") -- once a
 * wave of the persistent kernel has no more rays to start, it may spend this many more rounds
 * (a run of node steps and a triangle step) on the rays it still holds; what is left then (a
 * ray along a fan of slivers can make thousands of steps) goes to a second kernel that searches
 * each such ray with a whole wave, so that one ray does not hold the launch (0: every wave
 * finishes its own rays);
 * "any.exact_order" 0|1: occlusion queries walk the reference's skip links
 * instead of the two-child nodes (default 0); "trace.sort_rays" -1|0|1: pt_trace_closest*
 * takes the rays in Morton order of their origins (default -1: when the scene exceeds the
 * 256 MB Infinity Cache and the call has at least 2^18 rays).  Results do not depend on any
 * of them.  "build.ploc_radius" 0..64 selects pt_scene_build's tree: 0 (default) a binary radix
 * tree over the Morton codes, r > 0 locally-ordered clustering with search radius r (lower
 * SAH cost, a few times the build time; it changes the tree, not any hit).
 * "build.optimize_passes" 0..32 (default 6): passes of parallel insertion-based optimisation
 * over pt_scene_build's tree (and over the meshes pt_scene_create_instanced builds): every
 * node looks for the place where it would cost the tree the least surface area, and the moves
 * that do not touch each other's nodes are made; "build.treelet_every" 0..32 (default 3): after
 * every so many of those passes one of treelet restructuring (the top of the subtree below
 * every node rebuilt into its cheapest arrangement; 0: none).  About 1 ms per insertion pass
 * and million triangles; six passes with the default treelet passes bring the radix tree to
 * the host SAH tree's closest-hit rate (within 1 % either way on the scenes measured).
 * Deterministic.  "build.split_ratio" 0..1000000 (default 4), "build.split_max" 1..256 (default
 * 64), "build.split_budget" 0..1000 per cent (default 50): triangles that go into several
 * leaves (see pt_scene_build).  "build.max_levels" 8..256 (default 256): an
 * optimised tree with more levels than this is given up for the plain radix tree, whose
 * depth the key length bounds (256 is what the level lists of the passes hold). */
int pt_context_set_option(pt_context *ctx, const char *key, const char *value);
/* Measurement aid, no counterpart in the reference: the rate (records per second) at which
 * the context's GPU serves dependent random gathers of record_bytes (16, 32, 48 or 64) from a
 * table of table_bytes (rounded down to a power-of-two number of records), every lane of
 * "extend.blocks_per_cu" resident workgroups per compute unit chasing `iterations` records
 * one after the other.  That is a BVH traversal's access pattern with the arithmetic taken
 * out: with the table sized to a cache level it is the ceiling of a traversal whose working
 * set lies there.  Allocates and frees the table; blocking. */
int pt_measure_gather(pt_context *ctx, uint64_t table_bytes, uint32_t record_bytes,
                      uint32_t iterations, double *records_per_s);
/* Fails with PT_ERR_INVALID, and destroys nothing, while a pt_comm made on the context is alive
 * (a communicator uses its context's device and stream until pt_comm_destroy). */
int pt_context_destroy(pt_context *ctx);

/* init_device_data, src/main_granite.cpp:180-194: uploads the three storage buffers the
 * shader binds (nodes / vertices / indices, shaders/RadeonRays/bvh.glslh:103-117) and
 * derives the re-laid-out copies the ordered traversal reads.  materials / material_ids
 * may be NULL (a single grey diffuse material is used).  The node array must be a tree in the
 * format of src/bvh/bvh_builder.h:26-34 in which every child's box lies within its parent's
 * (what a builder that unites its children's boxes produces; PT_ERR_INVALID otherwise): the
 * hits of the nearer-child-first traversal equal the shader's loop only on such trees. */
int pt_scene_create(pt_context *ctx, const pt_bbox *nodes, uint32_t n_nodes,
                    uint32_t link_encoding, const float *vtx, uint32_t n_vtx,
                    const int32_t *idx, uint32_t n_tri, const pt_material *materials,
                    uint32_t n_materials, const int32_t *material_ids, pt_scene **out);
/* build_mesh_bvh + init_device_data (src/main_granite.cpp:115-194) entirely on the GPU: leaf
 * boxes (:162-172), their inflation (src/bvh/bvh_builder.cpp:53-59, bit-identical to the host
 * path), a linear BVH over the Morton-ordered box centres, the skip-link array in the format
 * of src/bvh/bvh_builder.h:26-34 and the traversal layouts (option "build.ploc_radius"
 * selects clustering instead of the radix tree, "build.optimize_passes" improves either).  Two orders of magnitude faster
 * than pt_bvh_build + pt_scene_create on large meshes; the tree is of lower quality than the
 * SAH builder's (more boxes per ray).  The result is deterministic. */
int pt_scene_build(pt_context *ctx, const float *vtx, uint32_t n_vtx, const int32_t *idx,
                   uint32_t n_tri, const pt_material *materials, uint32_t n_materials,
                   const int32_t *material_ids, uint32_t link_encoding, pt_scene **out);
/* ---- two-level scenes: meshes placed by affine transforms ---------------------------------
 * The format the reference documents for a "top-level BVH" whose leaf payload is an instance
 * index (src/bvh/bvh_builder.h:29-33) and never builds (its two-level translator is disabled,
 * src/bvh/RadeonRays/plain_bvh_translator.cpp:98-101,134-137).  Every tree comes from
 * pt_bvh_build, like the reference's single one: one per mesh over its triangles' boxes, one
 * over the instances' world boxes from pt_instance_bounds.
 *
 * Semantics.  The top level is walked like IntersectSceneClosest walks its nodes
 * (shaders/RadeonRays/bvh.glslh:431-455) with the world-space ray; at a leaf the ray goes to
 * the instance's object space (o' = W o + w, d' = W d in binary32, ((a+b)+c)+d order, W|w the
 * inverse of object_to_world computed in binary64 and rounded once) and the mesh is searched
 * with the running nearest distance as limit; d' is not renormalised, so distances are the
 * same in both spaces.  A hit reports shapeid = instance index and primid = triangle of the
 * instance's mesh.  Shading uses the world-space vertices M v (same operation order). */
typedef struct pt_mesh_desc {
    const pt_bbox *nodes;        /* the mesh's skip-link BVH, as for pt_scene_create; NULL (with
                                    n_nodes 0) to have it built on the GPU, as pt_scene_build does */
    const float *vertices;       /* xyz packed, object space */
    const int32_t *indices;      /* 3 per triangle */
    const int32_t *material_ids; /* per triangle, or NULL (all 0) */
    uint32_t n_nodes, n_vertices, n_triangles, _pad;
} pt_mesh_desc;
typedef struct pt_instance {
    float object_to_world[12];   /* 3 rows of (linear part | translation) */
    uint32_t mesh;
    int32_t material_offset;     /* added to the mesh's material ids */
} pt_instance;
/* World-space box of every instance (the eight corners of its mesh's root box, transformed):
 * the leaf boxes of the top-level build.  Host only. */
int pt_instance_bounds(const pt_mesh_desc *meshes, uint32_t n_meshes, const pt_instance *instances,
                       uint32_t n_instances, pt_bbox *out);
/* top_nodes: skip-link tree over the instances (leaf payload = instance index), all trees in
 * the same link_encoding; NULL (with n_top_nodes 0) to have it built here, by the builder of
 * pt_bvh_build (SAH, two optimisation passes) over the boxes of pt_instance_bounds.  materials may
 * be NULL (one grey diffuse material).  The scene works with every pt_trace_* and pt_render
 * call (PT_TRACE_MOLLER_TRUMBORE excepted). */
int pt_scene_create_instanced(pt_context *ctx, const pt_mesh_desc *meshes, uint32_t n_meshes,
                              const pt_instance *instances, uint32_t n_instances,
                              const pt_bbox *top_nodes, uint32_t n_top_nodes,
                              uint32_t link_encoding, const pt_material *materials,
                              uint32_t n_materials, pt_scene **out);
uint32_t pt_scene_instance_count(const pt_scene *scene);   /* 0 for a single-level scene */
/* A glTF 2.0 / GLB scene kept as it is structured (pt_gltf_load flattens it instead): every
 * mesh that a node shows, once, in its own space (all its triangle-list primitives), and one
 * instance per showing node with the node hierarchy composed into object_to_world -- the
 * arrays pt_scene_create_instanced takes (material_offset 0: the meshes' material ids index
 * pt_model_materials).  Meshes are owned by the model; read them with pt_mesh_*. */
int pt_gltf_load_model(const char *path, pt_model **out);
uint32_t pt_model_mesh_count(const pt_model *m);
const pt_mesh *pt_model_mesh(const pt_model *m, uint32_t i);
uint32_t pt_model_instance_count(const pt_model *m);
const pt_instance *pt_model_instances(const pt_model *m);
uint32_t pt_model_material_count(const pt_model *m);
const pt_material *pt_model_materials(const pt_model *m);
int pt_model_destroy(pt_model *m);
/* The trees of a two-level scene, given or built (for validation, caching, or a CPU walk of the
 * same trees).  Blocking. */
uint32_t pt_scene_top_node_count(const pt_scene *scene);
int pt_scene_read_top_nodes(pt_scene *scene, pt_bbox *nodes);
uint32_t pt_scene_mesh_node_count(const pt_scene *scene, uint32_t mesh);
int pt_scene_read_mesh_nodes(pt_scene *scene, uint32_t mesh, pt_bbox *nodes);

/* Several meshes flattened into one scene: what a hit on each triangle reports.  The
 * reference's multi-mesh builder writes one Face{idx[3], shape_id, prim_id} per triangle
 * (src/bvh/RadeonRays/intersector_skip_links.h:82-92, filled at intersector_skip_links.cpp:236-264)
 * for the shader to return as isect.shapeid / isect.primid (shaders/RadeonRays/bvh.glslh:342-343,
 * commented out in the checked-in shader, which returns 0 and the triangle index).  With a
 * table set, pt_trace_closest* report shapeid = shape_ids[t], primid = prim_ids[t] for a hit
 * on triangle t of the scene's index buffer; misses are unchanged.  n_triangles entries each,
 * which must be the scene's triangle count (pt_scene_triangle_count; anything else is
 * PT_ERR_INVALID); NULL, NULL removes the table.  pt_render and its statistics are not affected.
 * Single-level scenes only (a two-level scene reports instance and mesh triangle).  Blocking. */
int pt_scene_set_face_ids(pt_scene *scene, const int32_t *shape_ids, const int32_t *prim_ids,
                          uint32_t n_triangles);
/* Triangles of the scene's index buffer (of all instances, for a two-level scene). */
uint32_t pt_scene_triangle_count(const pt_scene *scene);

/* The scene's skip-link array (what pt_scene_create was given, or what pt_scene_build made):
 * for validation, caching, or a CPU traversal of the same tree.  Blocking. */
uint32_t pt_scene_node_count(const pt_scene *scene);
uint32_t pt_scene_link_encoding(const pt_scene *scene);
/* Levels of inner nodes below the root of a single-level scene's tree (what the per-scene choice
 * of "extend.stack_entries" / "extend.cache_nodes" goes by). */
uint32_t pt_scene_tree_depth(const pt_scene *scene);
int pt_scene_read_nodes(pt_scene *scene, pt_bbox *nodes);
int pt_scene_destroy(pt_scene *scene);

/* Most rays one pt_trace_* call takes. */
#define PT_TRACE_MAX_RAYS 0x7FFFFFC0ull   /* 2^31 - 64 */
/* IntersectSceneClosest, shaders/RadeonRays/bvh.glslh:417-456, for n rays.
 * flags: PT_TRACE_*.  stats may be NULL. */
int pt_trace_closest(pt_context *ctx, const pt_scene *scene, const pt_ray *rays, uint64_t n,
                     pt_hit *hits, uint32_t flags, pt_trace_stats *stats);
int pt_trace_closest_device(pt_context *ctx, const pt_scene *scene, const pt_ray *rays_dev,
                            uint64_t n, pt_hit *hits_dev, uint32_t flags);
/* As pt_trace_closest_device, also counting slab and triangle tests (slower; for
 * measurement).  stats is a host pointer and the call synchronizes. */
int pt_trace_closest_device_counted(pt_context *ctx, const pt_scene *scene,
                                    const pt_ray *rays_dev, uint64_t n, pt_hit *hits_dev,
                                    uint32_t flags, pt_trace_stats *stats);

/* pt_trace_closest_device for callers that cannot allocate device memory (code under HIP
 * graph capture, serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Same hits, bit for bit; what the kernels need beyond rays_dev and hits_dev (a
 * counter block and the global slots behind the per-lane LDS stacks) lies in a static device
 * array of PT_STATIC_SCRATCH_BYTES that every GPU the library runs on holds until the process
 * ends, and the call makes no hipMalloc / hipFree.  For that the persistent grid is limited
 * to 1280 workgroups with 16 global stack slots per lane (a ray with more than 46 pending
 * nodes takes the fixed-order walk: slower, same answer) and the rays are always taken in
 * queue order (option "trace.sort_rays" is ignored).  Fails with PT_ERR_INVALID above
 * PT_TRACE_MAX_RAYS rays.  All calls on one GPU share the array: calls that overlap on one
 * GPU (two contexts or two streams) race and must be serialised by the caller. */
#define PT_STATIC_SCRATCH_BYTES (256u + 1280u * 256u * 16u * 4u)
int pt_trace_closest_device_static(pt_context *ctx, const pt_scene *scene, const pt_ray *rays_dev,
                                   uint64_t n, pt_hit *hits_dev, uint32_t flags);

/* IntersectSceneAny, shaders/RadeonRays/bvh.glslh:378-414: occluded[i] = 1 or 0. */
int pt_trace_any(pt_context *ctx, const pt_scene *scene, const pt_ray *rays, uint64_t n,
                 int32_t *occluded);
int pt_trace_any_device(pt_context *ctx, const pt_scene *scene, const pt_ray *rays_dev,
                        uint64_t n, int32_t *occluded_dev);

/* pt_camera from what the reference's host side holds a camera in.  Arithmetic in binary64,
 * rounded once; PT_ERR_INVALID for values that are not finite or describe no camera.
 *
 * pt_camera_from_matrices: the fields of Granite's `global` uniform block the shader reads --
 * mat3(inv_view), inv_projection[0][0] and [1][1], camera_position (shaders/depth.frag:59-60,
 * 70,74; bound by renderer_.bind_global_parameters, src/main_granite.cpp:432-433).  Both
 * matrices are 16 floats, column-major as GLSL stores a mat4; camera_position may be NULL
 * (the fourth column of inv_view is taken).
 * pt_camera_from_orientation: the reference's own Camera struct, src/camera.h:11-24
 * (orientation quaternion x,y,z,w taking camera axes to world axes, forward = orientation *
 * (0,0,-1); eye_pos; viewport_size), with the vertical field of view the legacy application
 * turns into image_size_norm = 2 tan(fovy/2) * (aspect, 1), src/main.cpp:1304-1309.
 * pt_camera_look_at: FPSCamera::look_at + set_depth_range/fovy of the live application,
 * src/main_granite.cpp:306-308; up may be NULL for (0,1,0). */
int pt_camera_from_matrices(const float *inv_view, const float *inv_projection,
                            const float *camera_position, uint32_t mode, pt_camera *out);
int pt_camera_from_orientation(const float *orientation_xyzw, const float *eye_pos, float fovy_radians,
                               uint32_t viewport_width, uint32_t viewport_height, uint32_t mode,
                               pt_camera *out);
int pt_camera_look_at(const float *eye, const float *target, const float *up, float fovy_radians,
                      float aspect, uint32_t mode, pt_camera *out);

/* Camera rays for every pixel of a width x height image, row-major from the top-left
 * pixel: shaders/depth.frag:47-83 with the pixel centre mapping of shaders/quad.vert:6,14
 * (PT_CAMERA_REFERENCE), or a plain pinhole (PT_CAMERA_PINHOLE). */
int pt_raygen(pt_context *ctx, const pt_camera *cam, uint32_t width, uint32_t height,
              pt_ray *rays);
int pt_raygen_device(pt_context *ctx, const pt_camera *cam, uint32_t width, uint32_t height,
                     pt_ray *rays_dev);

/* The colour the reference writes per pixel, shaders/depth.frag:90-107. rgba: n*4. */
int pt_depth_shade(pt_context *ctx, const pt_hit *hits, uint64_t n, float *rgba);

/* ------------------------------------------------------------------------ */
/* Film and rendering                                                        */

/* The colour attachment of the reference's "path_trace" pass (src/main_granite.cpp:388-400),
 * as width*height float4 in row-major order from the top-left pixel: rgb = radiance summed
 * over samples, a = number of samples (the caller divides).  In PT_RENDER_DEPTH_GREY mode it
 * holds the reference's (grey, grey, grey, 1). */
int pt_film_create(pt_context *ctx, uint32_t width, uint32_t height, pt_film **out);
/* Same, over device memory the caller owns (width*height*4 floats, 16-byte aligned), e.g. a
 * buffer that is also handed to RCCL. */
int pt_film_wrap(pt_context *ctx, uint32_t width, uint32_t height, float *rgba_dev,
                 pt_film **out);
int pt_film_clear(pt_film *film);
float *pt_film_device_ptr(pt_film *film);
/* readback_ssbo, src/main_granite.cpp:499-522: blocking copy to host memory. */
int pt_film_read(pt_film *film, float *rgba);
int pt_film_destroy(pt_film *film);

/* Files.  No counterpart in the reference, which rebuilds its BVH at every start
 * (src/main_granite.cpp:289-293) and keeps no accumulation; the usage pattern is that of its
 * Vulkan pipeline cache (src/main.cpp:290-315,1187-1196): load if present and valid, else
 * build and save. */
/* Key of a build's inputs (leaf boxes + options) for validating a cached BVH. */
uint64_t pt_bvh_cache_key(const pt_bbox *leaf_bounds, uint32_t n_leaves, const pt_bvh_options *opt);
int pt_bvh_save(const pt_bvh *bvh, const char *path, uint64_t key);
/* PT_ERR_IO: unreadable or corrupt; PT_ERR_STALE: intact but saved under another key. */
int pt_bvh_load(const char *path, uint64_t key, pt_bvh **out);
/* Accumulation checkpoint: the film's sums and sample counts, exactly (blocking). */
int pt_film_save(pt_film *film, const char *path);
int pt_film_load(pt_film *film, const char *path);
#define PT_IMAGE_PPM 0u   /* 8-bit binary PPM, sRGB transfer curve, clamped */
#define PT_IMAGE_PFM 1u   /* 32-bit float PFM, linear */
#define PT_IMAGE_PNG 2u   /* 8-bit RGB PNG (uncompressed data stream), sRGB transfer curve */
/* Tone curve for the 8-bit formats, OR-ed into format; without one values are clamped at 1.
 * (Image output is on the reference's to-do list only, src/main_granite.cpp:285,389.) */
#define PT_IMAGE_TONEMAP_REINHARD 0x100u   /* L / (1 + L) on luminance */
#define PT_IMAGE_TONEMAP_FILMIC   0x200u   /* per-channel rational fit of the ACES curve */
/* Writes rgb / sample count, scaled by exposure. */
int pt_film_write_image(pt_film *film, const char *path, uint32_t format, float exposure);

/* Host-only description of a tile sharding: *n_slots = pixel slots of the shard (owned tiles
 * x tile_size^2, the length of its packed buffer); if xy != NULL it receives, per slot, the
 * pixel's x and y, or (-1,-1) for a slot outside the image.  Needs no GPU. */
int pt_shard_layout(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t shard_index,
                    uint32_t shard_count, uint64_t *n_slots, int32_t *xy);

/* Number of float4 entries pt_film_pack_tiles writes for this sharding: tiles owned by
 * shard_index (tile t belongs to shard t % shard_count) times tile_size^2. */
uint64_t pt_film_packed_size(const pt_film *film, uint32_t tile_size, uint32_t shard_index,
                             uint32_t shard_count);
/* Copies the shard's tiles into / out of a dense device buffer, so that ranks can exchange
 * only the tiles they rendered (one RCCL all-gather) instead of whole frames. */
int pt_film_pack_tiles(pt_film *film, uint32_t tile_size, uint32_t shard_index,
                       uint32_t shard_count, float *packed_dev);
int pt_film_unpack_tiles(pt_film *film, uint32_t tile_size, uint32_t shard_index,
                         uint32_t shard_count, const float *packed_dev);

/* ------------------------------------------------------------------------ */
/* Multi-GPU frame assembly over RCCL.  The reference is single-device and reads its image
 * back with readback_ssbo (src/main_granite.cpp:499-522); with one context per GPU, scene
 * replicated and the frame sharded by pt_render_params (tiles or sample ranges), these calls
 * bring the shards together on the context's stream before that read-back.  RCCL is loaded
 * at run time (librccl.so.1) on first use; PT_ERR_UNSUPPORTED if it cannot be found. */
typedef struct pt_comm pt_comm;
#define PT_COMM_ID_BYTES 128
int pt_comm_available(void);                         /* 1 if RCCL could be loaded */
/* ncclGetUniqueId: call on one rank, hand the 128 bytes to all others (any transport). */
int pt_comm_unique_id(void *id);
/* ncclCommInitRank on the context's GPU: one process (or thread) per GPU; collective, every
 * rank calls it with the same id and n_ranks. */
int pt_comm_create(pt_context *ctx, const void *id, int n_ranks, int rank, pt_comm **out);
/* ncclCommInitAll: one process driving n GPUs, one context each; out receives n handles, rank
 * i on ctxs[i].  Collective calls below are then made for every handle, each from the thread
 * that drives its context. */
int pt_comm_create_all(pt_context *const *ctxs, int n, pt_comm **out);
int pt_comm_rank(const pt_comm *comm);
int pt_comm_size(const pt_comm *comm);
int pt_comm_destroy(pt_comm *comm);
/* Sample sharding (every rank rendered its own sample range of the whole frame): the films
 * -- radiance sums and sample counts -- are added up, in place, on rank `root` (ncclReduce;
 * other ranks' films are left as they are) or on every rank when root < 0 (ncclAllReduce).
 * Asynchronous on the context's stream; pt_film_read orders after it. */
int pt_film_reduce(pt_film *film, pt_comm *comm, int root);
/* Tile sharding (rank r rendered the tiles t with t % size == r, pt_render_params.shard_*):
 * every peer packs its tiles and sends them straight to `root` (grouped ncclSend/ncclRecv:
 * each slice crosses only the link between its owner and the root), where they are scattered
 * into the film; root < 0: every rank receives every slice (ncclAllGather).  The assembled
 * frame equals a single-GPU render bit for bit.  Asynchronous on the context's stream. */
int pt_film_gather_tiles(pt_film *film, pt_comm *comm, uint32_t tile_size, int root);

/* One frame's worth of work of the "path_trace" pass callback, src/main_granite.cpp:402-451
 * (bind the scene buffers and the camera, draw): renders params->sample_count samples per
 * pixel of the tiles this shard owns and adds them to the film.  Asynchronous on the
 * context's stream unless stats != NULL (then it synchronizes and fills stats). */
int pt_render(pt_context *ctx, const pt_scene *scene, const pt_camera *cam,
              const pt_render_params *params, pt_film *film, pt_render_stats *stats);
void pt_render_params_default(pt_render_params *params, uint32_t width, uint32_t height);

#ifdef __cplusplus
}
#endif
#endif
