/*
 * pt_config.h -- configuration shared by host code and HIP device code.
 *
 * Plays the role of the reference's shaders/config-inc.h:1-3 (one header
 * included by both the C++ host, src/main_granite.cpp:34, and the shader,
 * shaders/depth.frag:5).  The reference's only entry, BVH_SET_BINDING, is a
 * Vulkan descriptor-set index and has no meaning for HIP; it is kept so that
 * host code written against the reference header still compiles.
 *
 * Plain C: usable from C, C++ and HIP translation units alike.
 */
#ifndef PT_CONFIG_H
#define PT_CONFIG_H

#include <stdint.h>

#define BVH_SET_BINDING 0 /* shaders/config-inc.h:3 */

/* ---- traversal ---------------------------------------------------------- */
/* Relative width of the near-tie window of the ordered traversal: two hits
 * whose distances differ by less than this fraction are re-resolved in the
 * reference's fixed skip-link order.  2^-18. */
#define PT_TIE_WINDOW 3.814697265625e-06f
/* Per-lane entries of the LDS traversal stack of the one-thread-per-ray kernels, and the most a
 * persistent kernel may be given: 31 x 4 B x 256 lanes with the sentinel, so that five
 * workgroups fit the 160 KB of a compute unit. */
#ifndef PT_LDS_STACK_ENTRIES
#define PT_LDS_STACK_ENTRIES 30
#endif
/* Most two-child nodes from the top of the tree (breadth first from the root) that a scene
 * prepares for the persistent traversal kernels to keep in LDS, so that the steps every ray
 * takes first do not go through the vector memory pipeline.  How many a launch really keeps,
 * and how many stack entries, is chosen per launch (options "extend.cache_nodes",
 * "extend.stack_entries"). */
#ifndef PT_LDS_NODE_CACHE_MAX
#define PT_LDS_NODE_CACHE_MAX 448
#endif
/* Deepest tree (levels) for which the persistent kernels trade stack entries for cached nodes
 * by default. */
#ifndef PT_CACHE_MAX_TREE_DEPTH
#define PT_CACHE_MAX_TREE_DEPTH 40
#endif
/* Lanes of a wave with a ray waiting for the fixed-order walk that start the walk for all. */
#ifndef PT_TIE_BATCH
#define PT_TIE_BATCH 16
#endif
/* Size of a traversal triangle record in 16-byte units: 3 (the 40 bytes in use, packed) or 4 (one
 * record per 64 bytes, so that none lies across two of the 64-byte pieces memory is fetched in). */
#ifndef PT_TRI_RECORD_16B
#define PT_TRI_RECORD_16B 3
#endif
/* Threads per workgroup of the traversal kernels (a multiple of 64). */
#ifndef PT_TRACE_BLOCK
#define PT_TRACE_BLOCK 256
#endif

/* ---- camera modes -------------------------------------------------------- */
/* Reproduces shaders/depth.frag:47-83 literally, including its debug
 * arithmetic (origin.y += 2, origin and tmax divided by 1/20, half-width
 * field of view). */
#define PT_CAMERA_REFERENCE 0u
/* Plain pinhole: full field of view, optional sub-pixel jitter, tmax 1e30. */
#define PT_CAMERA_PINHOLE 1u

/* ---- trace flags --------------------------------------------------------- */
#define PT_TRACE_DEFAULT      0u
/* Visit nodes in the reference's skip-link order for every ray. */
#define PT_TRACE_EXACT_ORDER  1u
/* Use the Moller-Trumbore closest-hit test (shaders/RadeonRays/bvh.glslh:181-202)
 * instead of the watertight one; implies PT_TRACE_EXACT_ORDER. */
#define PT_TRACE_MOLLER_TRUMBORE 2u
/* Ordered traversal only: skip the exact-order re-resolution of near ties. */
#define PT_TRACE_NO_TIE_FIXUP 4u

/* ---- render flags -------------------------------------------------------- */
#define PT_RENDER_DEFAULT 0u
/* Only the primary hit: write the reference's depth-grey debug colour
 * (shaders/depth.frag:90-107) instead of radiance. */
#define PT_RENDER_DEPTH_GREY 1u
/* Add next-event estimation (one shadow ray per path vertex). */
#define PT_RENDER_NEE 2u
/* Camera samples at pixel centres instead of jittered positions. */
#define PT_RENDER_PIXEL_CENTRE 4u
/* Count slab and triangle tests of the closest-hit kernels into pt_render_stats (slower;
 * for measurement). */
#define PT_RENDER_COUNT_WORK 8u

#ifdef __cplusplus
extern "C" {
#endif

/* Camera as the shader consumes it (shaders/depth.frag:59-60,70,74):
 * c0,c1,c2 are the columns of mat3(inv_view); sx, sy the two diagonal
 * inverse-projection entries; eye the camera position. */
typedef struct pt_camera {
    float eye[3]; float sx;
    float c0[3];  float sy;
    float c1[3];  uint32_t mode;
    float c2[3];  float _pad;
} pt_camera;

/* One surface description per material id.  48 bytes. */
typedef struct pt_material {
    float kd[3]; float roughness;   /* Lambert albedo; GGX alpha = roughness^2 */
    float ks[3]; float _pad0;       /* specular reflectance at normal incidence */
    float ke[3]; float _pad1;       /* emitted radiance */
} pt_material;

/* Parameters of one pt_render call. */
typedef struct pt_render_params {
    uint32_t width, height;         /* full image size */
    uint32_t sample_begin;          /* first sample index of this call */
    uint32_t sample_count;          /* samples per pixel rendered by this call */
    uint32_t max_depth;             /* path vertices after the camera (>= 1) */
    uint32_t seed;
    uint32_t rr_start_depth;        /* Russian roulette from this bounce on */
    uint32_t flags;                 /* PT_RENDER_* */
    uint32_t tile_size;             /* tile sharding: edge of a square tile, 0 = whole image */
    uint32_t shard_index;           /* this call renders tiles t with t % shard_count == shard_index */
    uint32_t shard_count;
    uint32_t samples_per_pass;      /* samples in flight together, 0 = library default */
    float background[3];            /* radiance of a ray that leaves the scene */
    float ray_offset;               /* origin offset along the normal, relative */
} pt_render_params;

#ifdef __cplusplus
}
#endif
#endif
