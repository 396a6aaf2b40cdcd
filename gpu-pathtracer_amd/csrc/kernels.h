// Internal launch interface between the C ABI (api.cpp) and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>

#include "../../include/mipt.h"

namespace mipt {

struct DeviceBuffer;

// One placed copy of a mesh in a two-level scene (80 bytes, read as five 16-byte words).
struct DeviceInstance {
    float inv[12];        // world -> object, 3 rows of (xyz | translation)
    // Where the instance's mesh starts in the scene-wide arrays: skip-link nodes, two-child
    // nodes, traversal triangle records, vertices, triangles (of idx); then its triangle
    // count and the index of its first triangle in the world-space shading records.
    uint32_t node_off, wnode_off, wtri_off, vtx_off;
    uint32_t tri_off, n_tri, first_tri;
    float cmax;           // largest coordinate of the mesh's root box, a little generous (four-child walk)
};

// Device-resident scene, passed to kernels by value.
struct DeviceScene {
    // The three buffers the reference shader binds (shaders/RadeonRays/bvh.glslh:103-117).
    const float4 *nodes;   // 2 x float4 per skip-link node
    const float *vtx;      // xyz packed
    const int32_t *idx;    // 3 per triangle
    // Re-laid-out copies for the ordered traversal.
    const float4 *wnodes;  // 4 x float4 per two-child node (64 B)
    const float4 *wtris;   // 3 x float4 per triangle record (48 B), leaf order
    // For shading: the same 48-byte record per triangle in the caller's triangle order, with
    // the material id where the traversal record keeps the triangle id.
    const float4 *stris;
    const int32_t *mat_ids;
    const pt_material *materials;
    // Emissive triangles (ascending) and the running sum of their areas.
    const int32_t *light_tri;
    const float *light_cdf;
    uint32_t n_lights;
    float light_area;
    uint32_t n_nodes, n_tri, n_wnodes, n_materials;
    // two-level scenes: traversal triangle records of all meshes together (n_tri counts the
    // instanced triangles there); 0 in single-level scenes, where every leaf has one
    uint32_t n_mesh_records;
    uint32_t link_enc;
    uint32_t wide_depth;
    uint32_t big;          // wnodes or wtris reach beyond 4 GiB: records need 64-bit addresses
    // The first n_top_cache two-child nodes in breadth-first order from the root (of a two-level
    // scene: from the top-level root, wnodes[top_root]), as the persistent kernels copy them to LDS.  A child that lies in this array itself
    // is named kLdsNodeBase + its position here; other children as in wnodes.
    const float4 *top_cache;
    const int32_t *top_global;   // index in wnodes of each node of top_cache
    uint32_t n_top_cache;
    // Single-level scenes in which every triangle has exactly one leaf: the index in `nodes` of
    // each triangle's leaf (null otherwise).  Lets a ray with two near-equal candidates be
    // decided at those two leaves instead of by a walk over the whole tree.
    const int32_t *tri_leaf;
    // The same for single-level scenes in which triangles have several leaves (spatial splits):
    // the leaves of triangle t are tri_leaf_list[tri_leaf_first[t] .. tri_leaf_first[t + 1]),
    // ascending.  Null where tri_leaf is set.
    const uint32_t *tri_leaf_first;
    const int32_t *tri_leaf_list;
    // The same tree with two levels folded into one (of a two-level scene: the meshes' trees; the
    // top-level tree has the two-child form only),
    // as 64-byte four-child nodes whose child boxes are 8-bit offsets on a per-node grid (never
    // inside the exact boxes), and 64-byte triangle records that carry the leaf's exact box.
    //   qnodes[4 i .. 4 i + 3], i = index of the two-child node the record was folded from:
    //     (origin.xyz, scale.x) (scale.y, scale.z, lo.x of 4 children, lo.y x4)
    //     (lo.z x4, hi.x x4, hi.y x4, hi.z x4) (child 0..3)
    //   a child is an index into qnodes (>= 0), ~(position of a qtris record in 16-byte units),
    //   or kNoChild; child k's box is origin + byte k * scale per component.
    //   qtris[4 j .. 4 j + 3]: (v0.xyz v1.x) (v1.yz v2.xy) (v2.z, triangle, lo.xy) (lo.z, hi.xyz)
    // Half as many node records per ray, for more arithmetic per record: what pays once the
    // records come from memory rather than from the caches.  Null when not built.
    const float4 *qnodes;
    const float4 *qtris;
    const float4 *q_top_cache;      // as top_cache / top_global, over qnodes
    const int32_t *q_top_global;
    uint32_t n_q_top_cache;
    uint32_t quad_depth;            // levels of four-child nodes
    // Optional: what a hit on each triangle reports as (shapeid, primid) from the stand-alone
    // closest-hit queries, the Face records of the reference's multi-mesh builder
    // (src/bvh/RadeonRays/intersector_skip_links.h:82-92).  Null: (0, triangle index).
    const int2 *face_ids;
    // Two-level scenes only (tlas != nullptr): a skip-link tree over the instances whose leaf
    // payload is an instance index (the "top-level BVH" of src/bvh/bvh_builder.h:31); the
    // arrays above then hold every mesh back to back, in object space, except stris, which
    // holds one world-space record per instanced triangle.
    const float4 *tlas;
    // the same top-level tree as two-child nodes, stored behind the meshes' nodes in wnodes (its
    // root is wnodes[top_root], inner children are indices into wnodes); leaf child = ~instance
    const float4 *wtop;
    uint32_t top_root;
    const DeviceInstance *instances;
    uint32_t n_tlas_nodes, n_instances;
};

constexpr int32_t kLdsNodeBase = 1 << 30;
constexpr int32_t kNoChild = INT32_MIN;      // an unused child slot of a four-child node
// What the child encodings can name: two-child nodes below kLdsNodeBase, triangle records by
// their position in 16-byte units (three per record) in 31 bits.
constexpr uint64_t kMaxWideNodes = (1ull << 30) - 1,
                   kMaxWideTris = 2100000000ull / PT_TRI_RECORD_16B;     // 700 M packed records
inline bool needs_big_addresses(uint64_t n_wnodes, uint64_t n_wtris) {
    // byte offsets of the records must fit 32 bits
    return n_wnodes > (1ull << 26) || n_wtris > 267000000ull / PT_TRI_RECORD_16B;
}
constexpr int kNodeCacheMax = PT_LDS_NODE_CACHE_MAX;
// Fills top_cache and top_global (room for kNodeCacheMax nodes each) from wnodes and writes the
// number of nodes to *n_out.
// root: the node to start from (0; the top-level root of a two-level scene).
void launch_build_top_cache(hipStream_t st, const float4 *wnodes, uint32_t n_wnodes, float4 *cache,
                            int32_t *global_index, uint32_t *n_out, uint32_t root = 0);
// Folds the two-child nodes into four-child nodes with quantised child boxes and writes the
// triangle records with their leaves' boxes (DeviceScene::qnodes / qtris, room for n_wnodes and
// n_tri records of 64 bytes).  *bad becomes non-zero if some box does not quantise (a bound
// that is not finite).
void launch_build_quad(hipStream_t st, const float4 *wnodes, uint32_t n_wnodes, const float4 *wtris,
                       uint32_t n_tri, float4 *qnodes, float4 *qtris, uint32_t *bad);
// launch_build_top_cache over four-child nodes.
void launch_build_quad_top_cache(hipStream_t st, const float4 *qnodes, uint32_t n_wnodes, float4 *cache,
                                 int32_t *global_index, uint32_t *n_out);
// Fills tri_leaf[0..n_tri) (preset to -1 by the caller) with the index of each triangle's leaf
// among the n_nodes skip-link nodes; *multi becomes non-zero if some triangle has several.
// Rewrites (shapeid, primid) of every hit record in hits[0..n) through face_ids.
void launch_map_face_ids(hipStream_t st, pt_hit *hits, uint64_t n, const int2 *face_ids);
void launch_build_tri_leaf(hipStream_t st, const float4 *nodes, uint32_t n_nodes, uint32_t link_enc,
                           uint32_t n_tri, int32_t *tri_leaf, uint32_t *multi);
// The form for triangles with several leaves: first (n_tri + 1 words) and list (room for n_nodes
// words, of which the first (n_nodes + 1) / 2 are the list) as in DeviceScene.  Every triangle
// must have a leaf.  Uses `scratch`; blocks until done.  Returns a HIP error or hipSuccess.
hipError_t build_tri_leaf_lists(hipStream_t st, const float4 *nodes, uint32_t n_nodes, uint32_t link_enc,
                                uint32_t n_tri, uint32_t *first, int32_t *list, DeviceBuffer &scratch);

// What a persistent traversal kernel is told about its own loop.
struct PersistentParams {
    int refill, leaf_lanes;
    int32_t *spill;
    int spill_entries;
    int stack_cap;      // per-lane LDS stack entries (without the bottom sentinel)
    int cache_nodes;    // nodes of the scene's top_cache a workgroup keeps in LDS
    int xcd_slots;      // deal every round's chunks to the XCDs in contiguous eighths
    // Rays a wave still holds tail_budget rounds after its stream has ended go to this list
    // (closest hits by the four-child kernels on single-level scenes; null or 0: every wave
    // finishes its own rays):
    // long_list[0] entries appended, [1] waves of the second kernel that are done, then from
    // kLongListHeader on up to long_cap ray indices.
    uint32_t *long_list;
    uint32_t long_cap;
    int tail_budget;
};
constexpr uint32_t kLongListHeader = 4;
constexpr uint32_t kLongListCap = 1u << 16;

// Device counters of a trace call (mirrors pt_trace_stats, device side).
struct DeviceTraceCounters {
    unsigned long long box_tests;
    unsigned long long tri_tests;
    unsigned int tie_count;   // entries appended to the tie list
    unsigned int long_rays;   // rays handed to the wave-per-ray search (see PersistentParams::long_list)
    unsigned long long lds_box_tests;   // of box_tests: on nodes read from the LDS copy of the top of the tree
};

// How the closest-hit kernels are launched.
struct ExtendConfig {
    int device;             // HIP device the launch goes to
    int persistent_blocks;  // > 0: persistent waves; 0: one thread per ray
    int num_cus;            // compute units of the device
    int blocks_per_cu;      // persistent form: upper bound on workgroups per compute unit
    int refill;             // persistent form: top up a wave when this many lanes are idle
    int leaf_lanes;         // persistent form: triangle step once this many lanes wait at one
    int sort_rays;          // trace calls, persistent form: take the rays in origin-cell order
    int sort_origin_bits;   // bits per axis of the origin cell in the sort key (<= 10)
    int sort_dir_bits;      // bits per octahedral coordinate of the direction in the key (0: none)
    // persistent form: global memory for the stack entries of rays that need more than the LDS
    // holds, spill_entries slots for each lane of the largest grid (kSpillEntries, or 0)
    int32_t *spill;
    int spill_entries;
    // persistent form: per-lane LDS stack entries, and how many two-child nodes from the top of
    // the tree every workgroup keeps in LDS (single-level scenes; 0: none)
    int stack_entries;
    int cache_nodes;
    int xcd_slots;          // see PersistentParams
    int quad;               // persistent closest-hit form: walk the four-child nodes (DeviceScene::qnodes)
    // persistent form: upper bound on the grid in workgroups (0: none), for a spill area that
    // was sized for fewer lanes than the device could keep resident
    unsigned int max_grid;
    // persistent closest-hit form, single-level scenes: see PersistentParams::long_list
    uint32_t *long_list;
    int tail_budget;
};

// Stack entries per lane beyond the LDS ones: a ray stays in the voted loop up to about
// PT_LDS_STACK_ENTRIES + kSpillEntries pending nodes before it is left to the fixed-order walk.
constexpr int kSpillEntries = 48;
constexpr int kMaxBlocksPerCu = 8;      // upper end of option "extend.blocks_per_cu"
constexpr unsigned int kLdsBudgetPerCu = 155u * 1024u;   // LDS that resident workgroups may add up to
constexpr int kMaxLanesPerCu = 2048;    // 32 resident waves: what the spill area is sized for

// Closest hit in the reference's fixed skip-link order.
void launch_trace_skiplink(hipStream_t st, const DeviceScene &s, const pt_ray *rays, pt_hit *hits,
                           uint64_t n, bool moller_trumbore, DeviceTraceCounters *counters);

// bracket, if given: two events recorded on the stream right before and right after the
// traversal kernel itself (after any sorting of the rays).
// Closest hit by ordered two-child traversal.  Rays whose result could depend on the
// visiting order are counted in counters->tie_count and, when fixup is set, re-resolved
// in the fixed order inside the same kernel.
void launch_trace_ordered(hipStream_t st, const DeviceScene &s, const pt_ray *rays, pt_hit *hits,
                          uint64_t n, bool fixup, DeviceTraceCounters *counters,
                          bool count_work, const ExtendConfig &cfg, DeviceBuffer *sort_scratch,
                          hipEvent_t *bracket = nullptr);

// Closest hit in a two-level scene (s.tlas set).  exact: meshes in the fixed skip-link order.
// counters must be non-null (its tie_count is always written).
void launch_trace_two_level(hipStream_t st, const DeviceScene &s, const pt_ray *rays, pt_hit *hits,
                            uint64_t n, bool fixup, bool exact, DeviceTraceCounters *counters,
                            bool count_work, const ExtendConfig &cfg);

// Fills s.stris (one world-space record per instanced triangle) from the object-space meshes:
// object_to_world holds 3 float4 rows per instance, mesh_mat_ids one id per mesh triangle in
// the order of s.idx (or null), material_offset one value per instance.
void launch_instance_shading_records(hipStream_t st, const DeviceScene &s,
                                     const float4 *object_to_world, const int32_t *mesh_mat_ids,
                                     const int32_t *material_offset, uint32_t max_mesh_tris,
                                     float4 *stris);

// Makes the child indices of n_nodes two-child nodes scene-wide: the mesh's nodes start at
// node_base, its triangle records at tri_record_base (in records).
void launch_rebase_children(hipStream_t st, float4 *wnodes, uint32_t n_nodes, uint32_t node_base,
                            uint32_t tri_record_base);

// Workgroups of `block` threads a persistent kernel may launch and still have them all
// resident: compute units x min(cap, what the occupancy query admits).
// lds_bytes: the workgroup's dynamic LDS (persistent_lds_bytes).
unsigned int persistent_grid(const void *kernel, int block, const ExtendConfig &cfg,
                             unsigned int lds_bytes);
// LDS of one workgroup of a persistent traversal kernel (passed as dynamic shared memory): the
// lanes' stacks with their bottom sentinels, then the cached top of the tree.
inline unsigned int persistent_lds_bytes(const PersistentParams &pp) {
    return (unsigned int)((pp.stack_cap + 1) * PT_TRACE_BLOCK * 4 + 64 * pp.cache_nodes);
}
// The loop parameters of a launch on scene s: the cache never holds more nodes than the scene
// has prepared (of a two-level scene: nodes of the top-level tree).
inline PersistentParams persistent_params(const ExtendConfig &cfg, const DeviceScene &s) {
    PersistentParams pp{};
    pp.long_list = (s.tlas || !(cfg.quad && s.qnodes)) ? nullptr : cfg.long_list;   // four-child kernels only
    pp.long_cap = kLongListCap;
    pp.tail_budget = cfg.tail_budget;
    pp.refill = cfg.refill;
    pp.leaf_lanes = cfg.leaf_lanes;
    pp.spill = cfg.spill;
    pp.spill_entries = cfg.spill_entries;
    pp.stack_cap = cfg.stack_entries;
    const int top_nodes = (int)(cfg.quad ? s.n_q_top_cache : s.n_top_cache);
    pp.cache_nodes = cfg.cache_nodes < top_nodes ? cfg.cache_nodes : top_nodes;
    if (pp.cache_nodes < 0) pp.cache_nodes = 0;
    pp.xcd_slots = cfg.xcd_slots;
    return pp;
}

// Launch of a persistent kernel given as a function pointer (the template arguments are picked
// by the caller): grid sized to what can be resident, dynamic LDS as the loop parameters say.
template <class... KArgs, class... Args>
inline void launch_persistent(void (*kernel)(KArgs...), hipStream_t st, uint64_t n_blocks_wanted,
                       const ExtendConfig &cfg, const PersistentParams &pp, Args... args) {
    const unsigned int lds = persistent_lds_bytes(pp);
    const unsigned int grid = (unsigned int)std::min<uint64_t>(
        n_blocks_wanted, persistent_grid((const void *)kernel, PT_TRACE_BLOCK, cfg, lds));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(PT_TRACE_BLOCK), lds, st, args...);
}

// The second kernel of a persistent closest-hit launch that hands its long rays over
// (PersistentParams::long_list): one wave per listed ray.  Launch it on the same stream right
// after the traversal kernel, with the same rays and hits; it empties the list.  No-op when
// pp.long_list is null.  work may be null.
void launch_long_rays(hipStream_t st, const DeviceScene &s, const pt_ray *rays, pt_hit *hits, bool fixup,
                      const PersistentParams &pp, unsigned int *tie_count, DeviceTraceCounters *work);

// Occlusion query.  exact_order: walk the skip links as the reference does; otherwise the
// two-child nodes, nearer child first (the answer is the same).
void launch_trace_any(hipStream_t st, const DeviceScene &s, const pt_ray *rays, int32_t *occluded,
                      uint64_t n, bool exact_order, const ExtendConfig &cfg);

// The closest-hit query with every byte of device memory it needs beyond the caller's rays and
// hits taken from one file-scope device array (work counters and the stack-spill slots), so
// that the call allocates nothing.  static_scratch_init: once per device, with that device
// current, before the first launch there; it resolves the array's address on that device.
constexpr int kStaticSpillEntries = 16;          // global stack slots per lane in this form
constexpr unsigned int kStaticMaxGrid = 1280;    // persistent workgroups the slots are sized for
constexpr size_t kStaticCounterBytes = 256;
constexpr size_t kStaticScratchBytes =
    kStaticCounterBytes + (size_t)kStaticMaxGrid * PT_TRACE_BLOCK * kStaticSpillEntries * sizeof(int32_t);
int static_scratch_init(int device);
int launch_trace_static(hipStream_t st, int device, const DeviceScene &s, const pt_ray *rays,
                        pt_hit *hits, uint64_t n, uint32_t flags, ExtendConfig cfg);

void launch_raygen(hipStream_t st, const pt_camera &cam, uint32_t width, uint32_t height,
                   pt_ray *rays);

void launch_depth_shade(hipStream_t st, const pt_hit *hits, uint64_t n, float *rgba);

} // namespace mipt
