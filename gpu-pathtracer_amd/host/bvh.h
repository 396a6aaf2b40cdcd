// Host-side BVH: deterministic binned-SAH / spatial-median builder, the
// skip-link linearizer and the re-layout used by the ordered GPU traversal.
//
// Replaces, on the host side of the boundary:
//   bvh::build_bvh                     src/bvh/bvh_builder.h:37, bvh_builder.cpp:50-76
//   bvh::make_bvh option handling      src/bvh/bvh_builder.cpp:16-48
//   RadeonRays::Bvh::Build             external to the reference checkout
//   PlainBvhTranslator::Process        src/bvh/RadeonRays/plain_bvh_translator.cpp:39-96,223-246
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mipt.h"
#include "../../include/pt_config.h"

namespace mipt {

// Binary tree stored in pre-order: the left child of internal node i is i+1,
// right[i] is its right child; a leaf has right[i] < 0 and prim[i] >= 0.
// A subtree over n primitives occupies exactly 2n-1 consecutive slots, which
// is what lets disjoint subtrees be built by different threads with no
// shared counters and a result that does not depend on the thread count.
struct BvhTree {
    std::vector<pt_bbox> bounds;
    std::vector<int32_t> right;
    std::vector<int32_t> prim;
    uint32_t n_prims = 0;
    size_t size() const { return bounds.size(); }
};

// Build over (already inflated) leaf boxes.  Throws std::runtime_error.
void build_tree(const pt_bbox *leaf_bounds, uint32_t n, const pt_bvh_options &opt, BvhTree &out);

// src/bvh/bvh_builder.cpp:53-59
void inflate_leaf_bounds(pt_bbox *boxes, uint32_t n);

// src/main_granite.cpp:162-172
void mesh_leaf_bounds(const float *vtx, const int32_t *idx, uint32_t n_tri, pt_bbox *out);

// Pre-order tree -> 32-byte skip-link nodes (src/bvh/bvh_builder.h:26-34).
// link_encoding: PT_LINKS_FLOAT_VALUE (the reference's) or PT_LINKS_INT_BITS.
void linearize_skip_links(const BvhTree &t, uint32_t link_encoding, std::vector<pt_bbox> &out);

// Parse a skip-link node array back into a pre-order tree, validating every
// structural property the traversal kernels rely on.  Throws on violation.
void tree_from_skip_links(const pt_bbox *nodes, uint32_t n_nodes, uint32_t link_encoding,
                          uint32_t n_tri, BvhTree &out);

// 64-byte node of the ordered traversal: both children's boxes in one record.
//   lo.x/y/z/w = c0.min.x, c0.max.x, c0.min.y, c0.max.y   (16-byte vector 0)
//   hi...      = c1.min.x, c1.max.x, c1.min.y, c1.max.y   (16-byte vector 1)
//   zz         = c0.min.z, c0.max.z, c1.min.z, c1.max.z   (16-byte vector 2)
//   child[0..1]: >= 0 inner-node index, < 0 : ~(PT_TRI_RECORD_16B * triangle slot), i.e. the complemented
//   position of the 48-byte record in 16-byte units;  child[2..3] unused
struct WideNode {
    float c0xy[4];
    float c1xy[4];
    float cz[4];
    int32_t child[4];
};
static_assert(sizeof(WideNode) == 64, "WideNode must be 64 bytes");

// 48-byte triangle record: three vertices, the original triangle index and
// padding, stored in leaf order.
struct TriRecord {
    float v0[3]; float v1x;
    float v1yz[2]; float v2xy[2];
    float v2z; int32_t primid; int32_t pad[2];
};
static_assert(sizeof(TriRecord) == 48, "TriRecord must be 48 bytes");
// The record as the traversal arrays hold it: PT_TRI_RECORD_16B 16-byte units each.
#if PT_TRI_RECORD_16B == 3
using WideTriRecord = TriRecord;
#else
struct WideTriRecord : TriRecord {
    int32_t tail[4 * (PT_TRI_RECORD_16B - 3)];
};
#endif
static_assert(sizeof(WideTriRecord) == 16 * PT_TRI_RECORD_16B, "traversal triangle record size");

struct WideBvh {
    std::vector<WideNode> nodes;
    std::vector<WideTriRecord> tris;
    uint32_t max_depth = 0;
};

void build_wide_layout(const BvhTree &t, const float *vtx, uint32_t n_vtx, const int32_t *idx,
                       uint32_t n_tri, WideBvh &out);

// Emissive triangles with non-zero area, ascending, and the running binary32 sum of their
// areas (host/lights.cpp).
void build_light_list(const float *vtx, const int32_t *idx, uint32_t n_tri,
                      const pt_material *materials, const int32_t *material_ids,
                      std::vector<int32_t> &tri, std::vector<float> &cdf);

// Insertion-based optimisation of a built tree (bvh_optimize.cpp): `passes` rounds of taking every
// subtree out and putting it back where it adds the least surface area.  Leaf boxes unchanged.
void optimize_tree(BvhTree &t, int passes);
// Sum of the inner nodes' surface areas over the root's (the SAH's traversal term).
double tree_area_cost(const BvhTree &t);

void build_wide_top(const BvhTree &t, std::vector<WideNode> &out);

// Two-level scenes (instances.cpp).  m, inv: 3x4, three rows of (linear part | translation).
bool invert_affine(const float m[12], float inv[12]);           // false: singular or non-finite
void transform_point(const float m[12], const float p[3], float out[3]);
void instance_world_box(const float m[12], const pt_bbox &root, pt_bbox &out);

// String-keyed options as the reference sets them on RadeonRays::Options
// (src/bvh/bvh_builder.cpp:18-32, src/main_granite.cpp:175-176).
bool set_option(pt_bvh_options &opt, const std::string &key, const std::string &value,
                std::string &err);

} // namespace mipt
