// Internal interface of the GPU BVH builder (build.hip <-> api.cpp).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "internal.h"

namespace mipt {

struct DeviceBuildResult {
    hipError_t hip_error = hipSuccess;   // first failing HIP call, if any
    const char *what = "";
    bool no_memory = false;              // the scratch allocation failed
    bool bad_input = false;              // a triangle's box is not finite
    bool too_deep = false;               // the tree outgrew the level lists (handled inside)
    uint32_t max_depth = 0;              // levels of the tree, leaves included
};

// Decides which triangles go into several leaves (see "triangle splitting" in build.hip): a
// triangle whose box has more than `ratio` times the area its own projections need is cut into up
// to `max_pieces` slabs, within `budget` extra leaves per triangle of the mesh in all.  Fills
// `first` with n + 1 words, the first leaf of each triangle and the leaf count, and sets
// n_leaves; when no triangle is cut n_leaves == n and `first` is left alone.  ratio <= 0,
// max_pieces < 2 or budget <= 0: no splitting.  Blocks until done.
DeviceBuildResult plan_triangle_splits(hipStream_t st, const float *d_vtx, const int32_t *d_idx,
                                       uint32_t n, float ratio, int max_pieces, float budget,
                                       DeviceBuffer &first, uint32_t &n_leaves);

// Builds, from device-resident vertices and indices of n_tris triangles in n_leaves >= 2 leaves
// (d_first from plan_triangle_splits, or null with n_leaves == n_tris), the skip-link node
// array (2 n_leaves - 1 nodes, 2 x float4 each), the two-child nodes (n_leaves - 1, 4 x float4
// each) and the triangle records (n_leaves, 3 x float4 each) in leaf order, and the shading
// records (the same vertices with the material id, in the caller's triangle order; skipped if
// d_stris is null).
// ploc_radius 0: binary
// radix tree over the Morton codes; > 0: locally-ordered clustering with that search radius.
// optimize_passes > 0: that many rounds of parallel insertion-based optimisation of the tree;
// treelet_every > 0: a round of treelet restructuring after every so many of them.
// max_levels: an optimised tree deeper than this (at most 256) is given up for the radix tree,
// whose depth the key length bounds.
// Blocks until done.
DeviceBuildResult build_scene_on_device(hipStream_t st, const float *d_vtx, const int32_t *d_idx,
                                        uint32_t n_tris, uint32_t n_leaves, const uint32_t *d_first,
                                        uint32_t link_encoding, DeviceBuffer &scratch,
                                        const int32_t *d_mat_ids, float4 *d_nodes,
                                        float4 *d_wnodes, float4 *d_wtris, float4 *d_stris,
                                        int ploc_radius, int optimize_passes, int treelet_every, int max_levels = 256);

} // namespace mipt
