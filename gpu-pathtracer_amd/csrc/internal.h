// Library-internal definitions of the opaque handles.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <string>
#include <vector>

#include "../../include/mipt.h"
#include "kernels.h"

namespace mipt {

int fail(int code, const std::string &msg);
int hip_fail(hipError_t e, const char *what);

// Grow-only device allocation reused across calls.
struct DeviceBuffer {
    void *ptr = nullptr;
    size_t cap = 0;
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    ~DeviceBuffer() { release(); }
    int reserve(size_t bytes, bool exact = false);   // exact: no head-room for growth
    void release();
};

} // namespace mipt

struct pt_context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    bool external_stream = false;
    bool kernel_timing = false;
    int num_cus = 0;
    int extend_persistent = 1;      // option "extend.persistent"
    int extend_refill = 24;         // option "extend.refill"
    int extend_leaf_lanes = 16;     // option "extend.leaf_lanes"
    int extend_blocks_per_cu = -1;  // option "extend.blocks_per_cu" (-1: chosen per scene, 5 or 4)
    int extend_tail_budget = 64;    // option "extend.tail_budget": rounds a wave may spend on its last rays (0: all it takes)
    int extend_stack_entries = -1;  // option "extend.stack_entries" (-1: chosen per scene)
    int extend_cache_nodes = -1;    // option "extend.cache_nodes" (-1: chosen per scene)
    int any_exact_order = 0;        // option "any.exact_order"
    int extend_xcd_slots = 0;       // option "extend.xcd_slots"
    int extend_quad = -1;           // option "extend.quad": four-child nodes (-1: per scene)
    bool scene_quad = true;         // option "scene.quad": scenes get the four-child form built
    int sort_origin_bits = 10;      // option "trace.sort_origin_bits"
    int sort_dir_bits = 0;          // option "trace.sort_dir_bits"
    int trace_sort_rays = -1;       // option "trace.sort_rays": 0 off, 1 on, -1 by scene size
    int build_ploc_radius = 0;      // option "build.ploc_radius": 0 = radix tree, else clustering
    int build_optimize_passes = 6;  // option "build.optimize_passes": insertion-based optimisation rounds
    int build_treelet_every = 3;    // option "build.treelet_every": a treelet round after so many of them (0: none)
    int build_max_levels = 256;     // option "build.max_levels"
    // Triangle splitting in the GPU builder (bvh.sah.use_splits of the reference's builder,
    // src/bvh/bvh_builder.cpp:38-44): options "build.split_ratio" (a triangle whose box has this
    // many times the area its projections need is cut; 0 = off), "build.split_max" (leaves per
    // triangle at most), "build.split_budget" (extra leaves, per cent of the triangle count;
    // the reference's bvh.sah.extra_node_budget is 0.5, bvh_builder.cpp:32)
    int build_split_ratio = 4;
    int build_split_max = 64;
    int build_split_budget = 50;
    mipt::DeviceTraceCounters *counters = nullptr;
    mipt::DeviceBuffer io_rays;     // staging for the host-pointer entry points
    mipt::DeviceBuffer io_hits;
    mipt::DeviceBuffer render_ws;   // wavefront state of pt_render
    mipt::DeviceBuffer sort_scratch;    // keys and order of a sorted trace call
    mipt::DeviceBuffer render_counts;   // queue sizes per pass and bounce
    mipt::DeviceBuffer stack_spill;     // see ExtendConfig::spill and ensure_stack_spill
    uint32_t *long_list = nullptr;      // see PersistentParams::long_list (allocated with the context)
    std::vector<hipEvent_t> events;     // pool for per-kernel timing
    hipEvent_t trace_events[2] = {nullptr, nullptr};   // around the last trace call's kernel
    bool trace_timed = false;
    std::atomic<int> live_comms{0};     // pt_comm objects made on this context and not yet destroyed

    void release_all() {
        render_counts.release();
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        events.clear();
        for (hipEvent_t &e : trace_events) {
            if (e) (void)hipEventDestroy(e);
            e = nullptr;
        }
        io_rays.release();
        io_hits.release();
        render_ws.release();
        sort_scratch.release();
        stack_spill.release();
        if (counters) (void)hipFree(counters);
        counters = nullptr;
        if (long_list) (void)hipFree(long_list);
        long_list = nullptr;
        if (own_stream) (void)hipStreamDestroy(own_stream);
        own_stream = nullptr;
    }
};

namespace mipt {
// How the traversal kernels of a call on `scene` are launched: the context's options, and where
// they are left at "auto" the per-scene choice of how a persistent workgroup's LDS is split
// between the lanes' stacks and the cached top of the tree.  coherent: neighbouring rays of the
// call start next to each other (a trace call that takes its rays in origin-cell order, the
// queues of a render pass).
ExtendConfig make_extend_config(const pt_context *ctx, const pt_scene *scene, bool coherent);
}

struct pt_bvh {
    std::vector<pt_bbox> nodes;
    uint32_t link_encoding = PT_LINKS_FLOAT_VALUE;
};

namespace mipt {
void ensure_stack_spill(pt_context *ctx, uint32_t tree_depth, bool two_level);   // api.cpp
}

struct pt_film {
    pt_context *ctx = nullptr;
    uint32_t width = 0, height = 0;
    float4 *data = nullptr;
    bool owned = false;
};

struct pt_scene {
    pt_context *ctx = nullptr;
    mipt::DeviceScene dev{};
    std::vector<void *> allocs;
    void *face_ids = nullptr;            // device table of pt_scene_set_face_ids (also in allocs)
    // two-level scenes: where each mesh's skip-link nodes lie in dev.nodes
    std::vector<uint32_t> mesh_node_off, mesh_node_count;
    void release() {
        for (void *p : allocs) (void)hipFree(p);
        allocs.clear();
    }
    ~pt_scene() { release(); }
};
