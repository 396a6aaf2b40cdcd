// C ABI of libmipt.so (see include/mipt.h for what each entry point replaces).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mipt.h"
#include "../../include/pt_config.h"
#include "../host/bvh.h"
#include "build.h"
#include "internal.h"
#include "kernels.h"

using namespace mipt;

namespace mipt {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

int hip_fail(hipError_t e, const char *what) {
    return fail(PT_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

// Device memory for a scene array.  An array of a gigabyte or more is rounded up to whole
// gigabytes: the driver then maps it in larger pieces, and a traversal that scatters its fetches
// over several gigabytes pays less for address translation (28 M triangles, 2^24 rays: 5.13 ->
// 5.07 ms; powers of two the same, multiples of 256 MB nothing).  MIPT_ALLOC_POW2 (experiment
// knob): 0 exact sizes, 1 powers of two, k > 1 multiples of 2^k bytes for arrays of a megabyte up.
hipError_t scene_malloc(void **dst, size_t bytes) {
    static const int knob = [] {
        const char *e = getenv("MIPT_ALLOC_POW2");
        return e ? atoi(e) : -1;
    }();
    size_t want = bytes ? bytes : 16;
    if (knob < 0) {
        const size_t gig = (size_t)1 << 30;
        if (want >= gig) want = (want + gig - 1) / gig * gig;
    } else if (knob == 1 && want >= ((size_t)1 << 20)) {
        size_t p = (size_t)1 << 20;
        while (p < want) p <<= 1;
        want = p;
    } else if (knob > 1 && want >= ((size_t)1 << 20)) {
        const size_t unit = (size_t)1 << knob;
        want = (want + unit - 1) / unit * unit;
    }
    return hipMalloc(dst, want);
}

// Stack slots in global memory for the few rays that outgrow the LDS stack, for every lane of the
// largest persistent grid (100 MB on 256 compute units); without them such rays fall back to the
// fixed-order walk.  A ray's stack never holds more entries than the tree has levels, so a
// context gets the slots with the first scene whose tree is deeper than the smallest LDS stack a
// launch may be given (option "extend.stack_entries" >= 9), and keeps them.
void ensure_stack_spill(pt_context *ctx, uint32_t tree_depth, bool two_level) {
    if (ctx->stack_spill.ptr || (!two_level && tree_depth <= 8)) return;
    if (ctx->stack_spill.reserve((size_t)ctx->num_cus * kMaxLanesPerCu * kSpillEntries * sizeof(int32_t),
                                 /*exact=*/true) != PT_OK)
        (void)hipGetLastError();
}

int DeviceBuffer::reserve(size_t bytes, bool exact) {
    if (bytes <= cap) return PT_OK;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
    const size_t want = exact ? bytes : bytes + bytes / 4;
    hipError_t e = hipMalloc(&ptr, want);
    if (e != hipSuccess) {
        ptr = nullptr;
        return hip_fail(e, "hipMalloc");
    }
    cap = want;
    return PT_OK;
}

void DeviceBuffer::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
}

} // namespace mipt

namespace mipt {
ExtendConfig make_extend_config(const pt_context *ctx, const pt_scene *scene, bool coherent) {
    ExtendConfig c;
    c.device = ctx->device;
    c.persistent_blocks = ctx->extend_persistent ? 1 : 0;
    c.num_cus = ctx->num_cus;
    c.blocks_per_cu = ctx->extend_blocks_per_cu > 0 ? ctx->extend_blocks_per_cu : 5;
    c.refill = ctx->extend_refill;
    c.leaf_lanes = ctx->extend_leaf_lanes;
    c.sort_rays = 0;
    c.sort_origin_bits = ctx->sort_origin_bits;
    c.sort_dir_bits = ctx->sort_dir_bits;
    c.spill = (int32_t *)ctx->stack_spill.ptr;
    c.spill_entries = ctx->stack_spill.ptr ? kSpillEntries : 0;
    c.max_grid = 0;
    c.xcd_slots = ctx->extend_xcd_slots > 0 ? 1 : 0;
    c.quad = 0;
    c.long_list = ctx->long_list;
    c.tail_budget = ctx->extend_tail_budget;
    // The LDS of a persistent workgroup (measured on MI355X: 262 K-triangle scene in the caches;
    // 10 M- and 28 M-triangle scenes beyond them, host-built trees of 32-36 levels and GPU-built
    // ones of 50-70):
    //  - a tree of moderate depth whose records stay in the caches gets by with 16 stack
    //    entries per lane, and 224 cached nodes (the top 7-8 levels) take a tenth off the launch;
    //  - a deep tree needs its stack, or its rays live in the global spill slots: 24 entries +
    //    96 nodes.  So does any tree beyond the caches, where a spilled entry costs a memory
    //    round trip (36-level tree over 28 M triangles, rays in origin order: 13.1 ms with
    //    16 + 224, 12.4 ms with 24 + 96);
    //  - rays in no particular order (`coherent` false: a trace call whose rays were not sorted)
    //    on a scene beyond the Infinity Cache run best with three workgroups per compute unit
    //    instead of five -- fewer rays in flight thrash the caches less: 15.0 -> 13.2 ms on the
    //    10 M scene, 15.0 -> 14.1 ms on the 28 M scene -- and the LDS that frees goes to the full
    //    stack and 320 nodes.  The rays of a render pass are not of that kind (a bounce queue is
    //    in pixel order, so its origins are neighbours): five workgroups, 17.9 against 23.6 ms.
    const DeviceScene &d = scene->dev;
    const bool two_level = d.tlas != nullptr;
    const bool shallow = !two_level && d.wide_depth <= PT_CACHE_MAX_TREE_DEPTH;
    const uint64_t scene_bytes = 64ull * d.n_wnodes + 48ull * (two_level ? d.n_mesh_records : d.n_tri);
    const bool beyond_caches = scene_bytes > (256ull << 20);
    int stack = PT_LDS_STACK_ENTRIES, cache = 0;
    if (two_level) {
        // One stack serves both levels, and it keeps the whole LDS: the top of the top-level tree
        // can be cached like a single-level scene's (option extend.cache_nodes), but a ray makes
        // few of its steps there and the tree is small enough for the L2s -- measured on 4096
        // placements of a 65 K-triangle mesh, 2 M primary rays: 30 entries / no copy 2.83 Grays/s,
        // 20 / 160 nodes 2.77, 16 / 224 2.76, 12 / 288 2.67.
        stack = PT_LDS_STACK_ENTRIES; cache = 0;
    } else if (shallow && !beyond_caches) {
        stack = 16; cache = 224;
    } else if (!two_level) {
        if (beyond_caches && !coherent) {
            stack = PT_LDS_STACK_ENTRIES; cache = 320;
            if (ctx->extend_blocks_per_cu <= 0) c.blocks_per_cu = 3;
        } else {
            stack = 24; cache = 96;
        }
    }
    // Four-child nodes halve the node records a ray fetches and cost more arithmetic per record:
    // for scenes whose records come from memory rather than from the caches (2^24 secondary rays,
    // origin / queue order: 10 M triangles 9.1 / 13.1 -> 8.5 / 10.0 ms, 28 M triangles
    // 12.3 / 13.8 -> 11.1 / 11.9 ms; the bench scene, in the caches, loses 7 %).  A step pushes up to
    // three children, so the LDS goes to the stacks rather than to a copy of the top, and the
    // rays in flight no longer need thinning out: five workgroups per compute unit again.
    if (d.qnodes)
        c.quad = ctx->extend_quad >= 0 ? (ctx->extend_quad ? 1 : 0) : (beyond_caches ? 1 : 0);
    if (c.quad && beyond_caches && !two_level) {
        stack = coherent ? PT_LDS_STACK_ENTRIES : 24;
        cache = coherent ? 0 : 96;
        if (ctx->extend_blocks_per_cu <= 0) c.blocks_per_cu = 5;
    }
    c.stack_entries = ctx->extend_stack_entries > 0 ? ctx->extend_stack_entries : stack;
    // a four-child step pushes up to three entries and stops when they do not all fit: with fewer
    // than 12 entries it would stop before a block of 8 could be moved to global memory
    if (c.quad && c.stack_entries < 12) c.stack_entries = 12;
    c.cache_nodes = ctx->extend_cache_nodes >= 0 ? ctx->extend_cache_nodes : cache;
    return c;
}
} // namespace mipt

#define PT_TRY try {
#define PT_CATCH                                                                   \
    }                                                                              \
    catch (const std::bad_alloc &) { return fail(PT_ERR_NOMEM, "out of host memory"); } \
    catch (const std::exception &e) { return fail(PT_ERR_INVALID, e.what()); }     \
    catch (...) { return fail(PT_ERR_INVALID, "unknown error"); }

#define HIP_CHECK(expr)                                                  \
    do {                                                                 \
        hipError_t _e = (expr);                                          \
        if (_e != hipSuccess) return hip_fail(_e, #expr);                \
    } while (0)

extern "C" {

const char *pt_last_error(void) { return g_last_error.c_str(); }

const char *pt_version(void) { return "mipt 0.1 (gfx950)"; }

int pt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

/* ---------------------------------------------------------------------- */

void pt_bvh_options_default(pt_bvh_options *opt) {
    if (!opt) return;
    opt->use_sah = 0;
    opt->use_splits = 0;
    opt->max_split_depth = 10;      // src/bvh/bvh_builder.cpp:28
    opt->num_bins = 64;             // :29
    opt->min_overlap = 0.05f;       // :30
    opt->traversal_cost = 10.f;     // :31
    opt->extra_node_budget = 0.5f;  // :32
    opt->link_encoding = PT_LINKS_FLOAT_VALUE;
    opt->num_threads = 0;
    opt->optimize_passes = 0;
}

int pt_bvh_options_set(pt_bvh_options *opt, const char *key, const char *value) {
    if (!opt || !key || !value) return fail(PT_ERR_INVALID, "pt_bvh_options_set: null argument");
    PT_TRY
    std::string err;
    if (!set_option(*opt, key, value, err)) return fail(PT_ERR_INVALID, err);
    return PT_OK;
    PT_CATCH
}

int pt_mesh_leaf_bounds(const float *vtx, uint32_t n_vtx, const int32_t *idx, uint32_t n_tri,
                        pt_bbox *out) {
    if (!vtx || !idx || !out) return fail(PT_ERR_INVALID, "pt_mesh_leaf_bounds: null argument");
    for (uint64_t i = 0; i < 3ull * n_tri; ++i)
        if (idx[i] < 0 || (uint32_t)idx[i] >= n_vtx)
            return fail(PT_ERR_INVALID, "pt_mesh_leaf_bounds: index " + std::to_string(idx[i]) +
                                            " outside the vertex buffer");
    PT_TRY
    mesh_leaf_bounds(vtx, idx, n_tri, out);
    return PT_OK;
    PT_CATCH
}

int pt_bvh_build(const pt_bbox *leaf_bounds, uint32_t n_leaves, const pt_bvh_options *opt,
                 pt_bvh **out) {
    if (!leaf_bounds || !out) return fail(PT_ERR_INVALID, "pt_bvh_build: null argument");
    *out = nullptr;
    PT_TRY
    pt_bvh_options o;
    if (opt) o = *opt; else pt_bvh_options_default(&o);
    std::vector<pt_bbox> scaled(leaf_bounds, leaf_bounds + n_leaves);
    inflate_leaf_bounds(scaled.data(), n_leaves);
    BvhTree tree;
    build_tree(scaled.data(), n_leaves, o, tree);
    optimize_tree(tree, o.optimize_passes);
    auto bvh = std::make_unique<pt_bvh>();
    bvh->link_encoding = o.link_encoding;
    linearize_skip_links(tree, o.link_encoding, bvh->nodes);
    *out = bvh.release();
    return PT_OK;
    PT_CATCH
}

int pt_bvh_optimize(const pt_bbox *nodes, uint32_t n_nodes, uint32_t link_encoding, int32_t passes,
                    pt_bvh **out) {
    if (!nodes || !out) return fail(PT_ERR_INVALID, "pt_bvh_optimize: null argument");
    *out = nullptr;
    if (passes < 0 || passes > 64) return fail(PT_ERR_INVALID, "pt_bvh_optimize: passes must be in [0, 64]");
    PT_TRY
    BvhTree tree;
    tree_from_skip_links(nodes, n_nodes, link_encoding, 0, tree);
    optimize_tree(tree, passes);
    auto bvh = std::make_unique<pt_bvh>();
    bvh->link_encoding = link_encoding;
    linearize_skip_links(tree, link_encoding, bvh->nodes);
    *out = bvh.release();
    return PT_OK;
    PT_CATCH
}

uint32_t pt_bvh_node_count(const pt_bvh *bvh) { return bvh ? (uint32_t)bvh->nodes.size() : 0; }
const pt_bbox *pt_bvh_nodes(const pt_bvh *bvh) { return bvh ? bvh->nodes.data() : nullptr; }
uint32_t pt_bvh_link_encoding(const pt_bvh *bvh) { return bvh ? bvh->link_encoding : 0; }
int pt_bvh_destroy(pt_bvh *bvh) {
    delete bvh;
    return PT_OK;
}

int pt_bvh_linearize(const pt_bbox *bounds, const int32_t *left, const int32_t *right,
                     const int32_t *prim, uint32_t n_nodes, uint32_t link_encoding,
                     pt_bbox *out_nodes) {
    if (!bounds || !left || !right || !prim || !out_nodes)
        return fail(PT_ERR_INVALID, "pt_bvh_linearize: null argument");
    if (n_nodes == 0) return fail(PT_ERR_INVALID, "pt_bvh_linearize: empty tree");
    PT_TRY
    // Number the nodes in pre-order (ProcessNode, plain_bvh_translator.cpp:223-246).
    BvhTree t;
    t.bounds.reserve(n_nodes);
    t.right.reserve(n_nodes);
    t.prim.reserve(n_nodes);
    std::vector<uint8_t> seen(n_nodes, 0);
    struct Frame { uint32_t node; int32_t parent_slot; };
    std::vector<Frame> stack;
    stack.push_back({0, -1});
    uint32_t leaves = 0;
    while (!stack.empty()) {
        const Frame f = stack.back();
        stack.pop_back();
        if (f.node >= n_nodes) return fail(PT_ERR_INVALID, "pt_bvh_linearize: child out of range");
        if (seen[f.node]) return fail(PT_ERR_INVALID, "pt_bvh_linearize: node reached twice");
        seen[f.node] = 1;
        const int32_t slot = (int32_t)t.bounds.size();
        if (f.parent_slot >= 0) t.right[(size_t)f.parent_slot] = slot;
        pt_bbox b = bounds[f.node];
        b.pmin[3] = b.pmax[3] = 0.f;
        t.bounds.push_back(b);
        if (left[f.node] < 0) {
            t.right.push_back(-1);
            t.prim.push_back(prim[f.node]);
            if (prim[f.node] < 0)
                return fail(PT_ERR_INVALID, "pt_bvh_linearize: leaf without a primitive");
            ++leaves;
        } else {
            if (right[f.node] < 0)
                return fail(PT_ERR_INVALID, "pt_bvh_linearize: internal node with one child");
            t.right.push_back(-1);  // patched when the right child is numbered
            t.prim.push_back(-1);
            stack.push_back({(uint32_t)right[f.node], slot});   // visited after the left subtree
            stack.push_back({(uint32_t)left[f.node], -1});
        }
    }
    if (t.bounds.size() != n_nodes)
        return fail(PT_ERR_INVALID, "pt_bvh_linearize: nodes not reachable from the root");
    t.n_prims = leaves;
    std::vector<pt_bbox> out;
    linearize_skip_links(t, link_encoding, out);
    std::memcpy(out_nodes, out.data(), sizeof(pt_bbox) * out.size());
    return PT_OK;
    PT_CATCH
}

int pt_bvh_validate(const pt_bbox *nodes, uint32_t n_nodes, uint32_t link_encoding,
                    uint32_t n_tri) {
    if (!nodes) return fail(PT_ERR_INVALID, "pt_bvh_validate: null argument");
    PT_TRY
    BvhTree t;
    tree_from_skip_links(nodes, n_nodes, link_encoding, n_tri, t);
    if (n_tri) {
        // A tree built with spatial splits holds a triangle in several leaves; none may be
        // missing.
        if (t.n_prims < n_tri)
            return fail(PT_ERR_INVALID, "skip-link array: " + std::to_string(t.n_prims) +
                                            " leaves for " + std::to_string(n_tri) + " triangles");
        std::vector<uint8_t> seen(n_tri, 0);
        for (size_t i = 0; i < t.size(); ++i)
            if (t.prim[i] >= 0) seen[(size_t)t.prim[i]] = 1;
        for (uint32_t k = 0; k < n_tri; ++k)
            if (!seen[k])
                return fail(PT_ERR_INVALID,
                            "skip-link array: triangle " + std::to_string(k) + " is in no leaf");
    }
    return PT_OK;
    PT_CATCH
}

/* ---------------------------------------------------------------------- */

int pt_context_create(int device, pt_context **out) {
    if (!out) return fail(PT_ERR_INVALID, "pt_context_create: null argument");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        (void)hipGetLastError();
        return fail(PT_ERR_DEVICE, "pt_context_create: no HIP device available (this library has "
                                   "no CPU path)");
    }
    if (device < 0 || device >= n)
        return fail(PT_ERR_INVALID, "pt_context_create: device " + std::to_string(device) +
                                        " out of range (" + std::to_string(n) + " visible)");
    PT_TRY
    HIP_CHECK(hipSetDevice(device));
    auto ctx = std::make_unique<pt_context>();
    ctx->device = device;
    HIP_CHECK(hipDeviceGetAttribute(&ctx->num_cus, hipDeviceAttributeMultiprocessorCount, device));
    HIP_CHECK(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    HIP_CHECK(hipMalloc((void **)&ctx->counters, sizeof(DeviceTraceCounters)));
    HIP_CHECK(hipMemsetAsync(ctx->counters, 0, sizeof(DeviceTraceCounters), ctx->stream));
    {   // the list of rays a closest-hit launch hands to its wave-per-ray kernel (256 KB)
        const size_t bytes = sizeof(uint32_t) * (kLongListHeader + (size_t)kLongListCap);
        HIP_CHECK(hipMalloc((void **)&ctx->long_list, bytes));
        HIP_CHECK(hipMemsetAsync(ctx->long_list, 0, bytes, ctx->stream));
    }
    // (the global stack slots of rays that outgrow the LDS stack come with the first scene whose
    // tree is deep enough to need them: ensure_stack_spill)
    // this device's address of the static scratch array of pt_trace_closest_device_static
    const int src = static_scratch_init(device);
    if (src != PT_OK) return src;
    *out = ctx.release();
    return PT_OK;
    PT_CATCH
}

int pt_context_set_stream(pt_context *ctx, void *hip_stream) {
    if (!ctx) return fail(PT_ERR_INVALID, "pt_context_set_stream: null context");
    HIP_CHECK(hipSetDevice(ctx->device));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    ctx->external_stream = hip_stream != nullptr;
    return PT_OK;
}

void *pt_context_stream(pt_context *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int pt_context_synchronize(pt_context *ctx) {
    if (!ctx) return fail(PT_ERR_INVALID, "pt_context_synchronize: null context");
    HIP_CHECK(hipSetDevice(ctx->device));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_context_set_kernel_timing(pt_context *ctx, int on) {
    if (!ctx) return fail(PT_ERR_INVALID, "pt_context_set_kernel_timing: null context");
    ctx->kernel_timing = on != 0;
    return PT_OK;
}

int pt_context_last_trace_ms(pt_context *ctx, double *ms) {
    if (!ctx || !ms) return fail(PT_ERR_INVALID, "pt_context_last_trace_ms: null argument");
    if (!ctx->trace_timed)
        return fail(PT_ERR_INVALID, "pt_context_last_trace_ms: no pt_trace_closest* call was made "
                                    "with kernel timing on");
    HIP_CHECK(hipSetDevice(ctx->device));
    HIP_CHECK(hipEventSynchronize(ctx->trace_events[1]));
    float t = 0.f;
    HIP_CHECK(hipEventElapsedTime(&t, ctx->trace_events[0], ctx->trace_events[1]));
    *ms = t;
    return PT_OK;
}

int pt_context_set_option(pt_context *ctx, const char *key, const char *value) {
    if (!ctx || !key || !value) return fail(PT_ERR_INVALID, "pt_context_set_option: null argument");
    char *end = nullptr;
    const long v = std::strtol(value, &end, 10);
    if (end == value) return fail(PT_ERR_INVALID, std::string("option '") + key + "': not an integer");
    const std::string k(key);
    if (k == "extend.persistent") {
        ctx->extend_persistent = v != 0;
    } else if (k == "extend.leaf_lanes") {
        if (v < 1 || v > 64) return fail(PT_ERR_INVALID, "extend.leaf_lanes must be in [1,64]");
        ctx->extend_leaf_lanes = (int)v;
    } else if (k == "extend.tail_budget") {
        if (v < 0 || v > 1000000) return fail(PT_ERR_INVALID, "extend.tail_budget must be in [0,1000000]");
        ctx->extend_tail_budget = (int)v;
    } else if (k == "build.ploc_radius") {
        if (v < 0 || v > 64) return fail(PT_ERR_INVALID, "build.ploc_radius must be in [0,64]");
        ctx->build_ploc_radius = (int)v;
    } else if (k == "build.optimize_passes") {
        if (v < 0 || v > 32) return fail(PT_ERR_INVALID, "build.optimize_passes must be in [0,32]");
        ctx->build_optimize_passes = (int)v;
    } else if (k == "build.treelet_every") {
        if (v < 0 || v > 32) return fail(PT_ERR_INVALID, "build.treelet_every must be in [0,32]");
        ctx->build_treelet_every = (int)v;
    } else if (k == "build.max_levels") {
        if (v < 8 || v > 256) return fail(PT_ERR_INVALID, "build.max_levels must be in [8,256]");
        ctx->build_max_levels = (int)v;
    } else if (k == "build.split_ratio") {
        if (v < 0 || v > 1000000) return fail(PT_ERR_INVALID, "build.split_ratio must be in [0,1000000]");
        ctx->build_split_ratio = (int)v;
    } else if (k == "build.split_max") {
        if (v < 1 || v > 256) return fail(PT_ERR_INVALID, "build.split_max must be in [1,256]");
        ctx->build_split_max = (int)v;
    } else if (k == "build.split_budget") {
        if (v < 0 || v > 1000) return fail(PT_ERR_INVALID, "build.split_budget must be in [0,1000] (per cent)");
        ctx->build_split_budget = (int)v;
    } else if (k == "trace.sort_rays") {
        if (v < -1 || v > 1) return fail(PT_ERR_INVALID, "trace.sort_rays must be -1, 0 or 1");
        ctx->trace_sort_rays = (int)v;
    } else if (k == "extend.xcd_slots") {
        ctx->extend_xcd_slots = (int)v;
    } else if (k == "extend.quad") {
        if (v < -1 || v > 1) return fail(PT_ERR_INVALID, "extend.quad must be -1, 0 or 1");
        ctx->extend_quad = (int)v;
    } else if (k == "scene.quad") {
        ctx->scene_quad = v != 0;
    } else if (k == "trace.sort_origin_bits") {
        if (v < 1 || v > 10) return fail(PT_ERR_INVALID, "trace.sort_origin_bits must be in [1,10]");
        ctx->sort_origin_bits = (int)v;
    } else if (k == "trace.sort_dir_bits") {
        if (v < 0 || v > 8) return fail(PT_ERR_INVALID, "trace.sort_dir_bits must be in [0,8]");
        ctx->sort_dir_bits = (int)v;
    } else if (k == "any.exact_order") {
        ctx->any_exact_order = v != 0;
    } else if (k == "extend.refill") {
        if (v < 1 || v > 64) return fail(PT_ERR_INVALID, "extend.refill must be in [1,64]");
        ctx->extend_refill = (int)v;
    } else if (k == "extend.stack_entries") {
        if (v != -1 && (v < 9 || v > PT_LDS_STACK_ENTRIES))
            return fail(PT_ERR_INVALID, "extend.stack_entries must be -1 or in [9," +
                                            std::to_string(PT_LDS_STACK_ENTRIES) + "]");
        ctx->extend_stack_entries = (int)v;
    } else if (k == "extend.cache_nodes") {
        if (v < -1 || v > PT_LDS_NODE_CACHE_MAX)
            return fail(PT_ERR_INVALID, "extend.cache_nodes must be in [-1," +
                                            std::to_string(PT_LDS_NODE_CACHE_MAX) + "]");
        ctx->extend_cache_nodes = (int)v;
    } else if (k == "extend.blocks_per_cu") {
        if (v != -1 && (v < 1 || v > 8))
            return fail(PT_ERR_INVALID, "extend.blocks_per_cu must be -1 or in [1,8]");
        ctx->extend_blocks_per_cu = (int)v;
    } else {
        return fail(PT_ERR_INVALID, "unknown option '" + k + "'");
    }
    return PT_OK;
}

int pt_context_destroy(pt_context *ctx) {
    if (!ctx) return PT_OK;
    // a communicator uses its context's device and stream until it is destroyed
    if (ctx->live_comms.load() > 0)
        return fail(PT_ERR_INVALID, "pt_context_destroy: " + std::to_string(ctx->live_comms.load()) +
                                        " communicator(s) of this context are still alive: "
                                        "pt_comm_destroy them first");
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->release_all();
    delete ctx;
    return PT_OK;
}

/* ---------------------------------------------------------------------- */


// The copy of the top of the tree that the persistent kernels keep in LDS (single-level scenes).
static int attach_top_cache(pt_context *ctx, pt_scene *sc) {
    if (sc->dev.n_wnodes < 2) return PT_OK;
    void *buf = nullptr;
    // nodes, then each node's index in wnodes, then the count the builder reports
    const size_t node_bytes = 64 * (size_t)kNodeCacheMax, index_bytes = 4 * (size_t)kNodeCacheMax;
    hipError_t e = hipMalloc(&buf, node_bytes + index_bytes + 16);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(top cache)");
    sc->allocs.push_back(buf);
    int32_t *index = reinterpret_cast<int32_t *>((char *)buf + node_bytes);
    uint32_t *n_dev = reinterpret_cast<uint32_t *>((char *)buf + node_bytes + index_bytes);
    launch_build_top_cache(ctx->stream, sc->dev.wnodes, sc->dev.n_wnodes, (float4 *)buf, index, n_dev);
    uint32_t n = 0;
    e = hipMemcpyAsync(&n, n_dev, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return hip_fail(e, "top cache build");
    sc->dev.top_cache = (const float4 *)buf;
    sc->dev.top_global = index;
    sc->dev.n_top_cache = n;
    return PT_OK;
}

// The four-child form of the tree (DeviceScene::qnodes / qtris) and its LDS top, for scenes
// whose records fit 32-bit byte offsets.  Costs 64 bytes per node and per leaf; a scene it cannot
// be made for (a bound that is not finite, no memory) simply keeps the two-child form only.
static int attach_quad(pt_context *ctx, pt_scene *sc) {
    DeviceScene &d = sc->dev;
    if (!ctx->scene_quad || d.n_wnodes < 1 || d.n_tri < 2 || d.tlas) return PT_OK;
    const size_t n_leaves = (size_t)d.n_wnodes + 1;
    // a leaf is named by its record's position in 16-byte units (four per record) in 31 bits;
    // records beyond 4 GiB take the kernels with 64-bit addresses, like the two-child form
    if (4 * n_leaves >= (1ull << 31)) return PT_OK;
    if (!d.big && needs_big_addresses(d.n_wnodes, n_leaves * 4 / PT_TRI_RECORD_16B + 1)) d.big = 1;
    const size_t node_bytes = 64 * (size_t)d.n_wnodes, tri_bytes = 64 * n_leaves;
    const size_t cache_bytes = 64 * (size_t)kNodeCacheMax, index_bytes = 4 * (size_t)kNodeCacheMax;
    void *qn = nullptr, *qt = nullptr, *top = nullptr;
    auto give_up = [&]() {
        (void)hipGetLastError();
        if (qn) (void)hipFree(qn);
        if (qt) (void)hipFree(qt);
        if (top) (void)hipFree(top);
        return PT_OK;
    };
    if (scene_malloc(&qn, node_bytes) != hipSuccess) return give_up();
    if (scene_malloc(&qt, tri_bytes) != hipSuccess) return give_up();
    if (hipMalloc(&top, cache_bytes + index_bytes + 32) != hipSuccess) return give_up();
    int32_t *index = reinterpret_cast<int32_t *>((char *)top + cache_bytes);
    uint32_t *words = reinterpret_cast<uint32_t *>((char *)top + cache_bytes + index_bytes);   // n, bad
    uint32_t host_words[2] = {0, 1};
    hipError_t e = hipMemsetAsync(words, 0, 32, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(qt, 0, tri_bytes, ctx->stream);
    if (e == hipSuccess) {
        launch_build_quad(ctx->stream, d.wnodes, d.n_wnodes, d.wtris, d.n_tri, (float4 *)qn, (float4 *)qt,
                          words + 1);
        launch_build_quad_top_cache(ctx->stream, (const float4 *)qn, d.n_wnodes, (float4 *)top, index, words);
        e = hipMemcpyAsync(host_words, words, 8, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess || host_words[1] != 0) return give_up();
    sc->allocs.push_back(qn);
    sc->allocs.push_back(qt);
    sc->allocs.push_back(top);
    d.qnodes = (const float4 *)qn;
    d.qtris = (const float4 *)qt;
    d.q_top_cache = (const float4 *)top;
    d.q_top_global = index;
    d.n_q_top_cache = host_words[0];
    d.quad_depth = (d.wide_depth + 1) / 2;
    return PT_OK;
}

// Each triangle's leaf in the skip-link array, for deciding two-candidate ties at their leaves
// (replay_candidates): one word per triangle, or, when a triangle has several leaves (a tree
// with spatial splits), a list per triangle.
static int attach_tri_leaf(pt_context *ctx, pt_scene *sc) {
    const DeviceScene &d = sc->dev;
    if (d.n_tri < 2) return PT_OK;
    void *buf = nullptr;
    const size_t bytes = 4 * (size_t)d.n_tri;
    hipError_t e = hipMalloc(&buf, bytes + 16);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(triangle leaves)");
    uint32_t *multi_dev = reinterpret_cast<uint32_t *>((char *)buf + bytes);
    uint32_t multi = 1;
    e = hipMemsetAsync(buf, 0xFF, bytes, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(multi_dev, 0, 16, ctx->stream);
    if (e == hipSuccess) {
        launch_build_tri_leaf(ctx->stream, d.nodes, d.n_nodes, d.link_enc, d.n_tri, (int32_t *)buf, multi_dev);
        e = hipMemcpyAsync(&multi, multi_dev, 4, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(buf);
        return hip_fail(e, "triangle leaf table");
    }
    if (!multi) {
        sc->allocs.push_back(buf);
        sc->dev.tri_leaf = (const int32_t *)buf;
        return PT_OK;
    }
    (void)hipFree(buf);
    // Several leaves per triangle.  Without the lists such a scene still answers every query:
    // its tied rays are then walked over the whole tree.
    void *first = nullptr, *list = nullptr;
    if (hipMalloc(&first, 4 * ((size_t)d.n_tri + 1)) != hipSuccess ||
        hipMalloc(&list, 4 * (size_t)d.n_nodes) != hipSuccess) {
        (void)hipGetLastError();
        if (first) (void)hipFree(first);
        return PT_OK;
    }
    DeviceBuffer scratch;
    e = build_tri_leaf_lists(ctx->stream, d.nodes, d.n_nodes, d.link_enc, d.n_tri, (uint32_t *)first,
                             (int32_t *)list, scratch);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(first);
        (void)hipFree(list);
        return PT_OK;
    }
    sc->allocs.push_back(first);
    sc->allocs.push_back(list);
    sc->dev.tri_leaf_first = (const uint32_t *)first;
    sc->dev.tri_leaf_list = (const int32_t *)list;
    return PT_OK;
}

int pt_scene_create(pt_context *ctx, const pt_bbox *nodes, uint32_t n_nodes,
                    uint32_t link_encoding, const float *vtx, uint32_t n_vtx, const int32_t *idx,
                    uint32_t n_tri, const pt_material *materials, uint32_t n_materials,
                    const int32_t *material_ids, pt_scene **out) {
    if (!ctx || !nodes || !vtx || !idx || !out)
        return fail(PT_ERR_INVALID, "pt_scene_create: null argument");
    *out = nullptr;
    if (n_tri == 0 || n_vtx == 0) return fail(PT_ERR_INVALID, "pt_scene_create: empty mesh");
    if ((materials == nullptr) != (material_ids == nullptr) || (materials && n_materials == 0))
        return fail(PT_ERR_INVALID,
                    "pt_scene_create: materials and material_ids must be given together");
    PT_TRY
    HIP_CHECK(hipSetDevice(ctx->device));
    BvhTree tree;
    tree_from_skip_links(nodes, n_nodes, link_encoding, n_tri, tree);
    if (tree.n_prims < n_tri)
        return fail(PT_ERR_INVALID, "pt_scene_create: " + std::to_string(tree.n_prims) +
                                        " leaves for " + std::to_string(n_tri) + " triangles");
    {
        // Several leaves may hold one triangle (spatial splits); every triangle must be held.
        std::vector<uint8_t> seen(n_tri, 0);
        for (size_t i = 0; i < tree.size(); ++i)
            if (tree.prim[i] >= 0) seen[(size_t)tree.prim[i]] = 1;
        for (uint32_t k = 0; k < n_tri; ++k)
            if (!seen[k])
                return fail(PT_ERR_INVALID,
                            "pt_scene_create: triangle " + std::to_string(k) + " is in no leaf");
    }
    for (uint64_t i = 0; i < 3ull * n_tri; ++i)
        if (idx[i] < 0 || (uint32_t)idx[i] >= n_vtx)
            return fail(PT_ERR_INVALID, "pt_scene_create: index " + std::to_string(idx[i]) +
                                            " outside the vertex buffer");
    std::vector<pt_material> mats;
    std::vector<int32_t> mids;
    if (materials) {
        mats.assign(materials, materials + n_materials);
        mids.assign(material_ids, material_ids + n_tri);
        for (uint32_t t = 0; t < n_tri; ++t)
            if (mids[t] < 0 || (uint32_t)mids[t] >= n_materials)
                return fail(PT_ERR_INVALID, "pt_scene_create: material id " +
                                                std::to_string(mids[t]) + " out of range");
    } else {
        pt_material m{};
        m.kd[0] = m.kd[1] = m.kd[2] = 0.5f;
        m.roughness = 1.f;
        mats.push_back(m);
        mids.assign(n_tri, 0);
    }
    WideBvh wide;
    build_wide_layout(tree, vtx, n_vtx, idx, n_tri, wide);
    std::vector<int32_t> light_tri;
    std::vector<float> light_cdf;
    build_light_list(vtx, idx, n_tri, mats.data(), mids.data(), light_tri, light_cdf);
    std::vector<TriRecord> stris(n_tri);
    for (uint32_t t = 0; t < n_tri; ++t) {
        const float *a = vtx + 3 * (size_t)idx[3 * (size_t)t], *b = vtx + 3 * (size_t)idx[3 * (size_t)t + 1],
                    *c = vtx + 3 * (size_t)idx[3 * (size_t)t + 2];
        TriRecord &r = stris[t];
        r.v0[0] = a[0]; r.v0[1] = a[1]; r.v0[2] = a[2];
        r.v1x = b[0]; r.v1yz[0] = b[1]; r.v1yz[1] = b[2];
        r.v2xy[0] = c[0]; r.v2xy[1] = c[1]; r.v2z = c[2];
        r.primid = mids[t];                     // this copy carries the material id
        r.pad[0] = r.pad[1] = 0;
    }

    auto sc = std::make_unique<pt_scene>();
    sc->ctx = ctx;
    auto upload = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = scene_malloc(dst, bytes);
        if (e != hipSuccess) return e;
        sc->allocs.push_back(*dst);
        return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    void *d_nodes, *d_vtx, *d_idx, *d_wn, *d_wt, *d_st, *d_mid, *d_mat, *d_lt = nullptr, *d_lc = nullptr;
    if (wide.nodes.size() > kMaxWideNodes || wide.tris.size() > kMaxWideTris)
        return fail(PT_ERR_UNSUPPORTED, "pt_scene_create: more nodes or triangle records than a child index can name");
    hipError_t e;
#define UP(dst, src, bytes)                                                     \
    if ((e = upload(&dst, src, bytes)) != hipSuccess) {                         \
        sc->release();                                                          \
        return hip_fail(e, "pt_scene_create upload");                           \
    }
    UP(d_nodes, nodes, sizeof(pt_bbox) * (size_t)n_nodes);
    UP(d_vtx, vtx, sizeof(float) * 3 * (size_t)n_vtx);
    UP(d_idx, idx, sizeof(int32_t) * 3 * (size_t)n_tri);
    UP(d_wn, wide.nodes.data(), sizeof(WideNode) * wide.nodes.size());
    UP(d_wt, wide.tris.data(), sizeof(WideTriRecord) * wide.tris.size());
    UP(d_st, stris.data(), sizeof(TriRecord) * stris.size());
    UP(d_mid, mids.data(), sizeof(int32_t) * mids.size());
    UP(d_mat, mats.data(), sizeof(pt_material) * mats.size());
    if (!light_tri.empty()) {
        UP(d_lt, light_tri.data(), sizeof(int32_t) * light_tri.size());
        UP(d_lc, light_cdf.data(), sizeof(float) * light_cdf.size());
    }
#undef UP
    DeviceScene &d = sc->dev;
    d.nodes = (const float4 *)d_nodes;
    d.vtx = (const float *)d_vtx;
    d.idx = (const int32_t *)d_idx;
    d.wnodes = (const float4 *)d_wn;
    d.wtris = (const float4 *)d_wt;
    d.stris = (const float4 *)d_st;
    d.mat_ids = (const int32_t *)d_mid;
    d.materials = (const pt_material *)d_mat;
    d.light_tri = (const int32_t *)d_lt;
    d.light_cdf = (const float *)d_lc;
    d.n_lights = (uint32_t)light_tri.size();
    d.light_area = light_cdf.empty() ? 0.f : light_cdf.back();
    d.n_nodes = n_nodes;
    d.n_tri = n_tri;
    d.n_wnodes = (uint32_t)wide.nodes.size();
    d.n_materials = (uint32_t)mats.size();
    d.link_enc = link_encoding;
    d.wide_depth = wide.max_depth;
    d.big = needs_big_addresses(wide.nodes.size(), wide.tris.size()) ? 1u : 0u;
    if (n_tri >= 2) {
        int rc = attach_top_cache(ctx, sc.get());
        if (rc == PT_OK) rc = attach_tri_leaf(ctx, sc.get());
        if (rc == PT_OK) rc = attach_quad(ctx, sc.get());
        if (rc != PT_OK) return rc;
    }
    ensure_stack_spill(ctx, d.wide_depth, false);
    *out = sc.release();
    return PT_OK;
    PT_CATCH
}

int pt_scene_build(pt_context *ctx, const float *vtx, uint32_t n_vtx, const int32_t *idx,
                   uint32_t n_tri, const pt_material *materials, uint32_t n_materials,
                   const int32_t *material_ids, uint32_t link_encoding, pt_scene **out) {
    if (!ctx || !vtx || !idx || !out) return fail(PT_ERR_INVALID, "pt_scene_build: null argument");
    *out = nullptr;
    if (n_tri == 0 || n_vtx == 0) return fail(PT_ERR_INVALID, "pt_scene_build: empty mesh");
    if (link_encoding != PT_LINKS_FLOAT_VALUE && link_encoding != PT_LINKS_INT_BITS)
        return fail(PT_ERR_INVALID, "pt_scene_build: unknown link encoding");
    if ((materials == nullptr) != (material_ids == nullptr) || (materials && n_materials == 0))
        return fail(PT_ERR_INVALID,
                    "pt_scene_build: materials and material_ids must be given together");
    if (link_encoding == PT_LINKS_FLOAT_VALUE && 2ull * n_tri - 1 > (1u << 24))
        return fail(PT_ERR_INVALID,
                    "pt_scene_build: more than 2^24 nodes need PT_LINKS_INT_BITS (indices above "
                    "2^24 are not exact as float values)");
    if (n_tri > kMaxWideTris)
        return fail(PT_ERR_UNSUPPORTED, "pt_scene_build: more triangles than a child index can name");
    PT_TRY
    for (uint64_t i = 0; i < 3ull * n_tri; ++i)
        if (idx[i] < 0 || (uint32_t)idx[i] >= n_vtx)
            return fail(PT_ERR_INVALID, "pt_scene_build: index " + std::to_string(idx[i]) +
                                            " outside the vertex buffer");
    if (n_tri == 1) {
        // A one-leaf tree has no inner node to build: take the host path.
        pt_bbox lb;
        mesh_leaf_bounds(vtx, idx, 1, &lb);
        pt_bvh_options opt;
        pt_bvh_options_default(&opt);
        opt.link_encoding = link_encoding;
        pt_bvh *b = nullptr;
        int rc = pt_bvh_build(&lb, 1, &opt, &b);
        if (rc != PT_OK) return rc;
        rc = pt_scene_create(ctx, b->nodes.data(), (uint32_t)b->nodes.size(), link_encoding, vtx,
                             n_vtx, idx, n_tri, materials, n_materials, material_ids, out);
        pt_bvh_destroy(b);
        return rc;
    }
    std::vector<pt_material> mats;
    std::vector<int32_t> mids;
    if (materials) {
        mats.assign(materials, materials + n_materials);
        mids.assign(material_ids, material_ids + n_tri);
        for (uint32_t t = 0; t < n_tri; ++t)
            if (mids[t] < 0 || (uint32_t)mids[t] >= n_materials)
                return fail(PT_ERR_INVALID, "pt_scene_build: material id " +
                                                std::to_string(mids[t]) + " out of range");
    } else {
        pt_material m{};
        m.kd[0] = m.kd[1] = m.kd[2] = 0.5f;
        m.roughness = 1.f;
        mats.push_back(m);
        mids.assign(n_tri, 0);
    }
    std::vector<int32_t> light_tri;
    std::vector<float> light_cdf;
    build_light_list(vtx, idx, n_tri, mats.data(), mids.data(), light_tri, light_cdf);

    HIP_CHECK(hipSetDevice(ctx->device));
    auto sc = std::make_unique<pt_scene>();
    sc->ctx = ctx;
    hipError_t e = hipSuccess;
    auto dev_alloc = [&](void **dst, size_t bytes) -> bool {
        e = scene_malloc(dst, bytes);
        if (e != hipSuccess) return false;
        sc->allocs.push_back(*dst);
        return true;
    };
    auto upload = [&](void **dst, const void *src, size_t bytes) -> bool {
        if (!dev_alloc(dst, bytes)) return false;
        e = hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream);
        return e == hipSuccess;
    };
    void *d_nodes, *d_vtx, *d_idx, *d_wn, *d_wt, *d_st, *d_mid, *d_mat, *d_lt = nullptr, *d_lc = nullptr;
    bool ok = upload(&d_vtx, vtx, sizeof(float) * 3 * (size_t)n_vtx) &&
              upload(&d_idx, idx, sizeof(int32_t) * 3 * (size_t)n_tri) &&
              upload(&d_mid, mids.data(), sizeof(int32_t) * mids.size()) &&
              upload(&d_mat, mats.data(), sizeof(pt_material) * mats.size());
    // Which triangles go into several leaves is decided before anything is sized: the arrays
    // below hold one record per leaf.
    DeviceBuffer first_leaf;
    uint32_t n_leaves = n_tri;
    if (ok) {
        const DeviceBuildResult pr = plan_triangle_splits(
            ctx->stream, (const float *)d_vtx, (const int32_t *)d_idx, n_tri, (float)ctx->build_split_ratio,
            ctx->build_split_max, 0.01f * (float)ctx->build_split_budget, first_leaf, n_leaves);
        if (pr.no_memory || pr.hip_error != hipSuccess) {
            (void)hipStreamSynchronize(ctx->stream);
            first_leaf.release();
            sc->release();
            if (pr.no_memory) return PT_ERR_DEVICE;
            return hip_fail(pr.hip_error, pr.what);
        }
        if (n_leaves > kMaxWideTris ||
            (link_encoding == PT_LINKS_FLOAT_VALUE && 2ull * n_leaves - 1 > (1u << 24))) {
            n_leaves = n_tri;                  // no room for the extra leaves: build without them
            first_leaf.release();
        }
    }
    const uint32_t n_nodes = 2 * n_leaves - 1;
    ok = ok && dev_alloc(&d_nodes, sizeof(pt_bbox) * (size_t)n_nodes) &&
         dev_alloc(&d_wn, sizeof(WideNode) * (size_t)(n_leaves - 1)) &&
         dev_alloc(&d_wt, sizeof(WideTriRecord) * (size_t)n_leaves) &&
         dev_alloc(&d_st, sizeof(TriRecord) * (size_t)n_tri);
    if (ok && !light_tri.empty())
        ok = upload(&d_lt, light_tri.data(), sizeof(int32_t) * light_tri.size()) &&
             upload(&d_lc, light_cdf.data(), sizeof(float) * light_cdf.size());
    if (!ok) {
        (void)hipStreamSynchronize(ctx->stream);
        first_leaf.release();
        sc->release();
        return hip_fail(e, "pt_scene_build allocation / upload");
    }
    DeviceBuffer scratch;
    const DeviceBuildResult br = build_scene_on_device(
        ctx->stream, (const float *)d_vtx, (const int32_t *)d_idx, n_tri, n_leaves,
        (const uint32_t *)first_leaf.ptr, link_encoding, scratch,
        (const int32_t *)d_mid, (float4 *)d_nodes, (float4 *)d_wn, (float4 *)d_wt, (float4 *)d_st,
        ctx->build_ploc_radius, ctx->build_optimize_passes, ctx->build_treelet_every,
        ctx->build_max_levels);
    (void)hipStreamSynchronize(ctx->stream);
    scratch.release();
    first_leaf.release();
    if (br.no_memory || br.hip_error != hipSuccess || br.bad_input) {
        sc->release();
        if (br.bad_input)
            return fail(PT_ERR_INVALID, "pt_scene_build: a triangle has a vertex that is not finite");
        if (br.no_memory) return PT_ERR_DEVICE;      // message set by the failing reserve
        return hip_fail(br.hip_error, br.what);
    }
    DeviceScene &d = sc->dev;
    d.nodes = (const float4 *)d_nodes;
    d.vtx = (const float *)d_vtx;
    d.idx = (const int32_t *)d_idx;
    d.wnodes = (const float4 *)d_wn;
    d.wtris = (const float4 *)d_wt;
    d.stris = (const float4 *)d_st;
    d.mat_ids = (const int32_t *)d_mid;
    d.materials = (const pt_material *)d_mat;
    d.light_tri = (const int32_t *)d_lt;
    d.light_cdf = (const float *)d_lc;
    d.n_lights = (uint32_t)light_tri.size();
    d.light_area = light_cdf.empty() ? 0.f : light_cdf.back();
    d.n_nodes = n_nodes;
    d.n_tri = n_tri;
    d.n_wnodes = n_leaves - 1;
    d.n_materials = (uint32_t)mats.size();
    d.link_enc = link_encoding;
    d.wide_depth = br.max_depth;
    d.big = needs_big_addresses(n_leaves - 1, n_leaves) ? 1u : 0u;
    {
        int rc = attach_top_cache(ctx, sc.get());
        if (rc == PT_OK) rc = attach_tri_leaf(ctx, sc.get());
        if (rc == PT_OK) rc = attach_quad(ctx, sc.get());
        if (rc != PT_OK) return rc;
    }
    ensure_stack_spill(ctx, d.wide_depth, false);
    *out = sc.release();
    return PT_OK;
    PT_CATCH
}

int pt_scene_set_face_ids(pt_scene *scene, const int32_t *shape_ids, const int32_t *prim_ids,
                          uint32_t n_triangles) {
    if (!scene) return fail(PT_ERR_INVALID, "pt_scene_set_face_ids: null scene");
    if (shape_ids && n_triangles != scene->dev.n_tri)
        return fail(PT_ERR_INVALID, "pt_scene_set_face_ids: " + std::to_string(n_triangles) +
                                        " ids for a scene of " + std::to_string(scene->dev.n_tri) +
                                        " triangles");
    if ((shape_ids == nullptr) != (prim_ids == nullptr))
        return fail(PT_ERR_INVALID, "pt_scene_set_face_ids: shape_ids and prim_ids must be given together");
    if (scene->dev.tlas)
        return fail(PT_ERR_UNSUPPORTED, "pt_scene_set_face_ids: a two-level scene reports the "
                                        "instance and the triangle of its mesh");
    PT_TRY
    pt_context *ctx = scene->ctx;
    HIP_CHECK(hipSetDevice(ctx->device));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));       // queries in flight read the old table
    if (scene->face_ids) {
        auto &a = scene->allocs;
        a.erase(std::remove(a.begin(), a.end(), scene->face_ids), a.end());
        (void)hipFree(scene->face_ids);
        scene->face_ids = nullptr;
        scene->dev.face_ids = nullptr;
    }
    if (!shape_ids) return PT_OK;
    const size_t n = scene->dev.n_tri;
    std::vector<int32_t> pairs(2 * n);
    for (size_t t = 0; t < n; ++t) {
        pairs[2 * t] = shape_ids[t];
        pairs[2 * t + 1] = prim_ids[t];
    }
    void *buf = nullptr;
    HIP_CHECK(hipMalloc(&buf, 8 * n));
    const hipError_t e = hipMemcpy(buf, pairs.data(), 8 * n, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(buf);
        return hip_fail(e, "pt_scene_set_face_ids upload");
    }
    scene->allocs.push_back(buf);
    scene->face_ids = buf;
    scene->dev.face_ids = (const int2 *)buf;
    return PT_OK;
    PT_CATCH
}

uint32_t pt_scene_node_count(const pt_scene *scene) { return scene ? scene->dev.n_nodes : 0; }
uint32_t pt_scene_triangle_count(const pt_scene *scene) { return scene ? scene->dev.n_tri : 0; }
uint32_t pt_scene_tree_depth(const pt_scene *scene) { return scene ? scene->dev.wide_depth : 0; }
uint32_t pt_scene_link_encoding(const pt_scene *scene) { return scene ? scene->dev.link_enc : 0; }

int pt_scene_read_nodes(pt_scene *scene, pt_bbox *nodes) {
    if (!scene || !nodes) return fail(PT_ERR_INVALID, "pt_scene_read_nodes: null argument");
    if (scene->dev.tlas)
        return fail(PT_ERR_UNSUPPORTED, "pt_scene_read_nodes: a two-level scene holds the trees its "
                                        "creator passed in; there is no single array to read");
    HIP_CHECK(hipSetDevice(scene->ctx->device));
    HIP_CHECK(hipMemcpyAsync(nodes, scene->dev.nodes, sizeof(pt_bbox) * (size_t)scene->dev.n_nodes,
                             hipMemcpyDeviceToHost, scene->ctx->stream));
    HIP_CHECK(hipStreamSynchronize(scene->ctx->stream));
    return PT_OK;
}

int pt_scene_destroy(pt_scene *scene) {
    if (!scene) return PT_OK;
    (void)hipSetDevice(scene->ctx->device);
    (void)hipStreamSynchronize(scene->ctx->stream);
    scene->release();
    delete scene;
    return PT_OK;
}

/* ---------------------------------------------------------------------- */

static int trace_closest_impl(pt_context *ctx, const pt_scene *scene, const pt_ray *rays_dev,
                              uint64_t n, pt_hit *hits_dev, uint32_t flags, bool count,
                              pt_trace_stats *stats) {
    if (n > PT_TRACE_MAX_RAYS)
        return fail(PT_ERR_INVALID, "pt_trace_closest: more than 2^31-64 rays in one call");
    HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const bool exact = (flags & (PT_TRACE_EXACT_ORDER | PT_TRACE_MOLLER_TRUMBORE)) != 0;
    const bool mt = (flags & PT_TRACE_MOLLER_TRUMBORE) != 0;
    const bool fixup = (flags & PT_TRACE_NO_TIE_FIXUP) == 0;
    // pt_context_last_trace_ms answers for this call only once both of its events are recorded
    ctx->trace_timed = false;
    if (scene->dev.tlas && mt)
        return fail(PT_ERR_UNSUPPORTED,
                    "pt_trace_closest: PT_TRACE_MOLLER_TRUMBORE is not available on two-level scenes");
    HIP_CHECK(hipMemsetAsync(ctx->counters, 0, sizeof(DeviceTraceCounters), st));
    hipEvent_t *bracket = nullptr;
    if (ctx->kernel_timing && n) {
        for (hipEvent_t &e : ctx->trace_events)
            if (!e) HIP_CHECK(hipEventCreate(&e));
        bracket = ctx->trace_events;
    }
    if (scene->dev.tlas) {
        if (bracket) HIP_CHECK(hipEventRecord(bracket[0], st));
        launch_trace_two_level(st, scene->dev, rays_dev, hits_dev, n, fixup, exact, ctx->counters,
                               count, make_extend_config(ctx, scene, false));
        if (bracket) HIP_CHECK(hipEventRecord(bracket[1], st));
    } else if (exact) {
        if (bracket) HIP_CHECK(hipEventRecord(bracket[0], st));
        launch_trace_skiplink(st, scene->dev, rays_dev, hits_dev, n, mt,
                              count ? ctx->counters : nullptr);
        if (bracket) HIP_CHECK(hipEventRecord(bracket[1], st));
    } else {
        // Taking the rays in origin-cell order pays once the traversal layouts no longer fit
        // the 256 MB Infinity Cache and the call is large enough to amortise the sort.
        const uint64_t scene_bytes = 64ull * scene->dev.n_wnodes + 48ull * scene->dev.n_tri;
        const bool sort = ctx->trace_sort_rays >= 0 ? ctx->trace_sort_rays != 0
                                                    : (scene_bytes > (256ull << 20) && n >= (1u << 18));
        ExtendConfig cfg = make_extend_config(ctx, scene, sort);
        cfg.sort_rays = sort ? 1 : 0;
        launch_trace_ordered(st, scene->dev, rays_dev, hits_dev, n, fixup, ctx->counters, count, cfg,
                             &ctx->sort_scratch, bracket);
    }
    HIP_CHECK(hipGetLastError());
    ctx->trace_timed = bracket != nullptr;
    launch_map_face_ids(st, hits_dev, n, scene->dev.face_ids);
    HIP_CHECK(hipGetLastError());
    if (stats) {
        DeviceTraceCounters c;
        HIP_CHECK(hipMemcpyAsync(&c, ctx->counters, sizeof(c), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        stats->rays = n;
        stats->box_tests = c.box_tests;
        stats->tri_tests = c.tri_tests;
        stats->tie_rays = c.tie_count;
        stats->lds_box_tests = c.lds_box_tests;
        stats->long_rays = c.long_rays;
    }
    return PT_OK;
}

int pt_trace_closest_device(pt_context *ctx, const pt_scene *scene, const pt_ray *rays_dev,
                            uint64_t n, pt_hit *hits_dev, uint32_t flags) {
    if (!ctx || !scene || (n && (!rays_dev || !hits_dev)))
        return fail(PT_ERR_INVALID, "pt_trace_closest_device: null argument");
    if (scene->ctx != ctx) return fail(PT_ERR_INVALID, "scene belongs to another context");
    PT_TRY
    return trace_closest_impl(ctx, scene, rays_dev, n, hits_dev, flags, false, nullptr);
    PT_CATCH
}

int pt_trace_closest_device_static(pt_context *ctx, const pt_scene *scene, const pt_ray *rays_dev,
                                   uint64_t n, pt_hit *hits_dev, uint32_t flags) {
    if (!ctx || !scene || (n && (!rays_dev || !hits_dev)))
        return fail(PT_ERR_INVALID, "pt_trace_closest_device_static: null argument");
    if (scene->ctx != ctx) return fail(PT_ERR_INVALID, "scene belongs to another context");
    if (n > PT_TRACE_MAX_RAYS)
        return fail(PT_ERR_INVALID, "pt_trace_closest_device_static: " + std::to_string(n) +
                                        " rays exceed the maximum of 2^31-64 per call");
    if ((flags & PT_TRACE_MOLLER_TRUMBORE) && scene->dev.tlas)
        return fail(PT_ERR_UNSUPPORTED, "pt_trace_closest_device_static: PT_TRACE_MOLLER_TRUMBORE "
                                        "is not available on two-level scenes");
    PT_TRY
    ctx->trace_timed = false;          // this call records no events: no time to report for it
    HIP_CHECK(hipSetDevice(ctx->device));
    const int rc = launch_trace_static(ctx->stream, ctx->device, scene->dev, rays_dev, hits_dev, n,
                                       flags, make_extend_config(ctx, scene, false));
    if (rc != PT_OK) return rc;
    HIP_CHECK(hipGetLastError());
    return PT_OK;
    PT_CATCH
}

int pt_trace_closest_device_counted(pt_context *ctx, const pt_scene *scene,
                                    const pt_ray *rays_dev, uint64_t n, pt_hit *hits_dev,
                                    uint32_t flags, pt_trace_stats *stats) {
    if (!ctx || !scene || !stats || (n && (!rays_dev || !hits_dev)))
        return fail(PT_ERR_INVALID, "pt_trace_closest_device_counted: null argument");
    if (scene->ctx != ctx) return fail(PT_ERR_INVALID, "scene belongs to another context");
    PT_TRY
    return trace_closest_impl(ctx, scene, rays_dev, n, hits_dev, flags, true, stats);
    PT_CATCH
}

int pt_trace_closest(pt_context *ctx, const pt_scene *scene, const pt_ray *rays, uint64_t n,
                     pt_hit *hits, uint32_t flags, pt_trace_stats *stats) {
    if (!ctx || !scene || (n && (!rays || !hits)))
        return fail(PT_ERR_INVALID, "pt_trace_closest: null argument");
    if (scene->ctx != ctx) return fail(PT_ERR_INVALID, "scene belongs to another context");
    PT_TRY
    HIP_CHECK(hipSetDevice(ctx->device));
    if (n == 0) {
        if (stats) *stats = pt_trace_stats{};
        return PT_OK;
    }
    int rc = ctx->io_rays.reserve(sizeof(pt_ray) * n);
    if (rc != PT_OK) return rc;
    rc = ctx->io_hits.reserve(sizeof(pt_hit) * n);
    if (rc != PT_OK) return rc;
    HIP_CHECK(hipMemcpyAsync(ctx->io_rays.ptr, rays, sizeof(pt_ray) * n, hipMemcpyHostToDevice,
                             ctx->stream));
    rc = trace_closest_impl(ctx, scene, (const pt_ray *)ctx->io_rays.ptr, n,
                            (pt_hit *)ctx->io_hits.ptr, flags, stats != nullptr, stats);
    if (rc != PT_OK) return rc;
    HIP_CHECK(hipMemcpyAsync(hits, ctx->io_hits.ptr, sizeof(pt_hit) * n, hipMemcpyDeviceToHost,
                             ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PT_OK;
    PT_CATCH
}

int pt_trace_any_device(pt_context *ctx, const pt_scene *scene, const pt_ray *rays_dev, uint64_t n,
                        int32_t *occluded_dev) {
    if (!ctx || !scene || (n && (!rays_dev || !occluded_dev)))
        return fail(PT_ERR_INVALID, "pt_trace_any_device: null argument");
    if (scene->ctx != ctx) return fail(PT_ERR_INVALID, "scene belongs to another context");
    HIP_CHECK(hipSetDevice(ctx->device));
    if (n > PT_TRACE_MAX_RAYS)
        return fail(PT_ERR_INVALID, "pt_trace_any: more than 2^31-64 rays in one call");
    launch_trace_any(ctx->stream, scene->dev, rays_dev, occluded_dev, n, ctx->any_exact_order != 0,
                     make_extend_config(ctx, scene, false));
    HIP_CHECK(hipGetLastError());
    return PT_OK;
}

int pt_trace_any(pt_context *ctx, const pt_scene *scene, const pt_ray *rays, uint64_t n,
                 int32_t *occluded) {
    if (!ctx || !scene || (n && (!rays || !occluded)))
        return fail(PT_ERR_INVALID, "pt_trace_any: null argument");
    if (scene->ctx != ctx) return fail(PT_ERR_INVALID, "scene belongs to another context");
    if (n == 0) return PT_OK;
    if (n > PT_TRACE_MAX_RAYS)
        return fail(PT_ERR_INVALID, "pt_trace_any: more than 2^31-64 rays in one call");
    HIP_CHECK(hipSetDevice(ctx->device));
    int rc = ctx->io_rays.reserve(sizeof(pt_ray) * n);
    if (rc != PT_OK) return rc;
    rc = ctx->io_hits.reserve(sizeof(int32_t) * n);
    if (rc != PT_OK) return rc;
    HIP_CHECK(hipMemcpyAsync(ctx->io_rays.ptr, rays, sizeof(pt_ray) * n, hipMemcpyHostToDevice,
                             ctx->stream));
    launch_trace_any(ctx->stream, scene->dev, (const pt_ray *)ctx->io_rays.ptr,
                     (int32_t *)ctx->io_hits.ptr, n, ctx->any_exact_order != 0, make_extend_config(ctx, scene, false));
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(occluded, ctx->io_hits.ptr, sizeof(int32_t) * n,
                             hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

/* ---------------------------------------------------------------------- */

static int check_camera(const pt_camera *cam, uint32_t width, uint32_t height) {
    if (!cam) return fail(PT_ERR_INVALID, "null camera");
    if (cam->mode != PT_CAMERA_REFERENCE && cam->mode != PT_CAMERA_PINHOLE)
        return fail(PT_ERR_INVALID, "unknown camera mode " + std::to_string(cam->mode));
    if (width == 0 || height == 0 || (uint64_t)width * height > 0xFFFFFFF0ull)
        return fail(PT_ERR_INVALID, "image size out of range");
    return PT_OK;
}

int pt_raygen_device(pt_context *ctx, const pt_camera *cam, uint32_t width, uint32_t height,
                     pt_ray *rays_dev) {
    if (!ctx || !rays_dev) return fail(PT_ERR_INVALID, "pt_raygen_device: null argument");
    int rc = check_camera(cam, width, height);
    if (rc != PT_OK) return rc;
    HIP_CHECK(hipSetDevice(ctx->device));
    launch_raygen(ctx->stream, *cam, width, height, rays_dev);
    HIP_CHECK(hipGetLastError());
    return PT_OK;
}

int pt_raygen(pt_context *ctx, const pt_camera *cam, uint32_t width, uint32_t height,
              pt_ray *rays) {
    if (!ctx || !rays) return fail(PT_ERR_INVALID, "pt_raygen: null argument");
    int rc = check_camera(cam, width, height);
    if (rc != PT_OK) return rc;
    HIP_CHECK(hipSetDevice(ctx->device));
    const uint64_t n = (uint64_t)width * height;
    rc = ctx->io_rays.reserve(sizeof(pt_ray) * n);
    if (rc != PT_OK) return rc;
    launch_raygen(ctx->stream, *cam, width, height, (pt_ray *)ctx->io_rays.ptr);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(rays, ctx->io_rays.ptr, sizeof(pt_ray) * n, hipMemcpyDeviceToHost,
                             ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_depth_shade(pt_context *ctx, const pt_hit *hits, uint64_t n, float *rgba) {
    if (!ctx || (n && (!hits || !rgba))) return fail(PT_ERR_INVALID, "pt_depth_shade: null argument");
    if (n == 0) return PT_OK;
    HIP_CHECK(hipSetDevice(ctx->device));
    int rc = ctx->io_hits.reserve(sizeof(pt_hit) * n);
    if (rc != PT_OK) return rc;
    rc = ctx->io_rays.reserve(sizeof(float) * 4 * n);
    if (rc != PT_OK) return rc;
    HIP_CHECK(hipMemcpyAsync(ctx->io_hits.ptr, hits, sizeof(pt_hit) * n, hipMemcpyHostToDevice,
                             ctx->stream));
    launch_depth_shade(ctx->stream, (const pt_hit *)ctx->io_hits.ptr, n, (float *)ctx->io_rays.ptr);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(rgba, ctx->io_rays.ptr, sizeof(float) * 4 * n, hipMemcpyDeviceToHost,
                             ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

} // extern "C"
