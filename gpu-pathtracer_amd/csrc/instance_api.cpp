// C ABI: two-level scenes (meshes placed by affine transforms under a top-level BVH).
//
// The reference documents the format (src/bvh/bvh_builder.h:29-33: a top-level BVH whose leaf
// payload is an instance index) but never builds one: its two-level translator is disabled
// (src/bvh/RadeonRays/plain_bvh_translator.cpp:98-101,134-137) and the application draws one
// mesh (src/main_granite.cpp:284-298).  The caller builds every tree with pt_bvh_build, the
// same way the reference builds its single one: one per mesh over the triangle boxes, one over
// the instances' world boxes (pt_instance_bounds).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mipt.h"
#include "../host/bvh.h"
#include "build.h"
#include "internal.h"

using namespace mipt;

namespace {

int check_meshes(const pt_mesh_desc *meshes, uint32_t n_meshes, const pt_instance *inst,
                 uint32_t n_inst, const char *who, bool need_nodes) {
    if (!meshes || !inst) return fail(PT_ERR_INVALID, std::string(who) + ": null argument");
    if (n_meshes == 0 || n_inst == 0)
        return fail(PT_ERR_INVALID, std::string(who) + ": no meshes or no instances");
    for (uint32_t m = 0; m < n_meshes; ++m) {
        const pt_mesh_desc &d = meshes[m];
        if (!d.vertices || !d.indices || d.n_vertices == 0 || d.n_triangles == 0 ||
            (d.nodes == nullptr) != (d.n_nodes == 0) || (need_nodes && !d.nodes))
            return fail(PT_ERR_INVALID, std::string(who) + ": mesh " + std::to_string(m) +
                                            " is empty or has a null array");
    }
    for (uint32_t i = 0; i < n_inst; ++i) {
        if (inst[i].mesh >= n_meshes)
            return fail(PT_ERR_INVALID, std::string(who) + ": instance " + std::to_string(i) +
                                            " names mesh " + std::to_string(inst[i].mesh));
        for (int k = 0; k < 12; ++k)
            if (!std::isfinite(inst[i].object_to_world[k]))
                return fail(PT_ERR_INVALID, std::string(who) + ": instance " + std::to_string(i) +
                                                " has a non-finite transform");
    }
    return PT_OK;
}

} // namespace

extern "C" {

int pt_instance_bounds(const pt_mesh_desc *meshes, uint32_t n_meshes, const pt_instance *instances,
                       uint32_t n_instances, pt_bbox *out) {
    if (!out) return fail(PT_ERR_INVALID, "pt_instance_bounds: null argument");
    const int rc = check_meshes(meshes, n_meshes, instances, n_instances, "pt_instance_bounds", true);
    if (rc != PT_OK) return rc;
    for (uint32_t i = 0; i < n_instances; ++i)
        instance_world_box(instances[i].object_to_world, meshes[instances[i].mesh].nodes[0], out[i]);
    return PT_OK;
}

int pt_scene_create_instanced(pt_context *ctx, const pt_mesh_desc *meshes, uint32_t n_meshes,
                              const pt_instance *instances, uint32_t n_instances,
                              const pt_bbox *top_nodes, uint32_t n_top_nodes,
                              uint32_t link_encoding, const pt_material *materials,
                              uint32_t n_materials, pt_scene **out) {
    if (!ctx || !out || (top_nodes == nullptr) != (n_top_nodes == 0))
        return fail(PT_ERR_INVALID, "pt_scene_create_instanced: null argument");
    *out = nullptr;
    if (link_encoding != PT_LINKS_FLOAT_VALUE && link_encoding != PT_LINKS_INT_BITS)
        return fail(PT_ERR_INVALID, "pt_scene_create_instanced: unknown link encoding");
    int rc = check_meshes(meshes, n_meshes, instances, n_instances, "pt_scene_create_instanced", false);
    if (rc != PT_OK) return rc;
    if (materials && n_materials == 0)
        return fail(PT_ERR_INVALID, "pt_scene_create_instanced: materials given but n_materials is 0");
    try {
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess) return hip_fail(e, "hipSetDevice");

        std::vector<pt_material> mats;
        if (materials) {
            mats.assign(materials, materials + n_materials);
        } else {
            pt_material m{};
            m.kd[0] = m.kd[1] = m.kd[2] = 0.5f;
            m.roughness = 1.f;
            mats.push_back(m);
        }

        // Meshes, back to back.  One whose tree is given is validated and re-laid-out on the
        // host like a scene of its own; one without (nodes == NULL) is built on the GPU below.
        struct MeshPlan {
            bool on_gpu = false;
            std::vector<pt_bbox> own_nodes;      // host-built tree of a one-triangle mesh
            const pt_bbox *nodes = nullptr;      // host copy of the skip-link tree, if any
            uint32_t n_nodes = 0, n_wnodes = 0, n_wtris = 0;
            uint32_t n_leaves = 0;               // on_gpu: leaves of the tree to build (>= triangles)
            DeviceBuffer first_leaf;             // on_gpu: first leaf of each triangle, if any is cut
            WideBvh wide;
            uint32_t node = 0, wnode = 0, wtri = 0, vtx = 0, tri = 0;   // offsets
            pt_bbox root{};
        };
        std::vector<MeshPlan> plan(n_meshes);
        uint32_t wide_depth = 0;
        uint64_t tot_nodes = 0, tot_wnodes = 0, tot_wtris = 0, tot_vtx = 0, tot_tri = 0;
        for (uint32_t m = 0; m < n_meshes; ++m) {
            const pt_mesh_desc &d = meshes[m];
            MeshPlan &pl = plan[m];
            const std::string tag = "pt_scene_create_instanced: mesh " + std::to_string(m) + ": ";
            for (uint64_t i = 0; i < 3ull * d.n_triangles; ++i)
                if (d.indices[i] < 0 || (uint32_t)d.indices[i] >= d.n_vertices)
                    return fail(PT_ERR_INVALID, tag + "index " + std::to_string(d.indices[i]) +
                                                    " outside the vertex buffer");
            pl.nodes = d.nodes;
            pl.n_nodes = d.n_nodes;
            if (!d.nodes && d.n_triangles == 1) {            // a one-leaf tree: nothing to build
                pl.own_nodes.resize(1);
                mesh_leaf_bounds(d.vertices, d.indices, 1, pl.own_nodes.data());
                inflate_leaf_bounds(pl.own_nodes.data(), 1);
                BvhTree one;
                pt_bvh_options o;
                pt_bvh_options_default(&o);
                build_tree(pl.own_nodes.data(), 1, o, one);
                linearize_skip_links(one, link_encoding, pl.own_nodes);
                pl.nodes = pl.own_nodes.data();
                pl.n_nodes = (uint32_t)pl.own_nodes.size();
            }
            if (pl.nodes) {
                BvhTree tree;
                try {
                    tree_from_skip_links(pl.nodes, pl.n_nodes, link_encoding, d.n_triangles, tree);
                } catch (const std::runtime_error &err) {
                    return fail(PT_ERR_INVALID, tag + err.what());
                }
                std::vector<uint8_t> seen(d.n_triangles, 0);
                for (size_t i = 0; i < tree.size(); ++i)
                    if (tree.prim[i] >= 0) seen[(size_t)tree.prim[i]] = 1;
                for (uint32_t k = 0; k < d.n_triangles; ++k)
                    if (!seen[k])
                        return fail(PT_ERR_INVALID, tag + "triangle " + std::to_string(k) + " is in no leaf");
                build_wide_layout(tree, d.vertices, d.n_vertices, d.indices, d.n_triangles, pl.wide);
                pl.n_wnodes = (uint32_t)pl.wide.nodes.size();
                pl.n_wtris = (uint32_t)pl.wide.tris.size();
                pl.root = pl.nodes[0];
                if (pl.wide.max_depth > wide_depth) wide_depth = pl.wide.max_depth;
            } else {
                if (link_encoding == PT_LINKS_FLOAT_VALUE && 2ull * d.n_triangles - 1 > (1u << 24))
                    return fail(PT_ERR_INVALID, tag + "more than 2^24 nodes need PT_LINKS_INT_BITS");
                pl.on_gpu = true;
                pl.n_leaves = d.n_triangles;
                if (ctx->build_split_ratio > 0) {
                    // which triangles go into several leaves decides the sizes below: counted on
                    // a device copy of the mesh made for that alone
                    DeviceBuffer tv, ti;
                    const size_t vb = sizeof(float) * 3 * (size_t)d.n_vertices;
                    const size_t ib = sizeof(int32_t) * 3 * (size_t)d.n_triangles;
                    if (tv.reserve(vb, true) != PT_OK || ti.reserve(ib, true) != PT_OK) return PT_ERR_DEVICE;
                    hipError_t ce = hipMemcpy(tv.ptr, d.vertices, vb, hipMemcpyHostToDevice);
                    if (ce == hipSuccess) ce = hipMemcpy(ti.ptr, d.indices, ib, hipMemcpyHostToDevice);
                    if (ce != hipSuccess) return hip_fail(ce, "pt_scene_create_instanced upload");
                    const DeviceBuildResult pr = plan_triangle_splits(
                        ctx->stream, (const float *)tv.ptr, (const int32_t *)ti.ptr, d.n_triangles,
                        (float)ctx->build_split_ratio, ctx->build_split_max,
                        0.01f * (float)ctx->build_split_budget, pl.first_leaf, pl.n_leaves);
                    if (pr.no_memory) return PT_ERR_DEVICE;
                    if (pr.hip_error != hipSuccess) return hip_fail(pr.hip_error, pr.what);
                    if (link_encoding == PT_LINKS_FLOAT_VALUE && 2ull * pl.n_leaves - 1 > (1u << 24)) {
                        pl.n_leaves = d.n_triangles;         // no room for the extra leaves
                        pl.first_leaf.release();
                    }
                }
                pl.n_nodes = 2 * pl.n_leaves - 1;
                pl.n_wnodes = pl.n_leaves - 1;
                pl.n_wtris = pl.n_leaves;
            }
            pl.node = (uint32_t)tot_nodes; pl.wnode = (uint32_t)tot_wnodes; pl.wtri = (uint32_t)tot_wtris;
            pl.vtx = (uint32_t)tot_vtx; pl.tri = (uint32_t)tot_tri;
            tot_nodes += pl.n_nodes; tot_wnodes += pl.n_wnodes; tot_wtris += pl.n_wtris;
            tot_vtx += d.n_vertices; tot_tri += d.n_triangles;
            if (tot_wnodes > kMaxWideNodes || tot_wtris > kMaxWideTris)
                return fail(PT_ERR_UNSUPPORTED,
                            "pt_scene_create_instanced: more nodes or triangle records than a child index can name");
            if (tot_nodes > 0x7fffffffull || tot_vtx > 0x7fffffffull)
                return fail(PT_ERR_UNSUPPORTED, "pt_scene_create_instanced: meshes too large together");
        }

        auto sc = std::make_unique<pt_scene>();
        sc->ctx = ctx;
        auto dev_alloc = [&](size_t bytes) -> void * {
            void *dst = nullptr;
            const hipError_t err = hipMalloc(&dst, bytes ? bytes : 16);
            if (err != hipSuccess) {
                sc->release();
                throw std::runtime_error(std::string("pt_scene_create_instanced allocation: ") +
                                         hipGetErrorString(err));
            }
            sc->allocs.push_back(dst);
            return dst;
        };
        auto copy_in = [&](void *dst, const void *src, size_t bytes) {
            const hipError_t err = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
            if (err != hipSuccess) {
                sc->release();
                throw std::runtime_error(std::string("pt_scene_create_instanced upload: ") +
                                         hipGetErrorString(err));
            }
        };
        auto upload = [&](const void *src, size_t bytes) -> void * {
            void *dst = dev_alloc(bytes);
            copy_in(dst, src, bytes);
            return dst;
        };
        char *d_nodes = (char *)dev_alloc(sizeof(pt_bbox) * (size_t)tot_nodes);
        char *d_vtx = (char *)dev_alloc(sizeof(float) * 3 * (size_t)tot_vtx);
        char *d_idx = (char *)dev_alloc(sizeof(int32_t) * 3 * (size_t)tot_tri);
        // the two-child nodes of the top level go behind those of the meshes, into the same index
        // space: the voted loop then steps through both levels with one piece of code
        const size_t top_room = top_nodes ? (size_t)n_top_nodes : 2 * (size_t)n_instances;
        if (tot_wnodes + top_room > kMaxWideNodes)
            return fail(PT_ERR_UNSUPPORTED,
                        "pt_scene_create_instanced: more nodes than a child index can name");
        char *d_wnodes = (char *)dev_alloc(sizeof(WideNode) * ((size_t)tot_wnodes + top_room));
        char *d_wtris = (char *)dev_alloc(sizeof(WideTriRecord) * (size_t)tot_wtris);
        sc->mesh_node_off.resize(n_meshes);
        sc->mesh_node_count.resize(n_meshes);
        DeviceBuffer scratch;
        for (uint32_t m = 0; m < n_meshes; ++m) {
            const pt_mesh_desc &d = meshes[m];
            MeshPlan &pl = plan[m];
            sc->mesh_node_off[m] = pl.node;
            sc->mesh_node_count[m] = pl.n_nodes;
            float *mv = (float *)(d_vtx + sizeof(float) * 3 * (size_t)pl.vtx);
            int32_t *mi = (int32_t *)(d_idx + sizeof(int32_t) * 3 * (size_t)pl.tri);
            float4 *mn = (float4 *)(d_nodes + sizeof(pt_bbox) * (size_t)pl.node);
            float4 *mw = (float4 *)(d_wnodes + sizeof(WideNode) * (size_t)pl.wnode);
            float4 *mt = (float4 *)(d_wtris + sizeof(WideTriRecord) * (size_t)pl.wtri);
            copy_in(mv, d.vertices, sizeof(float) * 3 * (size_t)d.n_vertices);
            copy_in(mi, d.indices, sizeof(int32_t) * 3 * (size_t)d.n_triangles);
            if (!pl.on_gpu) {
                // child indices become scene-wide: an inner child is the index of a two-child
                // node, a leaf child the complement of its record's position in 16-byte units
                const int32_t node_base = (int32_t)pl.wnode, tri_base = PT_TRI_RECORD_16B * (int32_t)pl.wtri;
                for (WideNode &w : pl.wide.nodes)
                    for (int c = 0; c < 2; ++c)
                        w.child[c] = w.child[c] >= 0 ? w.child[c] + node_base : ~(~w.child[c] + tri_base);
                copy_in(mn, pl.nodes, sizeof(pt_bbox) * (size_t)pl.n_nodes);
                copy_in(mw, pl.wide.nodes.data(), sizeof(WideNode) * pl.wide.nodes.size());
                copy_in(mt, pl.wide.tris.data(), sizeof(WideTriRecord) * pl.wide.tris.size());
                pl.wide = WideBvh();
                continue;
            }
            {   // the uploads above went through the null stream
                const hipError_t se = hipDeviceSynchronize();
                if (se != hipSuccess) {
                    sc->release();
                    return hip_fail(se, "pt_scene_create_instanced");
                }
            }
            const DeviceBuildResult br =
                build_scene_on_device(ctx->stream, mv, mi, d.n_triangles, pl.n_leaves,
                                      (const uint32_t *)pl.first_leaf.ptr, link_encoding, scratch,
                                      nullptr, mn, mw, mt, nullptr, ctx->build_ploc_radius,
                                      ctx->build_optimize_passes, ctx->build_treelet_every,
                                      ctx->build_max_levels);
            hipError_t err = br.hip_error;
            if (err == hipSuccess && !br.no_memory && !br.bad_input) {
                launch_rebase_children(ctx->stream, mw, pl.n_wnodes, pl.wnode, pl.wtri);
                err = hipGetLastError();
                if (err == hipSuccess)
                    err = hipMemcpyAsync(&pl.root, mn, sizeof(pt_bbox), hipMemcpyDeviceToHost, ctx->stream);
                if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
            }
            if (br.no_memory || br.bad_input || err != hipSuccess) {
                (void)hipStreamSynchronize(ctx->stream);
                scratch.release();
                sc->release();
                if (br.bad_input)
                    return fail(PT_ERR_INVALID, "pt_scene_create_instanced: mesh " + std::to_string(m) +
                                                    " has a vertex that is not finite");
                if (br.no_memory) return PT_ERR_DEVICE;       // message set by the failing reserve
                return hip_fail(err, br.hip_error != hipSuccess ? br.what : "pt_scene_create_instanced build");
            }
            if (br.max_depth > wide_depth) wide_depth = br.max_depth;
        }
        scratch.release();

        // Top level: a skip-link tree whose leaves name every instance; built here (same
        // builder and inflation as pt_bvh_build, default options) when the caller gives none.
        std::vector<pt_bbox> own_top;
        if (!top_nodes) {
            std::vector<pt_bbox> lb(n_instances);
            for (uint32_t i = 0; i < n_instances; ++i)
                instance_world_box(instances[i].object_to_world, plan[instances[i].mesh].root, lb[i]);
            inflate_leaf_bounds(lb.data(), n_instances);
            pt_bvh_options o;
            pt_bvh_options_default(&o);
            o.use_sah = 1;
            BvhTree tt;
            build_tree(lb.data(), n_instances, o, tt);
            optimize_tree(tt, 2);
            linearize_skip_links(tt, link_encoding, own_top);
            top_nodes = own_top.data();
            n_top_nodes = (uint32_t)own_top.size();
        }
        BvhTree top;
        try {
            tree_from_skip_links(top_nodes, n_top_nodes, link_encoding, n_instances, top);
        } catch (const std::runtime_error &err) {
            sc->release();
            return fail(PT_ERR_INVALID, std::string("pt_scene_create_instanced: top level: ") + err.what());
        }
        {
            std::vector<uint8_t> seen(n_instances, 0);
            for (size_t i = 0; i < top.size(); ++i)
                if (top.prim[i] >= 0) seen[(size_t)top.prim[i]] = 1;
            for (uint32_t k = 0; k < n_instances; ++k)
                if (!seen[k]) {
                    sc->release();
                    return fail(PT_ERR_INVALID, "pt_scene_create_instanced: instance " +
                                                    std::to_string(k) + " is in no top-level leaf");
                }
        }
        struct MeshOffsets { uint32_t node, wnode, wtri, vtx, tri; };
        std::vector<MeshOffsets> off(n_meshes);
        for (uint32_t m = 0; m < n_meshes; ++m)
            off[m] = {plan[m].node, plan[m].wnode, plan[m].wtri, plan[m].vtx, plan[m].tri};

        // Instances.  Their world-space shading records are written on the GPU below; the host
        // only transforms the emissive triangles, for the light list.
        uint64_t total_tris = 0;
        uint32_t max_mesh_tris = 0;
        for (uint32_t i = 0; i < n_instances; ++i) total_tris += meshes[instances[i].mesh].n_triangles;
        if (total_tris > 0x7fffffffull)
            return sc->release(), fail(PT_ERR_UNSUPPORTED, "pt_scene_create_instanced: more than 2^31 instanced triangles");
        std::vector<int32_t> mesh_mids;                       // per mesh triangle, meshes back to back
        std::vector<std::vector<int32_t>> used_ids(n_meshes); // distinct material ids of each mesh
        bool any_ids = false;
        for (uint32_t m = 0; m < n_meshes; ++m) any_ids = any_ids || meshes[m].material_ids != nullptr;
        for (uint32_t m = 0; m < n_meshes; ++m) {
            const pt_mesh_desc &d = meshes[m];
            if (d.n_triangles > max_mesh_tris) max_mesh_tris = d.n_triangles;
            if (d.material_ids) {
                mesh_mids.insert(mesh_mids.end(), d.material_ids, d.material_ids + d.n_triangles);
                std::vector<int32_t> u(d.material_ids, d.material_ids + d.n_triangles);
                std::sort(u.begin(), u.end());
                u.erase(std::unique(u.begin(), u.end()), u.end());
                used_ids[m] = std::move(u);
            } else {
                if (any_ids) mesh_mids.insert(mesh_mids.end(), d.n_triangles, 0);
                used_ids[m] = {0};
            }
        }
        std::vector<DeviceInstance> dinst(n_instances);
        std::vector<int32_t> moffs(n_instances);
        std::vector<float> fwd(12 * (size_t)n_instances);
        std::vector<float> light_vtx;      // world-space vertices of the emissive triangles
        std::vector<int32_t> light_ids;    // their indices among all instanced triangles
        std::vector<int32_t> light_mids;
        uint32_t first = 0;
        for (uint32_t i = 0; i < n_instances; ++i) {
            const pt_instance &in = instances[i];
            const pt_mesh_desc &d = meshes[in.mesh];
            DeviceInstance &di = dinst[i];
            if (!invert_affine(in.object_to_world, di.inv))
                return sc->release(), fail(PT_ERR_INVALID, "pt_scene_create_instanced: the transform of instance " +
                                                std::to_string(i) + " is not invertible");
            di.node_off = off[in.mesh].node;
            di.wnode_off = off[in.mesh].wnode;
            di.wtri_off = off[in.mesh].wtri;
            di.vtx_off = off[in.mesh].vtx;
            di.tri_off = off[in.mesh].tri;
            di.n_tri = d.n_triangles;
            di.first_tri = first;
            {
                const pt_bbox &rb = plan[in.mesh].root;
                float c = 0.f;
                for (int k = 0; k < 3; ++k) c = std::max(c, std::max(std::fabs(rb.pmin[k]), std::fabs(rb.pmax[k])));
                di.cmax = c * 1.001f;
            }
            moffs[i] = in.material_offset;
            std::memcpy(&fwd[12 * (size_t)i], in.object_to_world, sizeof(float) * 12);
            bool emissive = false;
            for (int32_t id : used_ids[in.mesh]) {
                const int64_t mid = (int64_t)id + in.material_offset;
                if (mid < 0 || mid >= (int64_t)mats.size())
                    return sc->release(), fail(PT_ERR_INVALID, "pt_scene_create_instanced: instance " +
                                                    std::to_string(i) + ": material " +
                                                    std::to_string(mid) + " out of range");
                const pt_material &pm = mats[(size_t)mid];
                emissive = emissive || (pm.ke[0] + pm.ke[1]) + pm.ke[2] > 0.0f;
            }
            if (emissive) {
                for (uint32_t t = 0; t < d.n_triangles; ++t) {
                    const int32_t mid = (d.material_ids ? d.material_ids[t] : 0) + in.material_offset;
                    const pt_material &pm = mats[(size_t)mid];
                    if (!((pm.ke[0] + pm.ke[1]) + pm.ke[2] > 0.0f)) continue;
                    for (int j = 0; j < 3; ++j) {
                        float w[3];
                        transform_point(in.object_to_world,
                                        d.vertices + 3 * (size_t)d.indices[3 * (size_t)t + j], w);
                        light_vtx.insert(light_vtx.end(), w, w + 3);
                    }
                    light_ids.push_back((int32_t)(first + t));
                    light_mids.push_back(mid);
                }
            }
            first += d.n_triangles;
        }
        // Same list as a single-level scene of all instanced triangles would get: the
        // non-emissive ones contribute nothing to the running area sum.
        std::vector<int32_t> light_tri;
        std::vector<float> light_cdf;
        {
            std::vector<int32_t> seq(light_vtx.size() / 3);
            for (size_t k = 0; k < seq.size(); ++k) seq[k] = (int32_t)k;
            build_light_list(light_vtx.data(), seq.data(), (uint32_t)light_ids.size(), mats.data(),
                             light_mids.data(), light_tri, light_cdf);
            for (int32_t &t : light_tri) t = light_ids[(size_t)t];
        }

        DeviceScene &d = sc->dev;
        d.nodes = (const float4 *)d_nodes;
        d.vtx = (const float *)d_vtx;
        d.idx = (const int32_t *)d_idx;
        d.wnodes = (const float4 *)d_wnodes;
        d.wtris = (const float4 *)d_wtris;
        void *d_stris = nullptr;
        {
            const hipError_t err = hipMalloc(&d_stris, sizeof(TriRecord) * (size_t)total_tris);
            if (err != hipSuccess) {
                sc->release();
                return hip_fail(err, "pt_scene_create_instanced: shading records");
            }
            sc->allocs.push_back(d_stris);
        }
        d.stris = (const float4 *)d_stris;
        // per mesh triangle here (the shading records carry the per-instance ids)
        d.mat_ids = mesh_mids.empty() ? nullptr
                                      : (const int32_t *)upload(mesh_mids.data(), sizeof(int32_t) * mesh_mids.size());
        d.materials = (const pt_material *)upload(mats.data(), sizeof(pt_material) * mats.size());
        d.light_tri = nullptr;
        d.light_cdf = nullptr;
        if (!light_tri.empty()) {
            d.light_tri = (const int32_t *)upload(light_tri.data(), sizeof(int32_t) * light_tri.size());
            d.light_cdf = (const float *)upload(light_cdf.data(), sizeof(float) * light_cdf.size());
        }
        d.n_lights = (uint32_t)light_tri.size();
        d.light_area = light_cdf.empty() ? 0.f : light_cdf.back();
        d.n_nodes = (uint32_t)tot_nodes;
        d.n_tri = (uint32_t)total_tris;
        d.n_wnodes = (uint32_t)tot_wnodes;
        d.n_mesh_records = (uint32_t)tot_wtris;
        d.n_materials = (uint32_t)mats.size();
        d.link_enc = link_encoding;
        d.wide_depth = wide_depth;
        ensure_stack_spill(ctx, wide_depth, true);
        d.big = needs_big_addresses(tot_wnodes + top_room, tot_wtris) ? 1u : 0u;
        d.tlas = (const float4 *)upload(top_nodes, sizeof(pt_bbox) * (size_t)n_top_nodes);
        {
            std::vector<WideNode> wtop;
            build_wide_top(top, wtop);
            if (wtop.size() > top_room)
                return fail(PT_ERR_INVALID, "pt_scene_create_instanced: top-level tree larger than its node array");
            for (WideNode &w : wtop)
                for (int c = 0; c < 2; ++c)
                    if (w.child[c] >= 0) w.child[c] += (int32_t)tot_wnodes;     // a leaf stays ~instance
            char *dst = d_wnodes + sizeof(WideNode) * (size_t)tot_wnodes;
            copy_in(dst, wtop.data(), sizeof(WideNode) * wtop.size());
            d.wtop = (const float4 *)dst;
            d.top_root = (uint32_t)tot_wnodes;
        }
        if (n_instances >= 2) {
            // the top of the top-level tree, which every ray walks first, as the persistent
            // kernels copy it to LDS (same layout as a single-level scene's)
            const size_t node_bytes = 64 * (size_t)kNodeCacheMax, index_bytes = 4 * (size_t)kNodeCacheMax;
            char *buf = (char *)dev_alloc(node_bytes + index_bytes + 16);
            int32_t *index = reinterpret_cast<int32_t *>(buf + node_bytes);
            uint32_t *n_dev = reinterpret_cast<uint32_t *>(buf + node_bytes + index_bytes);
            hipError_t err = hipDeviceSynchronize();            // the uploads above used the null stream
            uint32_t n_cached = 0;
            if (err == hipSuccess) {
                launch_build_top_cache(ctx->stream, d.wnodes, (uint32_t)(tot_wnodes + top_room), (float4 *)buf, index,
                                       n_dev, d.top_root);
                err = hipMemcpyAsync(&n_cached, n_dev, 4, hipMemcpyDeviceToHost, ctx->stream);
            }
            if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
            if (err != hipSuccess) {
                sc->release();
                return hip_fail(err, "pt_scene_create_instanced: top-level cache");
            }
            d.top_cache = (const float4 *)buf;
            d.top_global = index;
            d.n_top_cache = n_cached;
        }
        // The meshes' trees as four-child nodes with their 64-byte triangle records, for scenes
        // whose mesh records come from memory rather than from the caches (option scene.quad; a
        // scene it cannot be made for keeps the two-child form only).
        if (ctx->scene_quad && tot_wnodes >= 1 && 4 * (uint64_t)tot_wtris < (1ull << 31)) {
            void *qn = nullptr, *qt = nullptr, *flag = nullptr;
            hipError_t err = hipDeviceSynchronize();            // the uploads above used the null stream
            if (err == hipSuccess) err = hipMalloc(&qn, 64 * (size_t)tot_wnodes);
            if (err == hipSuccess) err = hipMalloc(&qt, 64 * (size_t)tot_wtris);
            if (err == hipSuccess) err = hipMalloc(&flag, 16);
            uint32_t bad = 1;
            if (err == hipSuccess) err = hipMemsetAsync(flag, 0, 16, ctx->stream);
            if (err == hipSuccess) err = hipMemsetAsync(qt, 0, 64 * (size_t)tot_wtris, ctx->stream);
            if (err == hipSuccess) {
                launch_build_quad(ctx->stream, d.wnodes, (uint32_t)tot_wnodes, d.wtris, (uint32_t)tot_wtris,
                                  (float4 *)qn, (float4 *)qt, (uint32_t *)flag);
                err = hipMemcpyAsync(&bad, flag, 4, hipMemcpyDeviceToHost, ctx->stream);
            }
            if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
            if (flag) (void)hipFree(flag);
            if (err == hipSuccess && bad == 0) {
                sc->allocs.push_back(qn);
                sc->allocs.push_back(qt);
                d.qnodes = (const float4 *)qn;
                d.qtris = (const float4 *)qt;
                if (needs_big_addresses(tot_wnodes, 4 * (uint64_t)tot_wtris / PT_TRI_RECORD_16B + 1)) d.big = 1;
            } else {
                (void)hipGetLastError();
                if (qn) (void)hipFree(qn);
                if (qt) (void)hipFree(qt);
            }
        }
        d.instances = (const DeviceInstance *)upload(dinst.data(), sizeof(DeviceInstance) * dinst.size());
        d.n_tlas_nodes = n_top_nodes;
        d.n_instances = n_instances;
        {
            // transforms and offsets are only needed to write the shading records
            void *d_fwd = nullptr, *d_moff = nullptr;
            hipError_t err = hipMalloc(&d_fwd, sizeof(float) * fwd.size());
            if (err == hipSuccess) err = hipMalloc(&d_moff, sizeof(int32_t) * moffs.size());
            if (err == hipSuccess)
                err = hipMemcpy(d_fwd, fwd.data(), sizeof(float) * fwd.size(), hipMemcpyHostToDevice);
            if (err == hipSuccess)
                err = hipMemcpy(d_moff, moffs.data(), sizeof(int32_t) * moffs.size(), hipMemcpyHostToDevice);
            if (err == hipSuccess) err = hipDeviceSynchronize();   // uploads went through the null stream
            if (err == hipSuccess) {
                launch_instance_shading_records(ctx->stream, d, (const float4 *)d_fwd, d.mat_ids,
                                                (const int32_t *)d_moff, max_mesh_tris,
                                                (float4 *)d_stris);
                err = hipGetLastError();
            }
            if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
            if (d_fwd) (void)hipFree(d_fwd);
            if (d_moff) (void)hipFree(d_moff);
            if (err != hipSuccess) {
                sc->release();
                return hip_fail(err, "pt_scene_create_instanced: shading records");
            }
        }
        *out = sc.release();
        return PT_OK;
    } catch (const std::bad_alloc &) {
        return fail(PT_ERR_NOMEM, "out of host memory");
    } catch (const std::exception &err) {
        return fail(PT_ERR_INVALID, err.what());
    } catch (...) {
        return fail(PT_ERR_INVALID, "unknown error");
    }
}

uint32_t pt_scene_instance_count(const pt_scene *scene) { return scene ? scene->dev.n_instances : 0; }

uint32_t pt_scene_top_node_count(const pt_scene *scene) { return scene ? scene->dev.n_tlas_nodes : 0; }

int pt_scene_read_top_nodes(pt_scene *scene, pt_bbox *nodes) {
    if (!scene || !nodes) return fail(PT_ERR_INVALID, "pt_scene_read_top_nodes: null argument");
    if (!scene->dev.tlas) return fail(PT_ERR_INVALID, "pt_scene_read_top_nodes: not a two-level scene");
    hipError_t e = hipSetDevice(scene->ctx->device);
    if (e == hipSuccess)
        e = hipMemcpyAsync(nodes, scene->dev.tlas, sizeof(pt_bbox) * (size_t)scene->dev.n_tlas_nodes,
                           hipMemcpyDeviceToHost, scene->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(scene->ctx->stream);
    return e == hipSuccess ? PT_OK : hip_fail(e, "pt_scene_read_top_nodes");
}

uint32_t pt_scene_mesh_node_count(const pt_scene *scene, uint32_t mesh) {
    return scene && mesh < scene->mesh_node_count.size() ? scene->mesh_node_count[mesh] : 0;
}

int pt_scene_read_mesh_nodes(pt_scene *scene, uint32_t mesh, pt_bbox *nodes) {
    if (!scene || !nodes) return fail(PT_ERR_INVALID, "pt_scene_read_mesh_nodes: null argument");
    if (mesh >= scene->mesh_node_count.size())
        return fail(PT_ERR_INVALID, "pt_scene_read_mesh_nodes: no such mesh");
    hipError_t e = hipSetDevice(scene->ctx->device);
    if (e == hipSuccess)
        e = hipMemcpyAsync(nodes, scene->dev.nodes + 2 * (size_t)scene->mesh_node_off[mesh],
                           sizeof(pt_bbox) * (size_t)scene->mesh_node_count[mesh], hipMemcpyDeviceToHost,
                           scene->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(scene->ctx->stream);
    return e == hipSuccess ? PT_OK : hip_fail(e, "pt_scene_read_mesh_nodes");
}

} // extern "C"
