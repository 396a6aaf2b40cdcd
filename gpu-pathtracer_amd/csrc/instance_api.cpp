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
#include "internal.h"

using namespace mipt;

namespace {

int check_meshes(const pt_mesh_desc *meshes, uint32_t n_meshes, const pt_instance *inst,
                 uint32_t n_inst, const char *who) {
    if (!meshes || !inst) return fail(PT_ERR_INVALID, std::string(who) + ": null argument");
    if (n_meshes == 0 || n_inst == 0)
        return fail(PT_ERR_INVALID, std::string(who) + ": no meshes or no instances");
    for (uint32_t m = 0; m < n_meshes; ++m) {
        const pt_mesh_desc &d = meshes[m];
        if (!d.nodes || !d.vertices || !d.indices || d.n_nodes == 0 || d.n_vertices == 0 ||
            d.n_triangles == 0)
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
    const int rc = check_meshes(meshes, n_meshes, instances, n_instances, "pt_instance_bounds");
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
    if (!ctx || !top_nodes || !out)
        return fail(PT_ERR_INVALID, "pt_scene_create_instanced: null argument");
    *out = nullptr;
    int rc = check_meshes(meshes, n_meshes, instances, n_instances, "pt_scene_create_instanced");
    if (rc != PT_OK) return rc;
    if (materials && n_materials == 0)
        return fail(PT_ERR_INVALID, "pt_scene_create_instanced: materials given but n_materials is 0");
    try {
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess) return hip_fail(e, "hipSetDevice");

        // Top level: a skip-link tree whose leaves name every instance.
        BvhTree top;
        tree_from_skip_links(top_nodes, n_top_nodes, link_encoding, n_instances, top);
        {
            std::vector<uint8_t> seen(n_instances, 0);
            for (size_t i = 0; i < top.size(); ++i)
                if (top.prim[i] >= 0) seen[(size_t)top.prim[i]] = 1;
            for (uint32_t k = 0; k < n_instances; ++k)
                if (!seen[k])
                    return fail(PT_ERR_INVALID, "pt_scene_create_instanced: instance " +
                                                    std::to_string(k) + " is in no top-level leaf");
        }

        std::vector<pt_material> mats;
        if (materials) {
            mats.assign(materials, materials + n_materials);
        } else {
            pt_material m{};
            m.kd[0] = m.kd[1] = m.kd[2] = 0.5f;
            m.roughness = 1.f;
            mats.push_back(m);
        }

        // Meshes, back to back, each validated and re-laid-out like a scene of its own.
        struct MeshOffsets { uint32_t node, wnode, wtri, vtx, tri; };
        std::vector<MeshOffsets> off(n_meshes);
        std::vector<pt_bbox> nodes_all;
        std::vector<float> vtx_all;
        std::vector<int32_t> idx_all;
        std::vector<WideNode> wnodes_all;
        std::vector<TriRecord> wtris_all;
        uint32_t wide_depth = 0;
        for (uint32_t m = 0; m < n_meshes; ++m) {
            const pt_mesh_desc &d = meshes[m];
            const std::string tag = "pt_scene_create_instanced: mesh " + std::to_string(m) + ": ";
            BvhTree tree;
            try {
                tree_from_skip_links(d.nodes, d.n_nodes, link_encoding, d.n_triangles, tree);
            } catch (const std::runtime_error &err) {
                return fail(PT_ERR_INVALID, tag + err.what());
            }
            std::vector<uint8_t> seen(d.n_triangles, 0);
            for (size_t i = 0; i < tree.size(); ++i)
                if (tree.prim[i] >= 0) seen[(size_t)tree.prim[i]] = 1;
            for (uint32_t k = 0; k < d.n_triangles; ++k)
                if (!seen[k]) return fail(PT_ERR_INVALID, tag + "triangle " + std::to_string(k) + " is in no leaf");
            for (uint64_t i = 0; i < 3ull * d.n_triangles; ++i)
                if (d.indices[i] < 0 || (uint32_t)d.indices[i] >= d.n_vertices)
                    return fail(PT_ERR_INVALID, tag + "index " + std::to_string(d.indices[i]) +
                                                    " outside the vertex buffer");
            WideBvh wide;
            build_wide_layout(tree, d.vertices, d.n_vertices, d.indices, d.n_triangles, wide);
            if (wnodes_all.size() + wide.nodes.size() > (1u << 26) ||
                wtris_all.size() + wide.tris.size() > 89000000u)
                return fail(PT_ERR_UNSUPPORTED,
                            "pt_scene_create_instanced: meshes too large together for 32-bit record offsets");
            // child indices become scene-wide: an inner child is the index of a two-child node,
            // a leaf child the complement of its record's position in 16-byte units
            {
                const int32_t node_base = (int32_t)wnodes_all.size();
                const int32_t tri_base = 3 * (int32_t)wtris_all.size();
                for (WideNode &w : wide.nodes)
                    for (int c = 0; c < 2; ++c)
                        w.child[c] = w.child[c] >= 0 ? w.child[c] + node_base : ~(~w.child[c] + tri_base);
            }
            if (nodes_all.size() + d.n_nodes > 0x7fffffffull ||
                vtx_all.size() / 3 + d.n_vertices > 0x7fffffffull ||
                wtris_all.size() + wide.tris.size() > 0x7fffffffull)
                return fail(PT_ERR_UNSUPPORTED, "pt_scene_create_instanced: meshes too large together");
            off[m] = {(uint32_t)nodes_all.size(), (uint32_t)wnodes_all.size(),
                      (uint32_t)wtris_all.size(), (uint32_t)(vtx_all.size() / 3),
                      (uint32_t)(idx_all.size() / 3)};
            nodes_all.insert(nodes_all.end(), d.nodes, d.nodes + d.n_nodes);
            vtx_all.insert(vtx_all.end(), d.vertices, d.vertices + 3 * (size_t)d.n_vertices);
            idx_all.insert(idx_all.end(), d.indices, d.indices + 3 * (size_t)d.n_triangles);
            wnodes_all.insert(wnodes_all.end(), wide.nodes.begin(), wide.nodes.end());
            wtris_all.insert(wtris_all.end(), wide.tris.begin(), wide.tris.end());
            if (wide.max_depth > wide_depth) wide_depth = wide.max_depth;
        }

        // Instances.  Their world-space shading records are written on the GPU below; the host
        // only transforms the emissive triangles, for the light list.
        uint64_t total_tris = 0;
        uint32_t max_mesh_tris = 0;
        for (uint32_t i = 0; i < n_instances; ++i) total_tris += meshes[instances[i].mesh].n_triangles;
        if (total_tris > 0x7fffffffull)
            return fail(PT_ERR_UNSUPPORTED, "pt_scene_create_instanced: more than 2^31 instanced triangles");
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
                return fail(PT_ERR_INVALID, "pt_scene_create_instanced: the transform of instance " +
                                                std::to_string(i) + " is not invertible");
            di.node_off = off[in.mesh].node;
            di.wnode_off = off[in.mesh].wnode;
            di.wtri_off = off[in.mesh].wtri;
            di.vtx_off = off[in.mesh].vtx;
            di.tri_off = off[in.mesh].tri;
            di.n_tri = d.n_triangles;
            di.first_tri = first;
            di._pad = 0;
            moffs[i] = in.material_offset;
            std::memcpy(&fwd[12 * (size_t)i], in.object_to_world, sizeof(float) * 12);
            bool emissive = false;
            for (int32_t id : used_ids[in.mesh]) {
                const int64_t mid = (int64_t)id + in.material_offset;
                if (mid < 0 || mid >= (int64_t)mats.size())
                    return fail(PT_ERR_INVALID, "pt_scene_create_instanced: instance " +
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

        auto sc = std::make_unique<pt_scene>();
        sc->ctx = ctx;
        auto upload = [&](const void *src, size_t bytes) -> void * {
            void *dst = nullptr;
            hipError_t err = hipMalloc(&dst, bytes);
            if (err == hipSuccess) {
                sc->allocs.push_back(dst);
                err = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
            }
            if (err != hipSuccess) {
                sc->release();
                throw std::runtime_error(std::string("pt_scene_create_instanced upload: ") +
                                         hipGetErrorString(err));
            }
            return dst;
        };
        DeviceScene &d = sc->dev;
        d.nodes = (const float4 *)upload(nodes_all.data(), sizeof(pt_bbox) * nodes_all.size());
        d.vtx = (const float *)upload(vtx_all.data(), sizeof(float) * vtx_all.size());
        d.idx = (const int32_t *)upload(idx_all.data(), sizeof(int32_t) * idx_all.size());
        d.wnodes = (const float4 *)upload(wnodes_all.data(), sizeof(WideNode) * wnodes_all.size());
        d.wtris = (const float4 *)upload(wtris_all.data(), sizeof(TriRecord) * wtris_all.size());
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
        d.n_nodes = (uint32_t)nodes_all.size();
        d.n_tri = (uint32_t)total_tris;
        d.n_wnodes = (uint32_t)wnodes_all.size();
        d.n_materials = (uint32_t)mats.size();
        d.link_enc = link_encoding;
        d.wide_depth = wide_depth;
        d.tlas = (const float4 *)upload(top_nodes, sizeof(pt_bbox) * (size_t)n_top_nodes);
        {
            std::vector<WideNode> wtop;
            build_wide_top(top, wtop);
            d.wtop = (const float4 *)upload(wtop.data(), sizeof(WideNode) * wtop.size());
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

} // extern "C"
