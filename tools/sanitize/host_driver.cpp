// Host code under AddressSanitizer + UndefinedBehaviorSanitizer: the BVH builders, the
// linearizer and its inverse, the traversal re-layout, the light list and the two scene
// readers, driven over generated inputs (no GPU, no HIP runtime calls).
//   make -C tools/sanitize run
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "../../gpu-pathtracer_amd/host/bvh.h"
#include "../../gpu-pathtracer_amd/host/mesh.h"
#include "../../include/mipt.h"

namespace mipt {
// the library's error sink lives in its HIP-facing translation unit; a stand-in for this build
static std::string g_err;
int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
} // namespace mipt

using namespace mipt;

static void check(bool ok, const char *what) {
    if (!ok) {
        std::fprintf(stderr, "FAILED: %s (%s)\n", what, mipt::g_err.c_str());
        std::exit(1);
    }
}

int main() {
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-10.f, 10.f), S(-1.f, 1.f);
    // transforms of two-level scenes
    for (int rep = 0; rep < 200; ++rep) {
        float m[12], inv[12];
        for (float &x : m) x = U(rng);
        if (rep % 50 == 0) for (int k = 0; k < 4; ++k) m[4 + k] = 2.f * m[k];   // singular
        const bool ok = invert_affine(m, inv);
        check(ok == (rep % 50 != 0), "invert_affine");
        pt_bbox root{{-1.f, -2.f, -3.f, 0.f}, {1.f, 2.f, 3.f, 0.f}}, world;
        instance_world_box(m, root, world);
        const float p[3] = {0.5f, -1.f, 2.f};
        float w[3];
        transform_point(m, p, w);
        for (int k = 0; k < 3; ++k) check(w[k] >= world.pmin[k] && w[k] <= world.pmax[k], "instance_world_box");
    }
    for (uint32_t n : {1u, 2u, 3u, 17u, 1000u, 20000u}) {
        std::vector<float> vtx;
        std::vector<int32_t> idx;
        for (uint32_t t = 0; t < n; ++t) {
            const float c[3] = {U(rng), U(rng), U(rng)};
            const float size = (t % 11 == 0) ? 8.f : 0.5f;             // some slivers
            for (int v = 0; v < 3; ++v)
                for (int k = 0; k < 3; ++k) vtx.push_back(c[k] + size * S(rng) * (k == v ? 1.f : 0.1f));
            for (int v = 0; v < 3; ++v) idx.push_back((int32_t)(3 * t + v));
        }
        if (n > 3) {                                                  // duplicates and a degenerate one
            for (int v = 0; v < 3; ++v) idx[3 + v] = idx[v];
            idx[6] = idx[7] = idx[8];
        }
        std::vector<pt_bbox> lb(n);
        mesh_leaf_bounds(vtx.data(), idx.data(), n, lb.data());
        inflate_leaf_bounds(lb.data(), n);
        for (int variant = 0; variant < 4; ++variant) {
            pt_bvh_options opt{};
            opt.max_split_depth = 10; opt.num_bins = variant == 1 ? 2 : 64; opt.min_overlap = 1e-4f;
            opt.traversal_cost = 10.f; opt.extra_node_budget = 0.5f;
            opt.use_sah = variant >= 1; opt.use_splits = variant == 3; opt.num_threads = variant == 2 ? 4 : 1;
            std::string err;
            check(set_option(opt, "bvh.sah.max_split_depth", "16", err), "set_option");
            BvhTree tree;
            build_tree(lb.data(), n, opt, tree);
            if (variant == 1) {
                const double before = tree_area_cost(tree);
                optimize_tree(tree, 2);
                check(tree.size() == 2 * (size_t)n - 1 || opt.use_splits, "optimised tree size");
                check(tree_area_cost(tree) <= before * 1.0000001, "optimised tree area");
            }
            for (uint32_t enc : {PT_LINKS_FLOAT_VALUE, PT_LINKS_INT_BITS}) {
                std::vector<pt_bbox> nodes;
                linearize_skip_links(tree, enc, nodes);
                BvhTree back;
                tree_from_skip_links(nodes.data(), (uint32_t)nodes.size(), enc, n, back);
                check(back.size() == tree.size() && back.n_prims == tree.n_prims, "round trip of the skip-link array");
                WideBvh wide;
                build_wide_layout(back, vtx.data(), (uint32_t)vtx.size() / 3, idx.data(), n, wide);
                check(wide.tris.size() >= tree.n_prims, "triangle records");
                // a damaged array must be rejected, not walked off
                std::vector<pt_bbox> bad = nodes;
                bad[bad.size() / 2].pmax[3] = enc == PT_LINKS_INT_BITS ? 0.f : 1.f;
                try {
                    BvhTree t2;
                    tree_from_skip_links(bad.data(), (uint32_t)bad.size(), enc, n, t2);
                } catch (const std::exception &) {
                }
            }
        }
        std::vector<pt_material> mats(2);
        std::memset(mats.data(), 0, sizeof(pt_material) * 2);
        mats[1].ke[0] = 3.f;
        std::vector<int32_t> mids(n);
        for (uint32_t t = 0; t < n; ++t) mids[t] = t % 5 == 0;
        std::vector<int32_t> lt;
        std::vector<float> lc;
        build_light_list(vtx.data(), idx.data(), n, mats.data(), mids.data(), lt, lc);
        check(lt.size() == lc.size(), "light list");
    }

    // readers: a good OBJ, then truncations of a good glTF and of the OBJ at every length
    const std::string dir = "/tmp/mipt_sanitize_";
    const std::string obj = "mtllib " + dir + "m.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nusemtl a\nf 1/1/1 2//2 3 4\nf -4 -3 -2\n";
    { std::ofstream(dir + "m.mtl") << "newmtl a\nKd 0.5 0.25 0.125\nKe 1 1 1\nNs 100\n"; }
    for (size_t len = 0; len <= obj.size(); ++len) {
        { std::ofstream(dir + "t.obj") << obj.substr(0, len); }
        pt_mesh *m = nullptr;
        if (pt_obj_load((dir + "t.obj").c_str(), &m) == PT_OK) pt_mesh_destroy(m);
    }
    const std::string gltf =
        "{\"asset\":{\"version\":\"2.0\"},\"scenes\":[{\"nodes\":[0]}],\"nodes\":[{\"mesh\":0,\"scale\":[2,2,2],"
        "\"rotation\":[0,0,0,1]}],\"meshes\":[{\"primitives\":[{\"attributes\":{\"POSITION\":0},\"indices\":1,"
        "\"material\":0}]}],\"materials\":[{\"emissiveFactor\":[1,1,1]}],\"buffers\":[{\"byteLength\":42,\"uri\":"
        "\"data:application/octet-stream;base64,AAAAAAAAAAAAAAAAAACAPwAAAAAAAAAAAAAAAAAAgD8AAAAAAAABAAIA\"}],"
        "\"bufferViews\":[{\"buffer\":0,\"byteOffset\":0,\"byteLength\":36},{\"buffer\":0,\"byteOffset\":36,"
        "\"byteLength\":6}],\"accessors\":[{\"bufferView\":0,\"componentType\":5126,\"count\":3,\"type\":\"VEC3\"},"
        "{\"bufferView\":1,\"componentType\":5123,\"count\":3,\"type\":\"SCALAR\"}]}";
    {
        { std::ofstream(dir + "t.gltf") << gltf; }
        pt_mesh *m = nullptr;
        check(pt_gltf_load((dir + "t.gltf").c_str(), 0, &m) == PT_OK, "glTF with a base64 buffer");
        check(m->idx.size() == 3 && m->vtx.size() == 9 && m->vtx[3] == 2.f, "glTF contents");
        pt_mesh_destroy(m);
        pt_model *mo = nullptr;
        check(pt_gltf_load_model((dir + "t.gltf").c_str(), &mo) == PT_OK, "glTF as meshes + instances");
        check(pt_model_mesh_count(mo) == 1 && pt_model_instance_count(mo) >= 1 &&
              pt_mesh_triangle_count(pt_model_mesh(mo, 0)) == 1, "glTF model contents");
        pt_model_destroy(mo);
    }
    for (size_t len = 0; len < gltf.size(); len += 3) {
        { std::ofstream(dir + "t.gltf") << gltf.substr(0, len); }
        pt_mesh *m = nullptr;
        if (pt_gltf_load((dir + "t.gltf").c_str(), 0, &m) == PT_OK) pt_mesh_destroy(m);
        pt_model *mo = nullptr;
        if (pt_gltf_load_model((dir + "t.gltf").c_str(), &mo) == PT_OK) pt_model_destroy(mo);
    }
    std::puts("host code clean under ASan + UBSan");
    return 0;
}
