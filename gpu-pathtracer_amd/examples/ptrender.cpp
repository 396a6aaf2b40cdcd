// ptrender: a host program of the shape of the reference's application, written against the
// C ABI only (include/mipt.h): load an OBJ (src/main.cpp:852-899) or glTF
// (src/main_granite.cpp:265-268) scene, build the BVH and upload
// the scene (src/main_granite.cpp:115-194), render, write an image.
//
//   ptrender scene.obj|scene.gltf|scene.glb out.ppm|.png|.pfm [--size W H] [--spp N] [--depth D] [--nee] [--gpu-build]
//            [--eye x y z] [--target x y z] [--fov deg] [--seed S] [--film out.raw]
//            [--gpus N] [--sharding tiles|samples] [--comm]
//
// --film writes the raw film (W*H float4: rgb sums and sample count) for exact comparisons.
// --gpus N renders one frame on N GPUs of this machine: one context, one host thread and one
// stream per GPU, the scene replicated, the frame split by 64x64 tiles (bit-identical to one
// GPU) or by sample ranges, and assembled on GPU 0 over RCCL (pt_comm_create_all,
// pt_film_gather_tiles / pt_film_reduce).  --comm makes the communicator even for one GPU.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mipt.h"

static void die(const char *what) {
    std::fprintf(stderr, "ptrender: %s: %s\n", what, pt_last_error());
    std::exit(1);
}
#define CHECK(call) do { if ((call) != PT_OK) die(#call); } while (0)

static void normalize(double v[3]) {
    const double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    for (int k = 0; k < 3; ++k) v[k] /= l;
}
static void cross(const double a[3], const double b[3], double out[3]) {
    out[0] = a[1] * b[2] - a[2] * b[1];
    out[1] = a[2] * b[0] - a[0] * b[2];
    out[2] = a[0] * b[1] - a[1] * b[0];
}

int main(int argc, char **argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: ptrender scene.obj out.ppm|.png|.pfm [--size W H] [--spp N] [--depth D] "
                             "[--nee] [--gpu-build] [--optimize PASSES] [--instanced] [--eye x y z] [--target x y z] [--fov deg] "
                             "[--seed S] [--film out.raw] [--gpus N] [--sharding tiles|samples] [--comm]\n");
        return 2;
    }
    uint32_t W = 512, H = 512, spp = 16, depth = 4, seed = 1;
    bool nee = false, gpu_build = false, instanced = false, force_comm = false, by_tiles = true;
    int n_gpus = 1;
    int optimize = 0;         // insertion-based optimisation passes over the tree (host or GPU build)
    double eye[3] = {278, 273, -800}, target[3] = {278, 273, 0}, fov = 39.3;
    const char *film_path = nullptr;
    for (int i = 3; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](int k) { if (i + k >= argc) { std::fprintf(stderr, "ptrender: %s needs %d value(s)\n", a.c_str(), k); std::exit(2); } };
        if (a == "--size") { need(2); W = std::atoi(argv[++i]); H = std::atoi(argv[++i]); }
        else if (a == "--spp") { need(1); spp = std::atoi(argv[++i]); }
        else if (a == "--depth") { need(1); depth = std::atoi(argv[++i]); }
        else if (a == "--seed") { need(1); seed = std::atoi(argv[++i]); }
        else if (a == "--fov") { need(1); fov = std::atof(argv[++i]); }
        else if (a == "--eye") { need(3); for (int k = 0; k < 3; ++k) eye[k] = std::atof(argv[++i]); }
        else if (a == "--target") { need(3); for (int k = 0; k < 3; ++k) target[k] = std::atof(argv[++i]); }
        else if (a == "--film") { need(1); film_path = argv[++i]; }
        else if (a == "--nee") nee = true;
        else if (a == "--gpu-build") gpu_build = true;
        else if (a == "--optimize") { need(1); optimize = std::atoi(argv[++i]); }
        else if (a == "--instanced") instanced = true;
        else if (a == "--gpus") { need(1); n_gpus = std::atoi(argv[++i]); }
        else if (a == "--comm") force_comm = true;
        else if (a == "--sharding") {
            need(1);
            const std::string v = argv[++i];
            if (v != "tiles" && v != "samples") { std::fprintf(stderr, "ptrender: --sharding tiles|samples\n"); return 2; }
            by_tiles = v == "tiles";
        }
        else { std::fprintf(stderr, "ptrender: unknown option %s\n", a.c_str()); return 2; }
    }

    const std::string scene_path = argv[1];
    auto ends_with = [&](const char *suffix) {
        const size_t n = std::strlen(suffix);
        return scene_path.size() >= n && scene_path.compare(scene_path.size() - n, n, suffix) == 0;
    };
    const bool gltf = ends_with(".glb") || ends_with(".gltf");
    if (n_gpus < 1 || n_gpus > pt_device_count()) {
        std::fprintf(stderr, "ptrender: --gpus %d, but %d GPU(s) are visible\n", n_gpus, pt_device_count());
        return 2;
    }
    if (instanced && !gltf) { std::fprintf(stderr, "ptrender: --instanced needs a glTF scene\n"); return 2; }
    const std::string passes = std::to_string(optimize);

    // ---- host side, once: the scene file, and the tree if it is built on the host ------------
    pt_mesh *mesh = nullptr;
    pt_model *model = nullptr;
    pt_bvh *bvh = nullptr;
    std::vector<pt_mesh_desc> descs;
    if (instanced) {
        // the glTF scene as structured: each mesh once, its nodes as instances; every tree is
        // left to the library (meshes on the GPU, top level on the host)
        CHECK(pt_gltf_load_model(argv[1], &model));
        descs.resize(pt_model_mesh_count(model));
        for (uint32_t k = 0; k < descs.size(); ++k) {
            const pt_mesh *m = pt_model_mesh(model, k);
            std::memset(&descs[k], 0, sizeof(pt_mesh_desc));
            descs[k].vertices = pt_mesh_vertices(m);
            descs[k].indices = pt_mesh_indices(m);
            descs[k].material_ids = pt_mesh_material_ids(m);
            descs[k].n_vertices = pt_mesh_vertex_count(m);
            descs[k].n_triangles = pt_mesh_triangle_count(m);
        }
        std::printf("%s: %u meshes, %u instances, %u materials\n", argv[1], (unsigned)descs.size(),
                    pt_model_instance_count(model), pt_model_material_count(model));
    } else {
        if (gltf) CHECK(pt_gltf_load(argv[1], 0, &mesh));   // src/main_granite.cpp:268
        else CHECK(pt_obj_load(argv[1], &mesh));
        const uint32_t n_vtx = pt_mesh_vertex_count(mesh), n_tri = pt_mesh_triangle_count(mesh);
        std::printf("%s: %u vertices, %u triangles, %u materials\n", argv[1], n_vtx, n_tri,
                    pt_mesh_material_count(mesh));
        if (!gpu_build) {
            std::vector<pt_bbox> leaves(n_tri);
            CHECK(pt_mesh_leaf_bounds(pt_mesh_vertices(mesh), n_vtx, pt_mesh_indices(mesh), n_tri, leaves.data()));
            pt_bvh_options opt;
            pt_bvh_options_default(&opt);
            CHECK(pt_bvh_options_set(&opt, "bvh.builder", "sah"));       // src/main_granite.cpp:176
            CHECK(pt_bvh_options_set(&opt, "mipt.link_encoding", "int"));
            CHECK(pt_bvh_options_set(&opt, "mipt.optimize_passes", passes.c_str()));
            CHECK(pt_bvh_build(leaves.data(), n_tri, &opt, &bvh));
        }
    }

    // the live application's camera set-up, src/main_granite.cpp:306-308
    pt_camera cam;
    const float eye_f[3] = {(float)eye[0], (float)eye[1], (float)eye[2]};
    const float target_f[3] = {(float)target[0], (float)target[1], (float)target[2]};
    CHECK(pt_camera_look_at(eye_f, target_f, nullptr, (float)(fov * 3.14159265358979323846 / 180.0),
                            (float)W / (float)H, PT_CAMERA_PINHOLE, &cam));

    // ---- one context per GPU, and their communicator --------------------------------------------
    struct Gpu {
        pt_context *ctx = nullptr;
        pt_comm *comm = nullptr;
        pt_scene *scene = nullptr;
        pt_film *film = nullptr;
        pt_render_stats st{};
    };
    std::vector<Gpu> gpus((size_t)n_gpus);
    for (int g = 0; g < n_gpus; ++g) {
        CHECK(pt_context_create(g, &gpus[(size_t)g].ctx));
        CHECK(pt_context_set_option(gpus[(size_t)g].ctx, "build.optimize_passes", passes.c_str()));   // trees built on the GPU
    }
    if (n_gpus > 1 || force_comm) {
        std::vector<pt_context *> ctxs;
        std::vector<pt_comm *> comms((size_t)n_gpus, nullptr);
        for (Gpu &g : gpus) ctxs.push_back(g.ctx);
        CHECK(pt_comm_create_all(ctxs.data(), n_gpus, comms.data()));
        for (int g = 0; g < n_gpus; ++g) gpus[(size_t)g].comm = comms[(size_t)g];
    }

    // ---- per GPU, each on its own host thread: upload, render this GPU's share, assemble ----------
    auto drive = [&](int rank) {
        Gpu &g = gpus[(size_t)rank];
        if (instanced) {
            CHECK(pt_scene_create_instanced(g.ctx, descs.data(), (uint32_t)descs.size(), pt_model_instances(model),
                                            pt_model_instance_count(model), nullptr, 0, PT_LINKS_INT_BITS,
                                            pt_model_materials(model), pt_model_material_count(model), &g.scene));
        } else if (gpu_build) {
            CHECK(pt_scene_build(g.ctx, pt_mesh_vertices(mesh), pt_mesh_vertex_count(mesh), pt_mesh_indices(mesh),
                                 pt_mesh_triangle_count(mesh), pt_mesh_materials(mesh), pt_mesh_material_count(mesh),
                                 pt_mesh_material_ids(mesh), PT_LINKS_INT_BITS, &g.scene));
        } else {
            CHECK(pt_scene_create(g.ctx, pt_bvh_nodes(bvh), pt_bvh_node_count(bvh), pt_bvh_link_encoding(bvh),
                                  pt_mesh_vertices(mesh), pt_mesh_vertex_count(mesh), pt_mesh_indices(mesh),
                                  pt_mesh_triangle_count(mesh), pt_mesh_materials(mesh),
                                  pt_mesh_material_count(mesh), pt_mesh_material_ids(mesh), &g.scene));
        }
        CHECK(pt_film_create(g.ctx, W, H, &g.film));
        pt_render_params rp;
        pt_render_params_default(&rp, W, H);
        rp.sample_count = spp;
        rp.max_depth = depth;
        rp.seed = seed;
        rp.flags = nee ? PT_RENDER_NEE : PT_RENDER_DEFAULT;
        if (by_tiles) {                      // tiles t with t % n_gpus == rank, every sample of them
            rp.tile_size = 64;
            rp.shard_index = (uint32_t)rank;
            rp.shard_count = (uint32_t)n_gpus;
        } else {                             // a contiguous range of the samples of every pixel
            const uint32_t base = spp / (uint32_t)n_gpus, extra = spp % (uint32_t)n_gpus;
            rp.sample_begin = (uint32_t)rank * base + ((uint32_t)rank < extra ? (uint32_t)rank : extra);
            rp.sample_count = base + ((uint32_t)rank < extra ? 1u : 0u);
        }
        if (rp.sample_count) CHECK(pt_render(g.ctx, g.scene, &cam, &rp, g.film, &g.st));
        if (g.comm) {
            if (by_tiles) CHECK(pt_film_gather_tiles(g.film, g.comm, 64, 0));
            else CHECK(pt_film_reduce(g.film, g.comm, 0));
            CHECK(pt_context_synchronize(g.ctx));
        }
    };
    if (n_gpus == 1) {
        drive(0);
    } else {
        std::vector<std::thread> threads;
        for (int g = 0; g < n_gpus; ++g) threads.emplace_back(drive, g);
        for (std::thread &th : threads) th.join();
    }
    pt_render_stats st{};
    for (const Gpu &g : gpus) {
        st.paths += g.st.paths;
        st.rays_primary += g.st.rays_primary;
        st.rays_bounce += g.st.rays_bounce;
        st.rays_shadow += g.st.rays_shadow;
    }
    std::printf("%llu paths, %llu primary + %llu bounce + %llu shadow rays\n",
                (unsigned long long)st.paths, (unsigned long long)st.rays_primary,
                (unsigned long long)st.rays_bounce, (unsigned long long)st.rays_shadow);
    pt_film *film = gpus[0].film;            // the assembled frame
    // container by file extension: .png, .pfm, anything else is PPM
    const std::string out_path = argv[2];
    auto out_ends_with = [&](const char *ext) {
        const size_t n = std::strlen(ext);
        return out_path.size() >= n && out_path.compare(out_path.size() - n, n, ext) == 0;
    };
    const uint32_t fmt = out_ends_with(".png") ? PT_IMAGE_PNG : out_ends_with(".pfm") ? PT_IMAGE_PFM : PT_IMAGE_PPM;
    CHECK(pt_film_write_image(film, argv[2], fmt, 1.0f));
    if (film_path) {
        std::vector<float> px(4ull * W * H);
        CHECK(pt_film_read(film, px.data()));
        FILE *f = std::fopen(film_path, "wb");
        if (!f || std::fwrite(px.data(), sizeof(float), px.size(), f) != px.size()) {
            std::fprintf(stderr, "ptrender: cannot write %s\n", film_path);
            return 1;
        }
        std::fclose(f);
    }
    for (Gpu &g : gpus) {
        pt_comm_destroy(g.comm);
        pt_film_destroy(g.film);
        pt_scene_destroy(g.scene);
        pt_context_destroy(g.ctx);
    }
    pt_bvh_destroy(bvh);
    pt_model_destroy(model);
    pt_mesh_destroy(mesh);
    return 0;
}
