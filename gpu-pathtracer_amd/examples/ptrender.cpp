// ptrender: a host program of the shape of the reference's application, written against the
// C ABI only (include/mipt.h): load an OBJ (src/main.cpp:852-899) or glTF
// (src/main_granite.cpp:265-268) scene, build the BVH and upload
// the scene (src/main_granite.cpp:115-194), render, write an image.
//
//   ptrender scene.obj|scene.gltf|scene.glb out.ppm|.png|.pfm [--size W H] [--spp N] [--depth D] [--nee] [--gpu-build]
//            [--eye x y z] [--target x y z] [--fov deg] [--seed S] [--film out.raw]
//
// --film writes the raw film (W*H float4: rgb sums and sample count) for exact comparisons.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
                             "[--seed S] [--film out.raw]\n");
        return 2;
    }
    uint32_t W = 512, H = 512, spp = 16, depth = 4, seed = 1;
    bool nee = false, gpu_build = false, instanced = false;
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
        else { std::fprintf(stderr, "ptrender: unknown option %s\n", a.c_str()); return 2; }
    }

    const std::string scene_path = argv[1];
    auto ends_with = [&](const char *suffix) {
        const size_t n = std::strlen(suffix);
        return scene_path.size() >= n && scene_path.compare(scene_path.size() - n, n, suffix) == 0;
    };
    const bool gltf = ends_with(".glb") || ends_with(".gltf");
    pt_context *ctx = nullptr;
    CHECK(pt_context_create(0, &ctx));
    const std::string passes = std::to_string(optimize);
    CHECK(pt_context_set_option(ctx, "build.optimize_passes", passes.c_str()));   // trees built on the GPU
    pt_scene *scene = nullptr;
    pt_mesh *mesh = nullptr;
    if (instanced) {
        // the glTF scene as structured: each mesh once, its nodes as instances; every tree is
        // left to the library (meshes on the GPU, top level on the host)
        if (!gltf) { std::fprintf(stderr, "ptrender: --instanced needs a glTF scene\n"); return 2; }
        pt_model *model = nullptr;
        CHECK(pt_gltf_load_model(argv[1], &model));
        std::vector<pt_mesh_desc> descs(pt_model_mesh_count(model));
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
        CHECK(pt_scene_create_instanced(ctx, descs.data(), (uint32_t)descs.size(), pt_model_instances(model),
                                        pt_model_instance_count(model), nullptr, 0, PT_LINKS_INT_BITS,
                                        pt_model_materials(model), pt_model_material_count(model), &scene));
        pt_model_destroy(model);
    } else {
        if (gltf) CHECK(pt_gltf_load(argv[1], 0, &mesh));   // src/main_granite.cpp:268
        else CHECK(pt_obj_load(argv[1], &mesh));
        const uint32_t n_vtx = pt_mesh_vertex_count(mesh), n_tri = pt_mesh_triangle_count(mesh);
        std::printf("%s: %u vertices, %u triangles, %u materials\n", argv[1], n_vtx, n_tri,
                    pt_mesh_material_count(mesh));
        if (gpu_build) {
            CHECK(pt_scene_build(ctx, pt_mesh_vertices(mesh), n_vtx, pt_mesh_indices(mesh), n_tri,
                                 pt_mesh_materials(mesh), pt_mesh_material_count(mesh),
                                 pt_mesh_material_ids(mesh), PT_LINKS_INT_BITS, &scene));
        } else {
            std::vector<pt_bbox> leaves(n_tri);
            CHECK(pt_mesh_leaf_bounds(pt_mesh_vertices(mesh), n_vtx, pt_mesh_indices(mesh), n_tri, leaves.data()));
            pt_bvh_options opt;
            pt_bvh_options_default(&opt);
            CHECK(pt_bvh_options_set(&opt, "bvh.builder", "sah"));       // src/main_granite.cpp:176
            CHECK(pt_bvh_options_set(&opt, "mipt.link_encoding", "int"));
            CHECK(pt_bvh_options_set(&opt, "mipt.optimize_passes", passes.c_str()));
            pt_bvh *bvh = nullptr;
            CHECK(pt_bvh_build(leaves.data(), n_tri, &opt, &bvh));
            CHECK(pt_scene_create(ctx, pt_bvh_nodes(bvh), pt_bvh_node_count(bvh), pt_bvh_link_encoding(bvh),
                                  pt_mesh_vertices(mesh), n_vtx, pt_mesh_indices(mesh), n_tri,
                                  pt_mesh_materials(mesh), pt_mesh_material_count(mesh),
                                  pt_mesh_material_ids(mesh), &scene));
            pt_bvh_destroy(bvh);
        }
    }

    // look-at camera: columns of mat3(inv_view) are right, up, back
    pt_camera cam;
    std::memset(&cam, 0, sizeof(cam));
    double back[3] = {eye[0] - target[0], eye[1] - target[1], eye[2] - target[2]}, right[3], up[3];
    const double world_up[3] = {0, 1, 0};
    normalize(back);
    cross(world_up, back, right);
    normalize(right);
    cross(back, right, up);
    const double t = std::tan(fov * 3.14159265358979323846 / 180.0 * 0.5);
    for (int k = 0; k < 3; ++k) {
        cam.eye[k] = (float)eye[k];
        cam.c0[k] = (float)right[k];
        cam.c1[k] = (float)up[k];
        cam.c2[k] = (float)back[k];
    }
    cam.sx = (float)(t * (double)W / (double)H);
    cam.sy = (float)t;
    cam.mode = PT_CAMERA_PINHOLE;

    pt_film *film = nullptr;
    CHECK(pt_film_create(ctx, W, H, &film));
    pt_render_params rp;
    pt_render_params_default(&rp, W, H);
    rp.sample_count = spp;
    rp.max_depth = depth;
    rp.seed = seed;
    rp.flags = nee ? PT_RENDER_NEE : PT_RENDER_DEFAULT;
    pt_render_stats st;
    CHECK(pt_render(ctx, scene, &cam, &rp, film, &st));
    std::printf("%llu paths, %llu primary + %llu bounce + %llu shadow rays\n",
                (unsigned long long)st.paths, (unsigned long long)st.rays_primary,
                (unsigned long long)st.rays_bounce, (unsigned long long)st.rays_shadow);
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
    pt_film_destroy(film);
    pt_scene_destroy(scene);
    pt_context_destroy(ctx);
    pt_mesh_destroy(mesh);
    return 0;
}
