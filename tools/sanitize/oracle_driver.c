/* Runs the CPU oracle (test infrastructure) under ASan + UBSan on generated scenes: leaf
 * boxes, both tree builders, the linearizer, every batch query, raygen and the path tracer
 * with and without light sampling.  Checks only a few cross-agreements; the point is that the
 * sanitizers see every function on realistic input. */
#include "../../oracle/pt_oracle.h"
#include "../../oracle/pt_shade_oracle.c"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint32_t rng_state = 12345u;
static float frand(void)
{
    rng_state = rng_state * 1664525u + 1013904223u;
    return (float)(rng_state >> 8) * (1.0f / 16777216.0f);
}

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #c); exit(1); } } while (0)

static void run_scene(uint32_t n_tri, int use_sah, int nthreads)
{
    const uint32_t n_vtx = 3 * n_tri;
    float *vtx = malloc(sizeof(float) * 3 * n_vtx);
    int32_t *idx = malloc(sizeof(int32_t) * 3 * n_tri);
    int32_t *mat_ids = malloc(sizeof(int32_t) * n_tri);
    for (uint32_t t = 0; t < n_tri; ++t) {
        const float cx = frand() * 2 - 1, cy = frand() * 2 - 1, cz = frand() * 2 - 1;
        for (int k = 0; k < 3; ++k) {
            vtx[9 * t + 3 * k + 0] = cx + (frand() - 0.5f) * 0.4f;
            vtx[9 * t + 3 * k + 1] = cy + (frand() - 0.5f) * 0.4f;
            vtx[9 * t + 3 * k + 2] = cz + (frand() - 0.5f) * 0.4f;
            idx[3 * t + k] = (int32_t)(3 * t + k);
        }
        mat_ids[t] = (int32_t)(t % 3);
    }
    os_material mats[3];
    memset(mats, 0, sizeof mats);
    for (int k = 0; k < 3; ++k) { mats[0].kd[k] = 0.7f; mats[1].kd[k] = 0.2f; mats[1].ks[k] = 0.5f; mats[2].ke[k] = 4.0f; }
    mats[1].roughness = 0.3f;

    o_bbox *leaf = malloc(sizeof(o_bbox) * n_tri);
    o_mesh_leaf_bounds(vtx, idx, n_tri, leaf);
    const uint32_t n_nodes = 2 * n_tri - 1;
    o_bbox *nodes = malloc(sizeof(o_bbox) * n_nodes);
    CHECK(o_build_bvh(leaf, n_tri, use_sah, 64, 10.0f, nodes) == 0);

    /* the same tree through the explicit steps */
    o_bbox *leaf2 = malloc(sizeof(o_bbox) * n_tri);
    memcpy(leaf2, leaf, sizeof(o_bbox) * n_tri);
    o_inflate_leaf_bounds(leaf2, n_tri);
    o_tree tree;
    CHECK(o_tree_build(leaf2, n_tri, use_sah, 64, 10.0f, &tree) == 0);
    o_bbox *nodes2 = malloc(sizeof(o_bbox) * n_nodes);
    CHECK(o_linearize(&tree, nodes2) == (int)n_nodes || o_linearize(&tree, nodes2) == 0);
    CHECK(memcmp(nodes, nodes2, sizeof(o_bbox) * n_nodes) == 0);
    o_tree_free(&tree);

    const uint64_t n_rays = 4096;
    o_ray *rays = malloc(sizeof(o_ray) * n_rays);
    for (uint64_t i = 0; i < n_rays; ++i) {
        for (int k = 0; k < 3; ++k) { rays[i].o[k] = frand() * 4 - 2; rays[i].d[k] = frand() * 2 - 1; }
        rays[i].o[3] = i % 7 == 0 ? 0.5f : 1e30f;
        rays[i].d[3] = 0.0f;
    }
    /* degenerate inputs the kernels are also given */
    memset(&rays[0], 0, sizeof(o_ray));
    rays[1].d[0] = 0.0f; rays[1].d[1] = 0.0f;
    rays[2].o[3] = 0.0f;

    o_isect *h0 = malloc(sizeof(o_isect) * n_rays), *h1 = malloc(sizeof(o_isect) * n_rays);
    int32_t *any = malloc(sizeof(int32_t) * n_rays);
    o_counters cnt = { 0, 0 };
    o_closest_batch(nodes, vtx, idx, rays, n_rays, h0, &cnt, nthreads);
    o_brute_closest_batch(vtx, idx, n_tri, rays, n_rays, h1, nthreads);
    o_any_batch(nodes, vtx, idx, rays, n_rays, any, nthreads);
    uint64_t differ = 0, edge = 0;
    for (uint64_t i = 0; i < n_rays; ++i) {
        if (i < 3) continue; /* degenerate rays: NaN distances compare unequal */
        /* the occlusion query tests triangles with an epsilon-tolerant test, the closest-hit
         * query with the watertight one (shaders/RadeonRays/bvh.glslh:166-203 against :206-287):
         * they may disagree at triangle edges */
        if ((h0[i].shapeid >= 0) != (any[i] != 0)) ++edge;
        /* the scan and the tree may pick different triangles only at an exact distance tie */
        if (h0[i].primid != h1[i].primid) { ++differ; CHECK(h0[i].uvwt[3] == h1[i].uvwt[3]); }
    }
    CHECK(differ < n_rays / 100 && edge < n_rays / 100);
    o_closest_mt_batch(nodes, vtx, idx, rays, n_rays, h1, nthreads);
    for (uint32_t i = 0; i < n_nodes; ++i) { /* the same tree with its indices stored as integer bit patterns */
        const int32_t a = (int32_t)nodes[i].pmin[3], b = (int32_t)nodes[i].pmax[3];
        nodes2[i] = nodes[i];
        memcpy(&nodes2[i].pmin[3], &a, 4);
        memcpy(&nodes2[i].pmax[3], &b, 4);
    }
    o_closest_intbits_batch(nodes2, vtx, idx, rays, n_rays, h1, nthreads);
    for (uint64_t i = 3; i < n_rays; ++i) CHECK(h0[i].primid == h1[i].primid && h0[i].uvwt[3] == h1[i].uvwt[3]);
    float *rgba = malloc(sizeof(float) * 4 * n_rays);
    o_depth_shade_batch(h0, n_rays, rgba);

    const uint32_t W = 24, H = 16;
    o_camera rc = { 1.0f, 1.0f, { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 }, { 0, 0, 3 } };
    o_ray *img = malloc(sizeof(o_ray) * W * H);
    o_raygen_reference_image(&rc, W, H, img);

    os_camera cam;
    memset(&cam, 0, sizeof cam);
    cam.eye[2] = 3.5f; cam.sx = 0.6f; cam.sy = 0.4f;
    cam.c0[0] = 1; cam.c1[1] = 1; cam.c2[2] = 1; cam.mode = 1;
    os_params P;
    memset(&P, 0, sizeof P);
    P.width = W; P.height = H; P.sample_count = 4; P.max_depth = 6; P.seed = 7; P.rr_start_depth = 2;
    P.tile_size = 8; P.shard_count = 1; P.samples_per_pass = 4; P.ray_offset = 1e-4f;
    P.background[0] = 0.1f;
    float *film = calloc(4u * W * H, sizeof(float));
    for (uint32_t flags = 0; flags < 8; ++flags) {
        P.flags = flags;
        cam.mode = (flags & OS_FLAG_DEPTH_GREY) ? 0u : 1u;
        const uint64_t r = os_render(nodes, vtx, idx, mats, mat_ids, &cam, &P, 0, 0, W, H, film, nthreads, n_tri);
        CHECK(r >= (uint64_t)W * H);
    }
    for (uint32_t i = 0; i < 4u * W * H; ++i) CHECK(film[i] == film[i]);

    /* the same mesh as a two-level scene: placed three times (one mirrored, two coincident) */
    if (n_tri >= 2) {
        const float tr[3][12] = {
            { 1, 0, 0, 0.5f,  0, 1, 0, 0,  0, 0, 1, 0 },
            { -0.7f, 0.2f, 0, -2.0f,  0.1f, 1.3f, 0.2f, 1.0f,  0, -0.3f, 0.9f, 0.5f },
            { 1, 0, 0, 0.5f,  0, 1, 0, 0,  0, 0, 1, 0 },
        };
        o_mesh mesh = { nodes, vtx, idx };
        o_instance inst[3];
        o_bbox ib[3], top[5];
        memset(inst, 0, sizeof inst);
        for (int k = 0; k < 3; ++k) {
            CHECK(o_invert_affine(tr[k], inst[k].inv));
            o_instance_bounds(tr[k], &nodes[0], &ib[k]);
        }
        CHECK(o_build_bvh(ib, 3, 1, 64, 10.0f, top) == 0);
        o_two_level_closest_batch(top, &mesh, inst, rays, n_rays, h0, &cnt, nthreads);
        o_two_level_any_batch(top, &mesh, inst, rays, n_rays, any, nthreads);
        float *wv = malloc(sizeof(float) * 9 * 3 * n_tri);
        int32_t *wi = malloc(sizeof(int32_t) * 9 * n_tri), *wm = malloc(sizeof(int32_t) * 3 * n_tri);
        uint32_t first[3];
        for (int k = 0; k < 3; ++k) {
            first[k] = (uint32_t)k * n_tri;
            for (uint32_t t = 0; t < n_tri; ++t) {
                for (int j = 0; j < 3; ++j)
                    o_transform_points(tr[k], vtx + 3 * (size_t)idx[3 * t + j], 1, wv + 9 * ((size_t)first[k] + t) + 3 * j);
                wm[first[k] + t] = mat_ids[t];
            }
        }
        for (uint32_t i = 0; i < 9 * n_tri; ++i) wi[i] = (int32_t)i;
        memset(film, 0, sizeof(float) * 4u * W * H);
        for (uint32_t flags = 0; flags < 4; ++flags) {
            P.flags = flags;
            cam.mode = (flags & OS_FLAG_DEPTH_GREY) ? 0u : 1u;
            const uint64_t r = os_render_two_level(top, &mesh, inst, first, wv, wi, mats, wm, &cam, &P,
                                                   0, 0, W, H, film, nthreads, 3 * n_tri);
            CHECK(r >= (uint64_t)W * H);
        }
        free(wm); free(wi); free(wv);
    }

    free(film); free(img); free(rgba); free(any); free(h1); free(h0); free(rays);
    free(nodes2); free(leaf2); free(nodes); free(leaf); free(mat_ids); free(idx); free(vtx);
}

int main(void)
{
    const uint32_t sizes[] = { 1, 2, 3, 17, 500, 4000 };
    for (unsigned s = 0; s < sizeof sizes / sizeof sizes[0]; ++s)
        for (int sah = 0; sah < 2; ++sah) run_scene(sizes[s], sah, s & 1 ? 4 : 1);
    puts("oracle clean under ASan + UBSan");
    return 0;
}
