// The mesh object behind pt_mesh (filled by the OBJ and glTF readers).
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/mipt.h"

struct pt_mesh {
    std::vector<float> vtx;             // xyz packed
    std::vector<int32_t> idx;           // 3 per triangle
    std::vector<int32_t> mat_ids;       // 1 per triangle
    // OBJ files: the shape (o / g group) each triangle belongs to and its position among that
    // shape's triangles -- the shape_id / prim_id of the reference's Face records
    // (src/bvh/RadeonRays/intersector_skip_links.h:82-92).  Empty for other sources.
    std::vector<int32_t> shape_ids, shape_prim_ids;
    uint32_t n_shapes = 0;
    std::vector<pt_material> materials;
};

// A scene kept as structured: every mesh once, in its own space, plus the instances that
// place them (filled by pt_gltf_load_model).
struct pt_model {
    std::vector<pt_mesh *> meshes;          // owned; their mat_ids index `materials`
    std::vector<pt_instance> instances;
    std::vector<pt_material> materials;
    ~pt_model() {
        for (pt_mesh *m : meshes) delete m;
    }
};
