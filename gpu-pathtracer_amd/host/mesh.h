// The mesh object behind pt_mesh (filled by the OBJ and glTF readers).
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/mipt.h"

struct pt_mesh {
    std::vector<float> vtx;             // xyz packed
    std::vector<int32_t> idx;           // 3 per triangle
    std::vector<int32_t> mat_ids;       // 1 per triangle
    std::vector<pt_material> materials;
};
