// Emissive-triangle list for next-event estimation.
//
// The reference has no lights or materials (its shader writes a depth grey,
// shaders/depth.frag:90-107); sampling emitters directly is the "NEE" step of the loop
// its author sketched at src/main_granite.cpp:411-412.  The list is part of this
// project's shading specification: the order of the entries and the binary32 running sum
// are reproduced by the CPU restatement bit for bit.
#include <cmath>

#include "bvh.h"

namespace mipt {

void build_light_list(const float *vtx, const int32_t *idx, uint32_t n_tri,
                      const pt_material *materials, const int32_t *material_ids,
                      std::vector<int32_t> &tri, std::vector<float> &cdf) {
    tri.clear();
    cdf.clear();
    float total = 0.0f;
    for (uint32_t t = 0; t < n_tri; ++t) {
        const pt_material &m = materials[material_ids[t]];
        if (!((m.ke[0] + m.ke[1]) + m.ke[2] > 0.0f)) continue;
        const float *a = vtx + 3 * (size_t)idx[3 * (size_t)t + 0];
        const float *b = vtx + 3 * (size_t)idx[3 * (size_t)t + 1];
        const float *c = vtx + 3 * (size_t)idx[3 * (size_t)t + 2];
        const float e1x = b[0] - a[0], e1y = b[1] - a[1], e1z = b[2] - a[2];
        const float e2x = c[0] - a[0], e2y = c[1] - a[1], e2z = c[2] - a[2];
        const float nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z,
                    nz = e1x * e2y - e1y * e2x;
        const float len2 = (nx * nx + ny * ny) + nz * nz;
        if (!(len2 > 0.0f)) continue;
        total += 0.5f * std::sqrt(len2);
        tri.push_back((int32_t)t);
        cdf.push_back(total);
    }
}

} // namespace mipt
