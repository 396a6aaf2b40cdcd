// Two-level scenes: the world-space shading record of every instanced triangle, written on the
// GPU (one 48-byte record per instanced triangle is the one per-instance cost of a placement).
#include "kernels.h"

#include <algorithm>

namespace mipt {

namespace {

// One workgroup column per instance (blockIdx.y), threads over the mesh's triangles.
// world = ((m0*x + m1*y) + m2*z) + m3 per row, in binary32 without contraction: the operation
// order every level of the stack uses for instance transforms.
__global__ __launch_bounds__(256) void k_instance_shading_records(DeviceScene s,
                                                                  const float4 *object_to_world,
                                                                  const int32_t *mesh_mat_ids,
                                                                  const int32_t *material_offset,
                                                                  float4 *stris) {
    const uint32_t inst = blockIdx.y;
    const DeviceInstance in = s.instances[inst];
    const float4 r0 = object_to_world[3 * (size_t)inst + 0], r1 = object_to_world[3 * (size_t)inst + 1],
                 r2 = object_to_world[3 * (size_t)inst + 2];
    const int32_t moff = material_offset[inst];
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < in.n_tri; t += gridDim.x * 256) {
        const int32_t *ip = s.idx + 3 * ((size_t)in.tri_off + t);
        float w[3][3];
        for (int j = 0; j < 3; ++j) {
            const float *p = s.vtx + 3 * ((size_t)in.vtx_off + (size_t)ip[j]);
            const float x = p[0], y = p[1], z = p[2];
            w[j][0] = ((r0.x * x + r0.y * y) + r0.z * z) + r0.w;
            w[j][1] = ((r1.x * x + r1.y * y) + r1.z * z) + r1.w;
            w[j][2] = ((r2.x * x + r2.y * y) + r2.z * z) + r2.w;
        }
        const int32_t mid = (mesh_mat_ids ? mesh_mat_ids[(size_t)in.tri_off + t] : 0) + moff;
        float4 *out = stris + 3 * ((size_t)in.first_tri + t);
        out[0] = make_float4(w[0][0], w[0][1], w[0][2], w[1][0]);
        out[1] = make_float4(w[1][1], w[1][2], w[2][0], w[2][1]);
        out[2] = make_float4(w[2][2], __int_as_float(mid), 0.f, 0.f);
    }
}

// Child indices of one mesh's two-child nodes from mesh-local to scene-wide (see
// enter_instance): inner child + node_base; leaf child ~(16-byte position) + record size * record base.
__global__ __launch_bounds__(256) void k_rebase_children(int4 *child_words, uint32_t n_nodes,
                                                         int32_t node_base, int32_t tri_base16) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_nodes) return;
    int4 c = child_words[4 * (size_t)i + 3];
    c.x = c.x >= 0 ? c.x + node_base : ~(~c.x + tri_base16);
    c.y = c.y >= 0 ? c.y + node_base : ~(~c.y + tri_base16);
    child_words[4 * (size_t)i + 3] = c;
}

} // namespace

void launch_rebase_children(hipStream_t st, float4 *wnodes, uint32_t n_nodes, uint32_t node_base,
                            uint32_t tri_record_base) {
    if (!n_nodes) return;
    hipLaunchKernelGGL(k_rebase_children, dim3((n_nodes + 255) / 256), dim3(256), 0, st,
                       reinterpret_cast<int4 *>(wnodes), n_nodes, (int32_t)node_base,
                       (int32_t)(PT_TRI_RECORD_16B * tri_record_base));
}

void launch_instance_shading_records(hipStream_t st, const DeviceScene &s,
                                     const float4 *object_to_world, const int32_t *mesh_mat_ids,
                                     const int32_t *material_offset, uint32_t max_mesh_tris,
                                     float4 *stris) {
    if (!s.n_instances || !max_mesh_tris) return;
    const uint32_t bx = std::min<uint32_t>((max_mesh_tris + 255) / 256, 1024);
    // the y dimension of a grid holds 65535 workgroups: instances go in slices of that size
    for (uint32_t first = 0; first < s.n_instances; first += 65535) {
        DeviceScene part = s;
        part.instances = s.instances + first;
        const uint32_t count = std::min<uint32_t>(65535, s.n_instances - first);
        hipLaunchKernelGGL(k_instance_shading_records, dim3(bx, count), dim3(256), 0, st, part,
                           object_to_world + 3 * (size_t)first, mesh_mat_ids,
                           material_offset + first, stris);
    }
}

} // namespace mipt
