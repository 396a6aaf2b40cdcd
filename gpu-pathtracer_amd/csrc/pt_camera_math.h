// Camera-ray construction shared by pt_raygen and the path tracer's raygen.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mipt.h"

namespace mipt {

// (fx, fy) is the sample position in pixels, x to the right, y down from the top row.
// PT_CAMERA_REFERENCE: shaders/depth.frag:47-83, pixel -> texCoords as
// shaders/quad.vert:6,14 (texCoords = position / size).
// PT_CAMERA_PINHOLE: d = normalize(c0*x*sx + c1*y*sy - c2), x,y in [-1,1], y up.
__device__ __forceinline__ void make_camera_ray(const pt_camera &cam, uint32_t width,
                                                uint32_t height, float fx, float fy,
                                                float4 &o, float4 &d) {
    if (cam.mode == PT_CAMERA_REFERENCE) {
        const float u = fx / (float)width;
        const float v = fy / (float)height;
        float dxyx = u - 0.5f, dxyy = v - 0.5f;      // :55
        dxyx *= cam.sx;                               // :59
        dxyy *= cam.sy;                               // :60
        const float vz = -1.0f;                       // :65
        float dx = (cam.c0[0] * dxyx + cam.c1[0] * dxyy) + cam.c2[0] * vz;   // :74
        float dy = (cam.c0[1] * dxyx + cam.c1[1] * dxyy) + cam.c2[1] * vz;
        float dz = (cam.c0[2] * dxyx + cam.c1[2] * dxyy) + cam.c2[2] * vz;
        const float len = sqrtf((dx * dx + dy * dy) + dz * dz);             // :75
        dx /= len; dy /= len; dz /= len;
        const float scale = 1.0f / 20.0f;             // :81
        float ox = cam.eye[0], oy = cam.eye[1], oz = cam.eye[2], ow = 1000.0f;   // :53,70
        oy += 2.0f;                                   // :82
        ox /= scale; oy /= scale; oz /= scale; ow /= scale;   // :83 (vec4 divide)
        o = make_float4(ox, oy, oz, ow);
        d = make_float4(dx, dy, dz, 0.0f);            // :66
    } else {
        const float x = (2.0f * fx) / (float)width - 1.0f;
        const float y = 1.0f - (2.0f * fy) / (float)height;
        const float ax = x * cam.sx, ay = y * cam.sy;
        float dx = (cam.c0[0] * ax + cam.c1[0] * ay) - cam.c2[0];
        float dy = (cam.c0[1] * ax + cam.c1[1] * ay) - cam.c2[1];
        float dz = (cam.c0[2] * ax + cam.c1[2] * ay) - cam.c2[2];
        const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
        dx /= len; dy /= len; dz /= len;
        o = make_float4(cam.eye[0], cam.eye[1], cam.eye[2], 1e30f);
        d = make_float4(dx, dy, dz, 0.0f);
    }
}

// GLSL smoothstep(edge0, edge1, x)
__device__ __forceinline__ float smoothstep_glsl(float e0, float e1, float x) {
    const float t = fminf(fmaxf((x - e0) / (e1 - e0), 0.0f), 1.0f);
    return (t * t) * (3.0f - 2.0f * t);
}

// The colour the reference writes for a pixel, shaders/depth.frag:90-107 (the primid hash
// at :99-104 is overwritten at :105, so only the grey survives).
__device__ __forceinline__ float4 depth_grey(int32_t shapeid, float t) {
    if (shapeid == -1) return make_float4(0.f, 0.f, 0.f, 1.f);
    const float scale = 1.0f / 20.0f;
    float x = t;
    x *= scale;
    x = smoothstep_glsl(x, 0.0f, 3.0f);   // argument order as written at :96
    return make_float4(x, x, x, 1.f);
}

} // namespace mipt
