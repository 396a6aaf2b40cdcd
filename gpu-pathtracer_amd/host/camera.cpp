// Host-side construction of pt_camera, the camera as the shader consumes it
// (shaders/depth.frag:59-60,70,74), from the three forms the reference's host code holds one in:
// the matrices of Granite's `global` uniform block, the Camera struct of src/camera.h, and the
// look-at call of the live application.  Arithmetic in binary64, rounded once to binary32.
#include <cmath>
#include <cstring>
#include <string>

#include "../../include/mipt.h"

namespace mipt {
int fail(int code, const std::string &msg);
}

namespace {

bool finite3(const double *v) { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); }

bool normalize3(double *v) {
    const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (!(n > 0.0) || !std::isfinite(n)) return false;
    v[0] /= n; v[1] /= n; v[2] /= n;
    return true;
}

void cross3(const double *a, const double *b, double *c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

bool mode_ok(uint32_t mode) { return mode == PT_CAMERA_REFERENCE || mode == PT_CAMERA_PINHOLE; }

void fill(pt_camera *out, const double *eye, const double *right, const double *up, const double *back,
          double sx, double sy, uint32_t mode) {
    std::memset(out, 0, sizeof(*out));
    for (int k = 0; k < 3; ++k) {
        out->eye[k] = (float)eye[k];
        out->c0[k] = (float)right[k];
        out->c1[k] = (float)up[k];
        out->c2[k] = (float)back[k];
    }
    out->sx = (float)sx;
    out->sy = (float)sy;
    out->mode = mode;
}

} // namespace

extern "C" {

int pt_camera_from_matrices(const float *inv_view, const float *inv_projection,
                            const float *camera_position, uint32_t mode, pt_camera *out) {
    if (!inv_view || !inv_projection || !out)
        return mipt::fail(PT_ERR_INVALID, "pt_camera_from_matrices: null argument");
    if (!mode_ok(mode)) return mipt::fail(PT_ERR_INVALID, "pt_camera_from_matrices: unknown camera mode");
    // column-major 4x4, as GLSL stores a mat4: mat3(inv_view) takes the upper-left 3x3, whose
    // columns are elements 0-2, 4-6, 8-10; the position is the fourth column unless
    // camera_position (the block's own field, depth.frag:70) is given
    double eye[3], c[3][3];
    for (int col = 0; col < 3; ++col)
        for (int k = 0; k < 3; ++k) c[col][k] = inv_view[4 * col + k];
    for (int k = 0; k < 3; ++k) eye[k] = camera_position ? camera_position[k] : inv_view[12 + k];
    const double sx = inv_projection[0], sy = inv_projection[5];      // [0][0] and [1][1], depth.frag:59-60
    if (!finite3(eye) || !finite3(c[0]) || !finite3(c[1]) || !finite3(c[2]) || !std::isfinite(sx) ||
        !std::isfinite(sy))
        return mipt::fail(PT_ERR_INVALID, "pt_camera_from_matrices: a matrix entry is not finite");
    fill(out, eye, c[0], c[1], c[2], sx, sy, mode);
    return PT_OK;
}

int pt_camera_from_orientation(const float *orientation_xyzw, const float *eye_pos, float fovy_radians,
                               uint32_t viewport_width, uint32_t viewport_height, uint32_t mode,
                               pt_camera *out) {
    if (!orientation_xyzw || !eye_pos || !out)
        return mipt::fail(PT_ERR_INVALID, "pt_camera_from_orientation: null argument");
    if (!mode_ok(mode)) return mipt::fail(PT_ERR_INVALID, "pt_camera_from_orientation: unknown camera mode");
    if (viewport_width == 0 || viewport_height == 0)
        return mipt::fail(PT_ERR_INVALID, "pt_camera_from_orientation: empty viewport");
    if (!(fovy_radians > 0.f) || !(fovy_radians < 3.14159265f))
        return mipt::fail(PT_ERR_INVALID, "pt_camera_from_orientation: field of view outside (0, pi)");
    double q[4] = {orientation_xyzw[0], orientation_xyzw[1], orientation_xyzw[2], orientation_xyzw[3]};
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (!(n > 0.0) || !std::isfinite(n))
        return mipt::fail(PT_ERR_INVALID, "pt_camera_from_orientation: the quaternion has no direction");
    for (double &v : q) v /= n;
    // Camera::getRightVector / getUpVector / getForwardVector (src/camera.h:20-22): the quaternion
    // takes camera axes to world axes; forward is -z, so the third column (back) is orientation * +z
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double right[3] = {1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)};
    const double up[3] = {2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)};
    const double back[3] = {2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)};
    const double eye[3] = {eye_pos[0], eye_pos[1], eye_pos[2]};
    if (!finite3(eye)) return mipt::fail(PT_ERR_INVALID, "pt_camera_from_orientation: eye position is not finite");
    // half of the legacy image_size_norm = 2 tan(fovy / 2) * (aspect, 1), src/main.cpp:1304-1309
    const double t = std::tan(0.5 * (double)fovy_radians);
    fill(out, eye, right, up, back, t * (double)viewport_width / (double)viewport_height, t, mode);
    return PT_OK;
}

int pt_camera_look_at(const float *eye, const float *target, const float *up, float fovy_radians,
                      float aspect, uint32_t mode, pt_camera *out) {
    if (!eye || !target || !out) return mipt::fail(PT_ERR_INVALID, "pt_camera_look_at: null argument");
    if (!mode_ok(mode)) return mipt::fail(PT_ERR_INVALID, "pt_camera_look_at: unknown camera mode");
    if (!(fovy_radians > 0.f) || !(fovy_radians < 3.14159265f) || !(aspect > 0.f) || !std::isfinite(aspect))
        return mipt::fail(PT_ERR_INVALID, "pt_camera_look_at: field of view outside (0, pi) or aspect not positive");
    const double e[3] = {eye[0], eye[1], eye[2]};
    double back[3] = {e[0] - (double)target[0], e[1] - (double)target[1], e[2] - (double)target[2]};
    double world_up[3] = {0.0, 1.0, 0.0};
    if (up) { world_up[0] = up[0]; world_up[1] = up[1]; world_up[2] = up[2]; }
    double right[3], upv[3];
    if (!finite3(e) || !normalize3(back))
        return mipt::fail(PT_ERR_INVALID, "pt_camera_look_at: eye and target coincide or are not finite");
    cross3(world_up, back, right);
    if (!normalize3(right))
        return mipt::fail(PT_ERR_INVALID, "pt_camera_look_at: the up vector is parallel to the viewing direction");
    cross3(back, right, upv);
    const double t = std::tan(0.5 * (double)fovy_radians);
    fill(out, e, right, upv, back, t * (double)aspect, t, mode);
    return PT_OK;
}

} // extern "C"
