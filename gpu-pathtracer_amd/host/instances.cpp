// Affine transforms of two-level scenes: the arithmetic every level of the stack (library,
// kernels, CPU checker) must agree on bit for bit, so each expression's order is fixed here.
#include "bvh.h"

#include <cmath>

namespace mipt {

// Inverse of the 3x4 transform m (three rows of linear part | translation): cofactors and
// determinant in binary64, one rounding to binary32 per entry.
bool invert_affine(const float m[12], float inv[12]) {
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9],
                 i = m[10];
    const double tx = m[3], ty = m[7], tz = m[11];
    const double A = e * i - f * h, B = f * g - d * i, C = d * h - e * g;
    const double det = (a * A + b * B) + c * C;
    if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
    const double r[3][3] = {
        {A / det, (c * h - b * i) / det, (b * f - c * e) / det},
        {B / det, (a * i - c * g) / det, (c * d - a * f) / det},
        {C / det, (b * g - a * h) / det, (a * e - b * d) / det},
    };
    for (int k = 0; k < 3; ++k) {
        inv[4 * k + 0] = (float)r[k][0];
        inv[4 * k + 1] = (float)r[k][1];
        inv[4 * k + 2] = (float)r[k][2];
        inv[4 * k + 3] = (float)(-((r[k][0] * tx + r[k][1] * ty) + r[k][2] * tz));
    }
    for (int k = 0; k < 12; ++k)
        if (!std::isfinite(inv[k])) return false;
    return true;
}

void transform_point(const float m[12], const float p[3], float out[3]) {
    for (int k = 0; k < 3; ++k)
        out[k] = ((m[4 * k + 0] * p[0] + m[4 * k + 1] * p[1]) + m[4 * k + 2] * p[2]) + m[4 * k + 3];
}

// World-space box of an instance: the eight corners of its mesh's root box, transformed.
void instance_world_box(const float m[12], const pt_bbox &root, pt_bbox &out) {
    for (int k = 0; k < 3; ++k) {
        out.pmin[k] = INFINITY;
        out.pmax[k] = -INFINITY;
    }
    out.pmin[3] = out.pmax[3] = 0.f;
    for (int corner = 0; corner < 8; ++corner) {
        const float p[3] = {(corner & 1) ? root.pmax[0] : root.pmin[0],
                            (corner & 2) ? root.pmax[1] : root.pmin[1],
                            (corner & 4) ? root.pmax[2] : root.pmin[2]};
        float w[3];
        transform_point(m, p, w);
        for (int k = 0; k < 3; ++k) {
            out.pmin[k] = std::fmin(out.pmin[k], w[k]);
            out.pmax[k] = std::fmax(out.pmax[k], w[k]);
        }
    }
}

} // namespace mipt
