// BVH construction on the GPU: the whole of build_mesh_bvh + init_device_data
// (src/main_granite.cpp:115-194) without the host in the loop.
//
//   leaf boxes          src/main_granite.cpp:162-172   (min/max of the three vertices)
//   inflation           src/bvh/bvh_builder.cpp:53-59  (same binary32 expression as the host)
//   tree                a linear BVH: triangles ordered by the 63-bit Morton code of their box
//                       centre, binary radix tree over the codes (Karras 2012).  The topology
//                       RadeonRays::Bvh::Build would produce is external to the reference, and
//                       hit results do not depend on it except between exact ties.
//   boxes               union of the children, one launch per tree level, deepest first
//   output              the 32-byte skip-link array of src/bvh/bvh_builder.h:26-34 in pre-order
//                       (PlainBvhTranslator::Process, plain_bvh_translator.cpp:39-96), plus the
//                       64-byte two-child nodes and 48-byte triangle records of the traversal.
//
// No thread reads what another thread of the same launch wrote: every hand-off crosses a
// kernel boundary, so no in-launch fences or cache-bypassing accesses are needed and the
// result is deterministic (sort keys are made unique by the triangle index).
//
// Pre-order position of a node without a traversal: if the node covers the sorted leaves
// [first, last] and `lt` of its ancestors hold it in their LEFT subtree, everything left of it
// (2*first nodes, less one per right turn) plus its ancestors (left + right turns) precede it:
// position = 2*first + lt.  Among inner nodes alone it is first + lt; among leaves, the leaf's
// sorted index.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>
#include <string>

#include "build.h"

namespace mipt {

namespace {

constexpr int kB = 256;
constexpr int kMaxDepth = 128;     // 63 code bits + 32 index bits bound the radix tree's depth

__device__ __forceinline__ unsigned int ordered(float f) {      // monotonic float -> uint
    const unsigned int b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float unordered(unsigned int u) {
    const unsigned int b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(b);
#else
    float f;
    memcpy(&f, &b, 4);
    return f;
#endif
}

__device__ __forceinline__ float encode_index(int32_t v, uint32_t enc) {
    return enc == PT_LINKS_INT_BITS ? __int_as_float(v) : (float)v;
}

// ---- 0. triangle splitting ---------------------------------------------------------------------
// The reference's builder has a split path (bvh.sah.use_splits -> SplitBvh,
// src/bvh/bvh_builder.cpp:38-44): a triangle whose box is mostly empty goes into several
// leaves with smaller boxes.  Here the decision is made per triangle before the sort: a triangle
// whose box has `ratio` times the area its own projections need (long, thin and diagonal: the
// slivers of a fan around a pole, a cone's tip, a cylinder's cap) is cut along the longest axis
// of its box into equal slabs, one leaf each, whose boxes bound the triangle's part inside the
// slab.  A fan of thousands of slivers is thousands of boxes that all cover the fan; cut this way
// they are strings of small boxes along the slivers, and a ray through the fan meets tens of them.
// The traversals take a triangle in several leaves as they take the host builder's.
//
// Half the surface area of a box, and the same for the "box" a triangle cannot do without: the
// sum of its projections on the three coordinate planes (half the components of |e1 x e2|).
__device__ __forceinline__ int split_pieces(const float *a, const float *b, const float *c,
                                            float ratio, int max_pieces, float scale, uint32_t t) {
    float ext[3], n[3];
    for (int k = 0; k < 3; ++k)
        ext[k] = fmaxf(fmaxf(a[k], b[k]), c[k]) - fminf(fminf(a[k], b[k]), c[k]);
    const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    const float e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    n[0] = e1[1] * e2[2] - e1[2] * e2[1];
    n[1] = e1[2] * e2[0] - e1[0] * e2[2];
    n[2] = e1[0] * e2[1] - e1[1] * e2[0];
    const float box = (ext[0] * ext[1] + ext[0] * ext[2]) + ext[1] * ext[2];
    const float need = 0.5f * ((fabsf(n[0]) + fabsf(n[1])) + fabsf(n[2]));
    if (!(box > 0.f) || !(box < INFINITY) || !(need == need)) return 1;      // a point, a segment along an axis, not finite
    // pieces = box / (ratio * need), at least 1: a piece's box then has about `ratio` times
    // what its part of the triangle needs
    float want = need > 0.f ? box / (ratio * need) : (float)max_pieces;
    want = fminf(fmaxf(want, 1.f), (float)max_pieces);
    const int k = (int)want;
    if (scale >= 1.f) return k;
    // over the budget: every triangle's surplus scaled down alike, the fractions rounded up or
    // down by a fixed sequence over the triangle numbers, so that they add up and no run differs
    const float u = (float)((t * 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f);
    return 1 + (int)((float)(k - 1) * scale + u);
}

// counts[t] = leaves of triangle t (counts[n] = 0, so that the scan's last entry is the total);
// extra: sum of (leaves - 1), for the budget.
__global__ __launch_bounds__(kB) void k_split_count(const float *vtx, const int32_t *idx, uint32_t n,
                                                    float ratio, int max_pieces, float scale,
                                                    uint32_t *counts, unsigned long long *extra) {
    const uint32_t t = blockIdx.x * kB + threadIdx.x;
    uint32_t more = 0;
    if (t < n) {
        const int32_t i0 = idx[3 * (size_t)t], i1 = idx[3 * (size_t)t + 1], i2 = idx[3 * (size_t)t + 2];
        const int k = split_pieces(vtx + 3 * (size_t)i0, vtx + 3 * (size_t)i1, vtx + 3 * (size_t)i2, ratio,
                                   max_pieces, scale, t);
        counts[t] = (uint32_t)k;
        more = (uint32_t)k - 1u;
    } else if (t == n) {
        counts[t] = 0;
    }
    unsigned long long w = more;
    for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
    if ((threadIdx.x & 63) == 0 && w) atomicAdd(extra, w);
}

// Box of the part of triangle (a, b, c) between the planes x[axis] = s0 and x[axis] = s1: the
// vertices inside the slab and the points where the edges cross its two planes.  The crossing
// points are computed with rounding, so the box is widened by a few units in the last place of
// the coordinates involved and then cut back to the triangle's own box, which holds every point
// of the triangle exactly.
__device__ __forceinline__ void slab_box(const float *a, const float *b, const float *c, int axis,
                                         float s0, float s1, float *mn, float *mx) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    const float *v[3] = {a, b, c};
    for (int i = 0; i < 3; ++i) {
        const float *p = v[i], *q = v[i == 2 ? 0 : i + 1];
        if (p[axis] >= s0 && p[axis] <= s1)
            for (int k = 0; k < 3; ++k) {
                lo[k] = fminf(lo[k], p[k]);
                hi[k] = fmaxf(hi[k], p[k]);
            }
        for (int side = 0; side < 2; ++side) {
            const float s = side ? s1 : s0;
            const float dp = p[axis] - s, dq = q[axis] - s;
            if ((dp < 0.f && dq > 0.f) || (dp > 0.f && dq < 0.f)) {
                const float u = dp / (dp - dq);
                for (int k = 0; k < 3; ++k) {
                    const float x = k == axis ? s : p[k] + (q[k] - p[k]) * u;
                    lo[k] = fminf(lo[k], x);
                    hi[k] = fmaxf(hi[k], x);
                }
            }
        }
    }
    for (int k = 0; k < 3; ++k) {
        const float tmin = fminf(fminf(a[k], b[k]), c[k]), tmax = fmaxf(fmaxf(a[k], b[k]), c[k]);
        if (!(lo[k] <= hi[k])) {               // nothing of the triangle found in the slab: keep it all
            lo[k] = tmin;
            hi[k] = tmax;
        }
        const float pad = ((fabsf(tmin) + fabsf(tmax)) + (tmax - tmin)) * 9.5367431640625e-07f;   // 2^-20
        mn[k] = fmaxf(lo[k] - pad, tmin);
        mx[k] = fminf(hi[k] + pad, tmax);
    }
}

// ---- 1. leaf boxes, centre bounds -----------------------------------------------------
// boxes[r] = inflated box of leaf r (w fields 0).  Without splits leaf r is triangle r; with them
// (first != null) the leaves of triangle t are first[t] .. first[t + 1] - 1 and leaf_tri names
// each leaf's triangle.  bounds[0..5]: ordered-uint min x,y,z and max x,y,z of the box centres;
// bounds[6]: set non-zero if a vertex or box is not finite.
__global__ __launch_bounds__(kB) void k_leaf_boxes(const float *vtx, const int32_t *idx,
                                                   uint32_t n, const uint32_t *first,
                                                   uint32_t *leaf_tri, float4 *boxes,
                                                   unsigned int *bounds) {
    const float kBoundsScale = 1.f + 1e-4f;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool bad = false;
    for (uint32_t t = blockIdx.x * kB + threadIdx.x; t < n; t += gridDim.x * kB) {
        const int32_t i0 = idx[3 * (size_t)t], i1 = idx[3 * (size_t)t + 1], i2 = idx[3 * (size_t)t + 2];
        const float *a = vtx + 3 * (size_t)i0, *b = vtx + 3 * (size_t)i1, *c = vtx + 3 * (size_t)i2;
        const uint32_t r0 = first ? first[t] : t;
        const uint32_t pieces = first ? first[t + 1] - r0 : 1u;
        int axis = 0;
        float t_lo[3], t_hi[3];
        for (int k = 0; k < 3; ++k) {
            t_lo[k] = fminf(fminf(a[k], b[k]), c[k]);
            t_hi[k] = fmaxf(fmaxf(a[k], b[k]), c[k]);
            bad |= !(fabsf((a[k] + b[k]) + c[k]) < INFINITY);   // min/max would skip a NaN vertex
        }
        if (t_hi[1] - t_lo[1] > t_hi[axis] - t_lo[axis]) axis = 1;
        if (t_hi[2] - t_lo[2] > t_hi[axis] - t_lo[axis]) axis = 2;
        const float ext = t_hi[axis] - t_lo[axis];
        // Planes at equal steps.  (Spacing them quadratically from the corner opposite the
        // shortest edge -- the corner the slivers of a fan share, so that the place where all
        // their boxes meet shrinks -- was measured: the 65 K-triangle mesh of the two-level
        // benchmark gained 7 %, the 28 M-triangle surface lost 20 %: not kept.)
        auto plane = [&](uint32_t j) {             // the same expression for both slabs it bounds
            return t_lo[axis] + ext * ((float)j / (float)pieces);
        };
        for (uint32_t j = 0; j < pieces; ++j) {
            float p_lo[3] = {t_lo[0], t_lo[1], t_lo[2]}, p_hi[3] = {t_hi[0], t_hi[1], t_hi[2]};
            if (pieces > 1) {
                const float s0 = j == 0 ? t_lo[axis] : plane(j);
                const float s1 = j + 1 == pieces ? t_hi[axis] : plane(j + 1);
                slab_box(a, b, c, axis, s0, s1, p_lo, p_hi);
            }
            float mn[3], mx[3];
            for (int k = 0; k < 3; ++k) {
                const float pmin = p_lo[k], pmax = p_hi[k];
                const float ctr = 0.5f * (pmax + pmin);                    // bvh_builder.cpp:55-58
                mn[k] = ctr + (pmin - ctr) * kBoundsScale;
                mx[k] = ctr + (pmax - ctr) * kBoundsScale;
                const float m = 0.5f * (mn[k] + mx[k]);
                lo[k] = fminf(lo[k], m);
                hi[k] = fmaxf(hi[k], m);
                bad |= !(fabsf(mn[k]) < INFINITY) || !(fabsf(mx[k]) < INFINITY);
            }
            boxes[2 * (size_t)(r0 + j) + 0] = make_float4(mn[0], mn[1], mn[2], 0.f);
            boxes[2 * (size_t)(r0 + j) + 1] = make_float4(mx[0], mx[1], mx[2], 0.f);
            if (leaf_tri) leaf_tri[r0 + j] = t;
        }
    }
    __shared__ float s_lo[3][kB / 64], s_hi[3][kB / 64];
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    for (int k = 0; k < 3; ++k) {
        float l = lo[k], h = hi[k];
        for (int o = 32; o > 0; o >>= 1) {
            l = fminf(l, __shfl_xor(l, o, 64));
            h = fmaxf(h, __shfl_xor(h, o, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            s_lo[k][threadIdx.x >> 6] = l;
            s_hi[k][threadIdx.x >> 6] = h;
        }
    }
    if (bad) s_bad = 1;
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        float l = s_lo[k][0], h = s_hi[k][0];
        for (int w = 1; w < kB / 64; ++w) {
            l = fminf(l, s_lo[k][w]);
            h = fmaxf(h, s_hi[k][w]);
        }
        if (l <= h) {
            atomicMin(&bounds[k], ordered(l));
            atomicMax(&bounds[3 + k], ordered(h));
        }
    }
    if (threadIdx.x == 0 && s_bad) atomicOr(&bounds[6], 1u);
}

// ---- 2. Morton codes ----------------------------------------------------------------------
__device__ __forceinline__ unsigned long long spread21(unsigned int v) {
    unsigned long long x = v & 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(kB) void k_morton(const float4 *boxes, uint32_t n,
                                               const unsigned int *bounds,
                                               unsigned long long *keys, uint32_t *vals) {
    const uint32_t t = blockIdx.x * kB + threadIdx.x;
    if (t >= n) return;
    const float4 mn = boxes[2 * (size_t)t], mx = boxes[2 * (size_t)t + 1];
    const float c[3] = {0.5f * (mn.x + mx.x), 0.5f * (mn.y + mx.y), 0.5f * (mn.z + mx.z)};
    unsigned int q[3];
    for (int k = 0; k < 3; ++k) {
        const float lo = unordered(bounds[k]), hi = unordered(bounds[3 + k]);
        const float ext = hi - lo;
        float x = ext > 0.f ? (c[k] - lo) / ext : 0.f;
        x = fminf(fmaxf(x * 2097152.f, 0.f), 2097151.f);
        q[k] = (unsigned int)x;
    }
    keys[t] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    vals[t] = t;
}

// ---- 3. binary radix tree over the sorted codes (Karras 2012) ------------------------------
// Common-prefix length of sorted positions i and j; equal codes are told apart by position.
__device__ __forceinline__ int prefix_len(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz(i ^ j);
}

// Inner node i (0 <= i < n-1; 0 is the root): children, covered range, and for both children
// the parent word (parent << 1 | is_left).  A child is an inner node index, or ~(leaf position).
__global__ __launch_bounds__(kB) void k_radix_tree(const unsigned long long *keys, int n,
                                                   int32_t *left, int32_t *right, uint32_t *first,
                                                   uint32_t *last, uint32_t *inner_parent,
                                                   uint32_t *leaf_parent) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n - 1) return;
    const int d = prefix_len(keys, n, i, i + 1) > prefix_len(keys, n, i, i - 1) ? 1 : -1;
    const int dmin = prefix_len(keys, n, i, i - d);
    int lmax = 2;
    while (prefix_len(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t > 0; t >>= 1)
        if (prefix_len(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = prefix_len(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (prefix_len(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const bool left_leaf = lo == gamma, right_leaf = hi == gamma + 1;
    left[i] = left_leaf ? ~gamma : gamma;
    right[i] = right_leaf ? ~(gamma + 1) : gamma + 1;
    first[i] = (uint32_t)lo;
    last[i] = (uint32_t)hi;
    if (left_leaf) leaf_parent[gamma] = ((uint32_t)i << 1) | 1u;
    else inner_parent[gamma] = ((uint32_t)i << 1) | 1u;
    if (right_leaf) leaf_parent[gamma + 1] = ((uint32_t)i << 1);
    else inner_parent[gamma + 1] = ((uint32_t)i << 1);
}

// ---- 4. depth and left-turn count of every inner node ---------------------------------------
__global__ __launch_bounds__(kB) void k_depth(const uint32_t *inner_parent, int n_inner,
                                              uint32_t *depth, uint32_t *lt, uint32_t *ids,
                                              unsigned int *histogram) {
    __shared__ unsigned int s_hist[kMaxDepth];
    if (threadIdx.x < kMaxDepth) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i < n_inner) {
        uint32_t d = 0, l = 0;
        for (int cur = i; cur != 0;) {
            const uint32_t p = inner_parent[cur];
            l += p & 1u;
            ++d;
            cur = (int)(p >> 1);
        }
        if (d >= (uint32_t)kMaxDepth) d = kMaxDepth - 1;      // cannot happen (see kMaxDepth)
        depth[i] = d;
        lt[i] = l;
        ids[i] = (uint32_t)i;
        atomicAdd(&s_hist[d], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kMaxDepth && s_hist[threadIdx.x])
        atomicAdd(&histogram[threadIdx.x], s_hist[threadIdx.x]);
}

// ---- 5. boxes of one level --------------------------------------------------------------------
// ibox: 2 x float4 per inner node.  Children one level down were written by the previous launch.
__global__ __launch_bounds__(kB) void k_refit_level(const uint32_t *level_ids, uint32_t count,
                                                    const int32_t *left, const int32_t *right,
                                                    const uint32_t *vals, const float4 *boxes,
                                                    float4 *ibox) {
    const uint32_t k = blockIdx.x * kB + threadIdx.x;
    if (k >= count) return;
    const uint32_t i = level_ids[k];
    const int32_t cl = left[i], cr = right[i];
    const float4 *bl = cl >= 0 ? ibox + 2 * (size_t)cl : boxes + 2 * (size_t)vals[~cl];
    const float4 *br = cr >= 0 ? ibox + 2 * (size_t)cr : boxes + 2 * (size_t)vals[~cr];
    const float4 lmin = bl[0], lmax = bl[1], rmin = br[0], rmax = br[1];
    ibox[2 * (size_t)i + 0] = make_float4(fminf(lmin.x, rmin.x), fminf(lmin.y, rmin.y),
                                          fminf(lmin.z, rmin.z), 0.f);
    ibox[2 * (size_t)i + 1] = make_float4(fmaxf(lmax.x, rmax.x), fmaxf(lmax.y, rmax.y),
                                          fmaxf(lmax.z, rmax.z), 0.f);
}

// ---- 6. output ------------------------------------------------------------------------------------
struct EmitArgs {
    const int32_t *left, *right;
    const uint32_t *first, *last, *lt, *leaf_parent, *vals;
    const float4 *boxes, *ibox;
    const float *vtx;
    const int32_t *idx;
    float4 *nodes;     // 2 x float4 per skip-link node
    float4 *wnodes;    // 4 x float4 per two-child node
    float4 *wtris;     // 3 x float4 per triangle record
    float4 *stris;     // 3 x float4 per shading record, by triangle id
    const int32_t *mat_ids;
    const uint32_t *leaf_tri;   // leaf -> triangle, null when every triangle has one leaf
    const uint32_t *tri_first;  // triangle -> its first leaf (with leaf_tri)
    uint32_t n;        // leaves
    uint32_t enc;
};

__global__ __launch_bounds__(kB) void k_emit_inner(EmitArgs a) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= a.n - 1) return;
    const uint32_t total = 2 * a.n - 1;
    const uint32_t f = a.first[i], cnt = a.last[i] - f + 1, lt = a.lt[i];
    const uint32_t pos = 2 * f + lt;
    const uint32_t next = pos + 2 * cnt - 1;
    const float4 mn = a.ibox[2 * (size_t)i], mx = a.ibox[2 * (size_t)i + 1];
    a.nodes[2 * (size_t)pos + 0] = make_float4(mn.x, mn.y, mn.z, encode_index(-1, a.enc));
    a.nodes[2 * (size_t)pos + 1] =
        make_float4(mx.x, mx.y, mx.z, encode_index(next < total ? (int32_t)next : -1, a.enc));
    // two-child node, numbered among inner nodes in pre-order
    const int32_t cl = a.left[i], cr = a.right[i];
    const float4 *bl = cl >= 0 ? a.ibox + 2 * (size_t)cl : a.boxes + 2 * (size_t)a.vals[~cl];
    const float4 *br = cr >= 0 ? a.ibox + 2 * (size_t)cr : a.boxes + 2 * (size_t)a.vals[~cr];
    const float4 lmin = bl[0], lmax = bl[1], rmin = br[0], rmax = br[1];
    const int32_t ref_l = cl >= 0 ? (int32_t)(a.first[cl] + lt + 1) : ~(PT_TRI_RECORD_16B * (~cl));
    const int32_t ref_r = cr >= 0 ? (int32_t)(a.first[cr] + lt) : ~(PT_TRI_RECORD_16B * (~cr));
    float4 *w = a.wnodes + 4 * (size_t)(f + lt);
    w[0] = make_float4(lmin.x, lmax.x, lmin.y, lmax.y);
    w[1] = make_float4(rmin.x, rmax.x, rmin.y, rmax.y);
    w[2] = make_float4(lmin.z, lmax.z, rmin.z, rmax.z);
    w[3] = make_float4(__int_as_float(ref_l), __int_as_float(ref_r), 0.f, 0.f);
}

__global__ __launch_bounds__(kB) void k_emit_leaf(EmitArgs a) {
    const uint32_t l = blockIdx.x * kB + threadIdx.x;
    if (l >= a.n) return;
    const uint32_t total = 2 * a.n - 1;
    const uint32_t p = a.leaf_parent[l];
    const uint32_t pos = 2 * l + a.lt[p >> 1] + (p & 1u);
    const uint32_t ref = a.vals[l];
    const uint32_t tri = a.leaf_tri ? a.leaf_tri[ref] : ref;
    const float4 mn = a.boxes[2 * (size_t)ref], mx = a.boxes[2 * (size_t)ref + 1];
    a.nodes[2 * (size_t)pos + 0] = make_float4(mn.x, mn.y, mn.z, encode_index((int32_t)tri, a.enc));
    a.nodes[2 * (size_t)pos + 1] =
        make_float4(mx.x, mx.y, mx.z, encode_index(pos + 1 < total ? (int32_t)(pos + 1) : -1, a.enc));
    const int32_t i0 = a.idx[3 * (size_t)tri], i1 = a.idx[3 * (size_t)tri + 1],
                  i2 = a.idx[3 * (size_t)tri + 2];
    const float *v0 = a.vtx + 3 * (size_t)i0, *v1 = a.vtx + 3 * (size_t)i1, *v2 = a.vtx + 3 * (size_t)i2;
    float4 *r = a.wtris + PT_TRI_RECORD_16B * (size_t)l;
    r[0] = make_float4(v0[0], v0[1], v0[2], v1[0]);
    r[1] = make_float4(v1[1], v1[2], v2[0], v2[1]);
    r[2] = make_float4(v2[2], __int_as_float((int32_t)tri), 0.f, 0.f);
    // shading copy, by the triangle's first leaf (not wanted for a mesh of a two-level scene)
    if (a.stris && (!a.leaf_tri || a.tri_first[tri] == ref)) {
        float4 *q = a.stris + 3 * (size_t)tri;
        q[0] = r[0];
        q[1] = r[1];
        q[2] = make_float4(v2[2], __int_as_float(a.mat_ids[tri]), 0.f, 0.f);
    }
}

// ---- the other tree: parallel locally-ordered clustering (Meister & Bittner 2018) -------------
// Bottom-up: the clusters (at first the Morton-ordered triangles) each look `radius` places
// to either side for the neighbour whose union with them has the least surface area; pairs
// that choose each other merge into a new node; the survivors are compacted, order kept, and
// the round repeats until one cluster is left.  Trees come out close to a full SAH build's.
//
// Nodes 0..n-1 are the leaves in Morton order, n.. the inner nodes in creation order (each
// round's in one block, so a child always has a smaller number than its parent's round).
// Pairs are ranked by (area, not i^1 == j, |i - j|, min(i, j)): symmetric in the pair and total
// for one cluster's candidates, so the best-ranked pair of a round is always mutual and every
// round merges at least once; the i^1 term pairs up clusters with identical boxes instead of
// merging them one per round.
struct PlocNodes {
    float4 *box;        // 2 x float4 per node
    int32_t *left, *right;
    uint32_t *size;     // leaves below
    uint32_t *pos;      // pre-order position
    uint32_t *before;   // leaves before it in pre-order
};

__global__ __launch_bounds__(kB) void k_ploc_init(uint32_t n, const uint32_t *vals,
                                                  const float4 *boxes, PlocNodes t, uint32_t *cl) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    const uint32_t tri = vals[i];
    t.box[2 * (size_t)i + 0] = boxes[2 * (size_t)tri + 0];
    t.box[2 * (size_t)i + 1] = boxes[2 * (size_t)tri + 1];
    t.size[i] = 1;
    cl[i] = i;
}

__global__ __launch_bounds__(kB) void k_ploc_nearest(uint32_t m, const uint32_t *cl, PlocNodes t,
                                                     int radius, uint32_t *nearest) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= (int)m) return;
    const uint32_t me = cl[i];
    const float4 alo = t.box[2 * (size_t)me], ahi = t.box[2 * (size_t)me + 1];
    float best_area = INFINITY;
    int best_j = -1, best_rank = 0;
    const int j0 = max(i - radius, 0), j1 = min(i + radius, (int)m - 1);
    for (int j = j0; j <= j1; ++j) {
        if (j == i) continue;
        const uint32_t o = cl[j];
        const float4 blo = t.box[2 * (size_t)o], bhi = t.box[2 * (size_t)o + 1];
        const float ex = fmaxf(ahi.x, bhi.x) - fminf(alo.x, blo.x);
        const float ey = fmaxf(ahi.y, bhi.y) - fminf(alo.y, blo.y);
        const float ez = fmaxf(ahi.z, bhi.z) - fminf(alo.z, blo.z);
        const float area = (ex * ey + ex * ez) + ey * ez;
        // rank of the tie-breaks, smaller is better: buddy, distance, lower index of the pair
        const int rank = (((i ^ 1) == j ? 0 : 1) << 30) | (abs(i - j) << 1) | (j < i ? 0 : 1);
        if (best_j < 0 || area < best_area || (area == best_area && rank < best_rank)) {
            best_area = area;
            best_j = j;
            best_rank = rank;
        }
    }
    nearest[i] = (uint32_t)best_j;
}

// merge[i] = 1 if cluster i starts a merged pair (it and its choice chose each other, i the
// lower); keep[i] = 0 only for the upper cluster of such a pair.
__global__ __launch_bounds__(kB) void k_ploc_flags(uint32_t m, const uint32_t *nearest,
                                                   uint32_t *merge, uint32_t *keep) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= m) return;
    const uint32_t j = nearest[i];
    const bool mutual = j < m && nearest[j] == i;
    merge[i] = mutual && i < j;
    keep[i] = !(mutual && i > j);
}

__global__ void k_ploc_count(uint32_t m, const uint32_t *merge, const uint32_t *merge_idx,
                             unsigned int *out) {
    *out = merge_idx[m - 1] + merge[m - 1];
}

__global__ __launch_bounds__(kB) void k_ploc_apply(uint32_t m, const uint32_t *cl,
                                                   const uint32_t *nearest, const uint32_t *merge,
                                                   const uint32_t *merge_idx, const uint32_t *keep,
                                                   const uint32_t *new_pos, uint32_t base, PlocNodes t,
                                                   uint32_t *cl_out) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= m || !keep[i]) return;
    uint32_t id = cl[i];
    if (merge[i]) {
        const uint32_t l = id, r = cl[nearest[i]];
        id = base + merge_idx[i];
        const float4 llo = t.box[2 * (size_t)l], lhi = t.box[2 * (size_t)l + 1];
        const float4 rlo = t.box[2 * (size_t)r], rhi = t.box[2 * (size_t)r + 1];
        t.box[2 * (size_t)id + 0] = make_float4(fminf(llo.x, rlo.x), fminf(llo.y, rlo.y), fminf(llo.z, rlo.z), 0.f);
        t.box[2 * (size_t)id + 1] = make_float4(fmaxf(lhi.x, rhi.x), fmaxf(lhi.y, rhi.y), fmaxf(lhi.z, rhi.z), 0.f);
        t.left[id] = (int32_t)l;
        t.right[id] = (int32_t)r;
        t.size[id] = t.size[l] + t.size[r];
    }
    cl_out[new_pos[i]] = id;
}

// Top-down, one round's nodes per launch, last round first: a node hands its children their
// pre-order position and the number of leaves before them.
__global__ __launch_bounds__(kB) void k_ploc_place(uint32_t first_id, uint32_t count, PlocNodes t,
                                                   uint32_t depth_of_round, unsigned int *max_depth) {
    const uint32_t k = blockIdx.x * kB + threadIdx.x;
    if (k >= count) return;
    const uint32_t id = first_id + k;
    const uint32_t p = t.pos[id], b = t.before[id];
    const uint32_t l = (uint32_t)t.left[id], r = (uint32_t)t.right[id];
    t.pos[l] = p + 1;
    t.before[l] = b;
    t.pos[r] = p + 2 * t.size[l];
    t.before[r] = b + t.size[l];
    (void)depth_of_round;
    (void)max_depth;
}

struct PlocEmitArgs {
    PlocNodes t;
    const uint32_t *vals;
    const float *vtx;
    const int32_t *idx;
    const int32_t *mat_ids;
    const uint32_t *leaf_tri, *tri_first;     // as in EmitArgs
    float4 *nodes, *wnodes, *wtris, *stris;
    uint32_t n, enc;
};

__global__ __launch_bounds__(kB) void k_ploc_emit(PlocEmitArgs a) {
    const uint32_t id = blockIdx.x * kB + threadIdx.x;
    const uint32_t total = 2 * a.n - 1;
    if (id >= total) return;
    const uint32_t pos = a.t.pos[id], before = a.t.before[id];
    const float4 lo = a.t.box[2 * (size_t)id], hi = a.t.box[2 * (size_t)id + 1];
    if (id < a.n) {                                      // leaf: Morton position id
        const uint32_t ref = a.vals[id];
        const uint32_t tri = a.leaf_tri ? a.leaf_tri[ref] : ref;
        a.nodes[2 * (size_t)pos + 0] = make_float4(lo.x, lo.y, lo.z, encode_index((int32_t)tri, a.enc));
        a.nodes[2 * (size_t)pos + 1] =
            make_float4(hi.x, hi.y, hi.z, encode_index(pos + 1 < total ? (int32_t)(pos + 1) : -1, a.enc));
        const int32_t i0 = a.idx[3 * (size_t)tri], i1 = a.idx[3 * (size_t)tri + 1],
                      i2 = a.idx[3 * (size_t)tri + 2];
        const float *v0 = a.vtx + 3 * (size_t)i0, *v1 = a.vtx + 3 * (size_t)i1,
                    *v2 = a.vtx + 3 * (size_t)i2;
        float4 *r = a.wtris + PT_TRI_RECORD_16B * (size_t)before;        // leaf slot = leaves before it
        r[0] = make_float4(v0[0], v0[1], v0[2], v1[0]);
        r[1] = make_float4(v1[1], v1[2], v2[0], v2[1]);
        r[2] = make_float4(v2[2], __int_as_float((int32_t)tri), 0.f, 0.f);
        if (a.stris && (!a.leaf_tri || a.tri_first[tri] == ref)) {
            float4 *q = a.stris + 3 * (size_t)tri;
            q[0] = r[0];
            q[1] = r[1];
            q[2] = make_float4(v2[2], __int_as_float(a.mat_ids[tri]), 0.f, 0.f);
        }
        return;
    }
    const uint32_t next = pos + 2 * a.t.size[id] - 1;
    a.nodes[2 * (size_t)pos + 0] = make_float4(lo.x, lo.y, lo.z, encode_index(-1, a.enc));
    a.nodes[2 * (size_t)pos + 1] =
        make_float4(hi.x, hi.y, hi.z, encode_index(next < total ? (int32_t)next : -1, a.enc));
    const uint32_t l = (uint32_t)a.t.left[id], r = (uint32_t)a.t.right[id];
    const float4 lmin = a.t.box[2 * (size_t)l], lmax = a.t.box[2 * (size_t)l + 1];
    const float4 rmin = a.t.box[2 * (size_t)r], rmax = a.t.box[2 * (size_t)r + 1];
    // inner nodes are numbered among themselves in pre-order: position minus leaves before
    const int32_t ref_l = l >= a.n ? (int32_t)(a.t.pos[l] - a.t.before[l]) : ~(int32_t)(PT_TRI_RECORD_16B * a.t.before[l]);
    const int32_t ref_r = r >= a.n ? (int32_t)(a.t.pos[r] - a.t.before[r]) : ~(int32_t)(PT_TRI_RECORD_16B * a.t.before[r]);
    float4 *w = a.wnodes + 4 * (size_t)(pos - before);
    w[0] = make_float4(lmin.x, lmax.x, lmin.y, lmax.y);
    w[1] = make_float4(rmin.x, rmax.x, rmin.y, rmax.y);
    w[2] = make_float4(lmin.z, lmax.z, rmin.z, rmax.z);
    w[3] = make_float4(__int_as_float(ref_l), __int_as_float(ref_r), 0.f, 0.f);
}


// ---- insertion-based optimisation of the finished tree, in parallel ---------------------------
// After Meister & Bittner 2018 ("Parallel reinsertion for bounding volume hierarchy
// optimization").  Works on the node numbering of the clustering above (leaves 0..n-1, inner
// nodes n..2n-2) with parent links.  One pass:
//   search  every node x looks, from the root down, for the node y next to which it would add
//           the least surface area (area of y+x, plus what y's ancestors grow by), pruning with
//           the best found so far; it wants to move if that beats what it costs where it is
//           (the area of its parent);
//   bid     a move rewires x, its parent p, its sibling, p's parent, y and y's parent.  In each
//           of a few rounds the nodes still in the running bid (round, gain, x) for their six
//           with an atomic max; whoever holds all six has won and keeps them for the rest of
//           the pass, and whoever finds one of its six kept by a winner drops out, which frees
//           the others it had outbid for the next round;
//   accept  winners hold disjoint nodes; one more check keeps subtrees from moving into each
//           other (see k_u_accept);
//   apply   rewires.
// Boxes, depths and the level lists are recomputed between passes.  Every step is
// deterministic (max is order independent), and nothing is handed between threads of a launch.
constexpr int kUMaxDepth = 256;
// Neither kind of pass bounds the depth of the tree; where triangles crowd (fans of slivers) the
// cheapest arrangement is a chain.  Past this depth a node is not moved further down and the
// subtree below a node is not rearranged, which keeps such chains from growing pass after pass.
#ifndef PT_U_SOFT_DEPTH
#define PT_U_SOFT_DEPTH 72
#endif
constexpr int kURounds = 4;          // bidding rounds of a pass (at most 15: four bits of the bid)
#ifndef PT_U_SEARCH_CAP
#define PT_U_SEARCH_CAP 512          // nodes one search may look at
#endif

struct UTree {
    PlocNodes t;
    int32_t *parent;
    uint32_t *height;    // levels below an inner node (a leaf: 0, not stored)
    uint32_t n;          // leaves
    uint32_t total;      // 2n-1
    int32_t root;
};

__device__ __forceinline__ float box_area(const float4 lo, const float4 hi) {
    const float ex = hi.x - lo.x, ey = hi.y - lo.y, ez = hi.z - lo.z;
    return (ex * ey + ex * ez) + ey * ez;
}
__device__ __forceinline__ float union_area(const float4 alo, const float4 ahi, const float4 blo,
                                            const float4 bhi) {
    const float ex = fmaxf(ahi.x, bhi.x) - fminf(alo.x, blo.x);
    const float ey = fmaxf(ahi.y, bhi.y) - fminf(alo.y, blo.y);
    const float ez = fmaxf(ahi.z, bhi.z) - fminf(alo.z, blo.z);
    return (ex * ey + ex * ez) + ey * ez;
}

// The radix tree in the common numbering: inner node i becomes n + i, leaf position l stays l.
__global__ __launch_bounds__(kB) void k_u_from_radix(uint32_t n, const int32_t *left,
                                                     const int32_t *right, const uint32_t *vals,
                                                     const float4 *boxes, const float4 *ibox,
                                                     PlocNodes t) {
    const uint32_t id = blockIdx.x * kB + threadIdx.x;
    if (id >= 2 * n - 1) return;
    if (id < n) {
        const uint32_t tri = vals[id];
        t.box[2 * (size_t)id + 0] = boxes[2 * (size_t)tri + 0];
        t.box[2 * (size_t)id + 1] = boxes[2 * (size_t)tri + 1];
        t.size[id] = 1;
        return;
    }
    const uint32_t i = id - n;
    t.box[2 * (size_t)id + 0] = ibox[2 * (size_t)i + 0];
    t.box[2 * (size_t)id + 1] = ibox[2 * (size_t)i + 1];
    const int32_t l = left[i], r = right[i];
    t.left[id] = l >= 0 ? (int32_t)n + l : ~l;
    t.right[id] = r >= 0 ? (int32_t)n + r : ~r;
}

__global__ __launch_bounds__(kB) void k_u_parents(UTree u) {
    const uint32_t id = u.n + blockIdx.x * kB + threadIdx.x;
    if (id >= u.total) return;
    u.parent[u.t.left[id]] = (int32_t)id;
    u.parent[u.t.right[id]] = (int32_t)id;
}

// depth of every node (root 0); for the inner nodes the identity list to sort by it and the level
// histogram (the level lists hold inner nodes only: no per-level kernel has work at a leaf);
// flag[0] is set if a node lies deeper than the level lists reach
__global__ __launch_bounds__(kB) void k_u_depth(UTree u, uint32_t *depth, uint32_t *ids,
                                                unsigned int *histogram, unsigned int *flag) {
    __shared__ unsigned int s_hist[kUMaxDepth];
    for (int k = threadIdx.x; k < kUMaxDepth; k += kB) s_hist[k] = 0;
    __syncthreads();
    const uint32_t id = blockIdx.x * kB + threadIdx.x;
    if (id < u.total) {
        uint32_t d = 0;
        for (int32_t cur = (int32_t)id; cur != u.root && d < 4u * kUMaxDepth; cur = u.parent[cur]) ++d;
        if (d >= (uint32_t)kUMaxDepth) {
            atomicOr(flag, 1u);
            d = kUMaxDepth - 1;
        }
        depth[id] = d;
        if (id >= u.n) {
            ids[id] = id;
            atomicAdd(&s_hist[d], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kUMaxDepth; k += kB)
        if (s_hist[k]) atomicAdd(&histogram[k], s_hist[k]);
}

// boxes and leaf counts of the inner nodes of one level (their children, one level down, are done)
__global__ __launch_bounds__(kB) void k_u_refit_level(const uint32_t *level_ids, uint32_t count,
                                                      UTree u) {
    const uint32_t k = blockIdx.x * kB + threadIdx.x;
    if (k >= count) return;
    const uint32_t id = level_ids[k];
    if (id < u.n) return;
    const uint32_t l = (uint32_t)u.t.left[id], r = (uint32_t)u.t.right[id];
    const float4 llo = u.t.box[2 * (size_t)l], lhi = u.t.box[2 * (size_t)l + 1];
    const float4 rlo = u.t.box[2 * (size_t)r], rhi = u.t.box[2 * (size_t)r + 1];
    u.t.box[2 * (size_t)id + 0] = make_float4(fminf(llo.x, rlo.x), fminf(llo.y, rlo.y), fminf(llo.z, rlo.z), 0.f);
    u.t.box[2 * (size_t)id + 1] = make_float4(fmaxf(lhi.x, rhi.x), fmaxf(lhi.y, rhi.y), fmaxf(lhi.z, rhi.z), 0.f);
    u.t.size[id] = u.t.size[l] + u.t.size[r];
    u.height[id] = 1 + max(l < u.n ? 0u : u.height[l], r < u.n ? 0u : u.height[r]);
}

// pre-order positions of the children of one level's inner nodes (shallowest level first)
__global__ __launch_bounds__(kB) void k_u_place_level(const uint32_t *level_ids, uint32_t count,
                                                      UTree u) {
    const uint32_t k = blockIdx.x * kB + threadIdx.x;
    if (k >= count) return;
    const uint32_t id = level_ids[k];
    if (id < u.n) return;
    const uint32_t p = u.t.pos[id], b = u.t.before[id];
    const uint32_t l = (uint32_t)u.t.left[id], r = (uint32_t)u.t.right[id];
    u.t.pos[l] = p + 1;
    u.t.before[l] = b;
    u.t.pos[r] = p + 2 * u.t.size[l];
    u.t.before[r] = b + u.t.size[l];
}

__global__ __launch_bounds__(kB) void k_u_search(UTree u, const uint32_t *depth, int32_t *target,
                                                 float *gain, uint32_t *state) {
    const uint32_t x = blockIdx.x * kB + threadIdx.x;
    if (x >= u.total) return;
    target[x] = -1;
    state[x] = 0;
    if ((int32_t)x == u.root) return;
    const int32_t p = u.parent[x];
    if (p == u.root) return;                       // the root's children stay where they are
    const float4 xlo = u.t.box[2 * (size_t)x], xhi = u.t.box[2 * (size_t)x + 1];
    const float ax = box_area(xlo, xhi);
    // x ends up one level below where y is now, and so does everything below y
    const uint32_t hx = x < u.n ? 0u : u.height[x];
    const uint32_t deepest_y = max(depth[x], (uint32_t)PT_U_SOFT_DEPTH - min(hx, (uint32_t)PT_U_SOFT_DEPTH - 1)) - 1;
    // Climb from x: at each ancestor c (the pivot) look into the subtree of c's other child.
    // Taking x out frees its parent and shrinks the ancestors below the pivot to the union of
    // the siblings passed on the way (`without`); `saved` is what that gives back so far.  The
    // pivot and everything above hold x before and after.  Putting x next to y costs the area of
    // the new pair plus what the nodes between the pivot and y grow by.
    float best_gain = 0.f;
    int32_t best_y = -1;
    constexpr int kStack = 64;
    int32_t st_node[kStack];
    float st_ind[kStack];
    float saved = 0.f, here = 0.f;                 // set at the first pivot, like `without`
    float4 wlo = xlo, whi = xhi;
    float4 clo = xlo, chi = xhi;                   // box of the node the climb comes from
    int visited = 0;
    int32_t below = (int32_t)x;
    for (int32_t c = p, k = 1; c >= 0 && visited < PT_U_SEARCH_CAP; below = c, c = u.parent[c], ++k) {
        const int32_t o = u.t.left[c] == below ? u.t.right[c] : u.t.left[c];
        const float4 olo = u.t.box[2 * (size_t)o], ohi = u.t.box[2 * (size_t)o + 1];
        // the pivot's box is the union of its children's (the boxes are tight)
        clo = make_float4(fminf(clo.x, olo.x), fminf(clo.y, olo.y), fminf(clo.z, olo.z), 0.f);
        chi = make_float4(fmaxf(chi.x, ohi.x), fmaxf(chi.y, ohi.y), fmaxf(chi.z, ohi.z), 0.f);
        const float area_c = box_area(clo, chi);
        if (k == 1) saved = here = area_c;
        int sp = 0;
        {   // most pivots end here: neither next to o nor anywhere below it
            const float direct = union_area(olo, ohi, xlo, xhi);
            const float room = saved - best_gain;
            if (direct < room || (direct - box_area(olo, ohi)) + ax < room) {
                st_node[sp] = o; st_ind[sp] = 0.f; ++sp;
            }
        }
        while (sp > 0 && visited < PT_U_SEARCH_CAP) {
            --sp;
            const int32_t y = st_node[sp];
            const float ind = st_ind[sp];
            if (ind + ax >= saved - best_gain || depth[y] > deepest_y) continue;
            ++visited;
            const float4 ylo = u.t.box[2 * (size_t)y], yhi = u.t.box[2 * (size_t)y + 1];
            const float direct = union_area(ylo, yhi, xlo, xhi);
            const float total = ind + direct;
            const bool pushes_too_deep =
                depth[y] + 1 + ((uint32_t)y < u.n ? 0u : u.height[y]) > (uint32_t)PT_U_SOFT_DEPTH;
            if (saved - total > best_gain && !(k == 1 && y == o) && !pushes_too_deep) {   // its own sibling: no move
                best_gain = saved - total;
                best_y = y;
            }
            if ((uint32_t)y >= u.n) {
                const float down = ind + (direct - box_area(ylo, yhi));
                if (down + ax < saved - best_gain && sp + 2 <= kStack) {
                    // the child whose box grows less by x is looked at first
                    const int32_t l = u.t.left[y], r = u.t.right[y];
                    const float gl = union_area(u.t.box[2 * (size_t)l], u.t.box[2 * (size_t)l + 1], xlo, xhi);
                    const float gr = union_area(u.t.box[2 * (size_t)r], u.t.box[2 * (size_t)r + 1], xlo, xhi);
                    const bool left_first = gl <= gr;
                    st_node[sp] = left_first ? r : l; st_ind[sp] = down; ++sp;
                    st_node[sp] = left_first ? l : r; st_ind[sp] = down; ++sp;
                }
            }
        }
        if (k == 1) {
            wlo = olo; whi = ohi;
        } else {
            wlo = make_float4(fminf(wlo.x, olo.x), fminf(wlo.y, olo.y), fminf(wlo.z, olo.z), 0.f);
            whi = make_float4(fmaxf(whi.x, ohi.x), fmaxf(whi.y, ohi.y), fmaxf(whi.z, ohi.z), 0.f);
            const float shrunk = box_area(wlo, whi);
            // next to the pivot itself (not above the root): the new pair has the pivot's box,
            // the pivot keeps what is left without x
            if (c != u.root && saved - shrunk > best_gain &&
                depth[c] + 1 + u.height[c] <= (uint32_t)PT_U_SOFT_DEPTH) {
                best_gain = saved - shrunk;
                best_y = c;
            }
            saved += area_c - shrunk;
        }
    }
    if (best_y >= 0 && best_gain > here * 1e-4f) {
        target[x] = best_y;
        gain[x] = best_gain;
        state[x] = 1;
    }
}

// the six nodes a move of x next to y rewires; f(node) is called on each and stops the walk
// by returning false
template <class F>
__device__ __forceinline__ bool for_each_rewired(const UTree &u, int32_t x, int32_t y, F f) {
    const int32_t p = u.parent[x], g = u.parent[p], yp = u.parent[y];
    const int32_t s = u.t.left[p] == x ? u.t.right[p] : u.t.left[p];
    return f(x) && f(p) && f(s) && f(g) && f(y) && f(yp);
}

// (round, gain, x): later rounds outbid what earlier ones left behind; gain > 0, so its bits order
// like the number (the top 29 of 31 are kept), and x fits 31 bits
constexpr unsigned long long kTaken = ~0ull;
__device__ __forceinline__ unsigned long long bid_key(float gain, uint32_t x, int round = 0) {
    return ((unsigned long long)round << 60) | ((unsigned long long)(__float_as_uint(gain) >> 2) << 31) | x;
}
enum : uint32_t { kStill = 0, kBidding = 1, kLost = 2, kWon = 3, kSettled = 4 };

__global__ __launch_bounds__(kB) void k_u_lock(UTree u, const int32_t *target, const float *gain,
                                               uint32_t *state, int round,
                                               unsigned long long *lock) {
    const uint32_t x = blockIdx.x * kB + threadIdx.x;
    if (x >= u.total || state[x] != kBidding) return;
    // out, if an earlier round's winner has taken one of the six (taken marks do not change here)
    if (!for_each_rewired(u, (int32_t)x, target[x], [&](int32_t node) { return lock[node] != kTaken; })) {
        state[x] = kLost;
        return;
    }
    const unsigned long long key = bid_key(gain[x], x, round);
    for_each_rewired(u, (int32_t)x, target[x], [&](int32_t node) {
        atomicMax(&lock[node], key);
        return true;
    });
}

// a bidder that holds all six has won this round
__global__ __launch_bounds__(kB) void k_u_verify(UTree u, const int32_t *target, const float *gain,
                                                 const unsigned long long *lock, int round,
                                                 uint32_t *state) {
    const uint32_t x = blockIdx.x * kB + threadIdx.x;
    if (x >= u.total || state[x] != kBidding) return;
    const unsigned long long key = bid_key(gain[x], x, round);
    if (for_each_rewired(u, (int32_t)x, target[x], [&](int32_t node) { return lock[node] == key; }))
        state[x] = kWon;
}

// the round's winners keep their nodes for the rest of the pass
__global__ __launch_bounds__(kB) void k_u_settle(UTree u, const int32_t *target, uint32_t *state,
                                                 unsigned long long *lock) {
    const uint32_t x = blockIdx.x * kB + threadIdx.x;
    if (x >= u.total || state[x] != kWon) return;
    for_each_rewired(u, (int32_t)x, target[x], [&](int32_t node) {
        lock[node] = kTaken;
        return true;
    });
    state[x] = kSettled;
}

// Winners hold disjoint nodes, but subtrees moving into each other would still close a cycle
// (x1 goes below x2 while x2 goes below x1, or round a longer ring).  Every move of such a ring
// has another one's moving node above its target, so the one with the lowest bid of the ring
// sees a higher one there and gives way.
__global__ __launch_bounds__(kB) void k_u_accept(UTree u, const int32_t *target, const float *gain,
                                                 const uint32_t *state, uint32_t *accepted,
                                                 unsigned int *moved) {
    const uint32_t x = blockIdx.x * kB + threadIdx.x;
    if (x >= u.total) return;
    accepted[x] = 0;
    if (state[x] != kSettled) return;
    const unsigned long long key = bid_key(gain[x], x);
    for (int32_t a = u.parent[target[x]]; a >= 0; a = u.parent[a])
        if (state[a] == kSettled && bid_key(gain[a], (uint32_t)a) > key) return;
    accepted[x] = 1;
    atomicAdd(moved, 1u);
}

__global__ __launch_bounds__(kB) void k_u_apply(UTree u, const int32_t *target,
                                                const uint32_t *accepted) {
    const uint32_t xi = blockIdx.x * kB + threadIdx.x;
    if (xi >= u.total || !accepted[xi]) return;
    const int32_t x = (int32_t)xi, y = target[xi];
    const int32_t p = u.parent[x], g = u.parent[p];
    const int32_t s = u.t.left[p] == x ? u.t.right[p] : u.t.left[p];
    // take p (with x) out: the sibling moves up
    if (u.t.left[g] == p) u.t.left[g] = s; else u.t.right[g] = s;
    u.parent[s] = g;
    // hang p where y was, with y and x below it
    const int32_t yp = u.parent[y];
    if (u.t.left[yp] == y) u.t.left[yp] = p; else u.t.right[yp] = p;
    u.parent[p] = yp;
    u.t.left[p] = y;
    u.t.right[p] = x;
    u.parent[y] = p;
}

// ---- treelet restructuring --------------------------------------------------------------------
// After Karras & Aila 2013 ("Fast parallel construction of high-quality bounding volume
// hierarchies").  The other neighbourhood: where the insertion passes move one subtree at a
// time, this one takes the top of the subtree below a node — the node and, largest box first, up
// to five of its descendants, leaving seven subtrees hanging below them — and rebuilds those six
// nodes into the arrangement of the seven subtrees with the least total surface area, found by
// dynamic programming over the 127 subsets.  Leaves stay single triangles, so the cost of the
// subtrees themselves does not enter.  Nodes of one depth are roots of disjoint subtrees; the
// levels run deepest first, one launch each, so a node's treelet is rebuilt after those of the
// nodes inside it.  One wave per workgroup: the tables of the 64 lanes sit side by side in LDS,
// every lane enumerating the same subsets in the same order.
constexpr int kTreeletLeaves = 7;
constexpr int kTreeletSets = 1 << kTreeletLeaves;
constexpr int kTreeletBlock = 64;

__global__ __launch_bounds__(kTreeletBlock) void k_u_treelet(const uint32_t *level_ids, uint32_t count,
                                                             uint32_t level, UTree u,
                                                             unsigned int *changed) {
    __shared__ float s_cost[kTreeletSets][kTreeletBlock];
    __shared__ uint8_t s_part[kTreeletSets][kTreeletBlock];
    const uint32_t k = blockIdx.x * kTreeletBlock + threadIdx.x;
    const int lane = threadIdx.x;
    if (k >= count) return;
    const int32_t r = (int32_t)level_ids[k];
    if ((uint32_t)r < u.n || u.t.size[r] < 3) return;
    if (level + u.height[r] + 4 > PT_U_SOFT_DEPTH) return;          // a rearrangement can add four levels

    // the treelet: seven subtrees (fewer near the bottom) and the nodes above them
    int32_t leaf[kTreeletLeaves], inner[kTreeletLeaves - 1];
    float larea[kTreeletLeaves];
    auto area_of = [&](int32_t id) {
        return box_area(u.t.box[2 * (size_t)id], u.t.box[2 * (size_t)id + 1]);
    };
#pragma unroll
    for (int i = 0; i < kTreeletLeaves; ++i) { leaf[i] = -1; larea[i] = -1.f; }
#pragma unroll
    for (int i = 0; i < kTreeletLeaves - 1; ++i) inner[i] = -1;
    inner[0] = r;
    leaf[0] = u.t.left[r];  larea[0] = (uint32_t)leaf[0] >= u.n ? area_of(leaf[0]) : -1.f;
    leaf[1] = u.t.right[r]; larea[1] = (uint32_t)leaf[1] >= u.n ? area_of(leaf[1]) : -1.f;
    int m = 2;
    float current = area_of(r);
    for (int step = 0; step < kTreeletLeaves - 2; ++step) {
        int pick = -1;
        float best = -1.f;
#pragma unroll
        for (int i = 0; i < kTreeletLeaves; ++i)
            if (larea[i] > best) { best = larea[i]; pick = i; }
        if (pick < 0) break;                       // only triangles left
        int32_t node = -1;
#pragma unroll
        for (int i = 0; i < kTreeletLeaves; ++i) if (i == pick) node = leaf[i];
        const int32_t a = u.t.left[node], b = u.t.right[node];
        const float aa = (uint32_t)a >= u.n ? area_of(a) : -1.f;
        const float ab = (uint32_t)b >= u.n ? area_of(b) : -1.f;
        current += best;
#pragma unroll
        for (int i = 0; i < kTreeletLeaves; ++i) {
            if (i == pick) { leaf[i] = a; larea[i] = aa; }
            if (i == m) { leaf[i] = b; larea[i] = ab; }
        }
#pragma unroll
        for (int i = 0; i < kTreeletLeaves - 1; ++i) if (i == m - 1) inner[i] = node;
        ++m;
    }
    if (m < 3) return;

    float lo[kTreeletLeaves][3], hi[kTreeletLeaves][3];
    uint32_t lsize[kTreeletLeaves];
#pragma unroll
    for (int i = 0; i < kTreeletLeaves; ++i) {
        const bool have = i < m;
        const float4 l4 = have ? u.t.box[2 * (size_t)leaf[i]] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 h4 = have ? u.t.box[2 * (size_t)leaf[i] + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
        lo[i][0] = l4.x; lo[i][1] = l4.y; lo[i][2] = l4.z;
        hi[i][0] = h4.x; hi[i][1] = h4.y; hi[i][2] = h4.z;
        lsize[i] = have ? u.t.size[leaf[i]] : 0u;
    }
    auto box_of = [&](int s, float4 &blo, float4 &bhi) {
        float l0 = 3.0e38f, l1 = 3.0e38f, l2 = 3.0e38f, h0 = -3.0e38f, h1 = -3.0e38f, h2 = -3.0e38f;
#pragma unroll
        for (int i = 0; i < kTreeletLeaves; ++i) {
            const bool in = (s >> i) & 1;
            l0 = in ? fminf(l0, lo[i][0]) : l0; l1 = in ? fminf(l1, lo[i][1]) : l1; l2 = in ? fminf(l2, lo[i][2]) : l2;
            h0 = in ? fmaxf(h0, hi[i][0]) : h0; h1 = in ? fmaxf(h1, hi[i][1]) : h1; h2 = in ? fmaxf(h2, hi[i][2]) : h2;
        }
        blo = make_float4(l0, l1, l2, 0.f);
        bhi = make_float4(h0, h1, h2, 0.f);
    };

    // least total area of the nodes that arrange subset s, s ascending: every proper subset of s
    // is a smaller number.  The lowest member stays on one side, so each split is seen once.
    const int full = (1 << m) - 1;
    for (int s = 1; s <= full; ++s) {
        if ((s & (s - 1)) == 0) {
            s_cost[s][lane] = 0.f;
            continue;
        }
        float4 blo, bhi;
        box_of(s, blo, bhi);
        const int low = s & -s, rest = s ^ low;
        float best = 3.0e38f;
        int best_p = low;
        // (four splits a turn, so that their table reads are in flight together; a turn that runs
        // past the last subset repeats the split with q = 0, which changes nothing)
        for (int q = (rest - 1) & rest;;) {
            const int q1 = (q - 1) & rest, q2 = (q1 - 1) & rest, q3 = (q2 - 1) & rest;
            const int qa = q, qb = q ? q1 : 0, qc = q && q1 ? q2 : 0, qd = q && q1 && q2 ? q3 : 0;
            const float ca = s_cost[low | qa][lane] + s_cost[rest ^ qa][lane];
            const float cb = s_cost[low | qb][lane] + s_cost[rest ^ qb][lane];
            const float cc = s_cost[low | qc][lane] + s_cost[rest ^ qc][lane];
            const float cd = s_cost[low | qd][lane] + s_cost[rest ^ qd][lane];
            if (ca < best) { best = ca; best_p = low | qa; }
            if (cb < best) { best = cb; best_p = low | qb; }
            if (cc < best) { best = cc; best_p = low | qc; }
            if (cd < best) { best = cd; best_p = low | qd; }
            if (qd == 0) break;
            q = (qd - 1) & rest;
        }
        s_cost[s][lane] = box_area(blo, bhi) + best;
        s_part[s][lane] = (uint8_t)best_p;
    }
    if (!(s_cost[full][lane] < current * 0.99999f)) return;
    atomicAdd(changed, 1u);

    // hand the six nodes out again, top down
    int stack_set[kTreeletLeaves], stack_node[kTreeletLeaves];
    int sp = 0, next = 1;
    stack_set[0] = full; stack_node[0] = r; sp = 1;
    while (sp > 0) {
        --sp;
        int s = 0;
        int32_t node = -1;
#pragma unroll
        for (int i = 0; i < kTreeletLeaves; ++i) if (i == sp) { s = stack_set[i]; node = stack_node[i]; }
        const int p = s_part[s][lane];
        int32_t child[2];
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int t = side ? s ^ p : p;
            int32_t c = -1;
            if ((t & (t - 1)) == 0) {
#pragma unroll
                for (int i = 0; i < kTreeletLeaves; ++i) if (t == (1 << i)) c = leaf[i];
            } else {
#pragma unroll
                for (int i = 0; i < kTreeletLeaves - 1; ++i) if (i == next) c = inner[i];
                ++next;
                float4 blo, bhi;
                box_of(t, blo, bhi);
                u.t.box[2 * (size_t)c] = blo;
                u.t.box[2 * (size_t)c + 1] = bhi;
                uint32_t size = 0;
#pragma unroll
                for (int i = 0; i < kTreeletLeaves; ++i) size += ((t >> i) & 1) ? lsize[i] : 0u;
                u.t.size[c] = size;
#pragma unroll
                for (int i = 0; i < kTreeletLeaves; ++i) if (i == sp) { stack_set[i] = t; stack_node[i] = c; }
                ++sp;
            }
            u.parent[c] = node;
            child[side] = c;
        }
        u.t.left[node] = child[0];
        u.t.right[node] = child[1];
    }
}

inline unsigned int grid_for(uint64_t n) { return (unsigned int)((n + kB - 1) / kB); }

#define BUILD_CHECK(expr)                                 \
    do {                                                  \
        hipError_t _e = (expr);                           \
        if (_e != hipSuccess) {                           \
            res.hip_error = _e;                           \
            res.what = #expr;                             \
            return res;                                   \
        }                                                 \
    } while (0)

// n_tris triangles in n leaves; d_first (n_tris + 1 entries: first leaf of each triangle) is null
// when n == n_tris, every triangle being one leaf.
DeviceBuildResult build_once(hipStream_t st, const float *d_vtx, const int32_t *d_idx, uint32_t n_tris,
                             const uint32_t *d_first,
                             uint32_t n, uint32_t enc, DeviceBuffer &scratch,
                             const int32_t *d_mat_ids, float4 *d_nodes,
                             float4 *d_wnodes, float4 *d_wtris, float4 *d_stris,
                             int ploc_radius, int optimize_passes, int treelet_every, int max_levels) {
    DeviceBuildResult res;
    const uint32_t n_inner = n - 1;
    // scratch carve-up
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const size_t o_boxes = take(32 * (size_t)n), o_ibox = take(32 * (size_t)n_inner);
    const size_t o_k0 = take(8 * (size_t)n), o_k1 = take(8 * (size_t)n);
    const size_t o_v0 = take(4 * (size_t)n), o_v1 = take(4 * (size_t)n);
    const size_t o_left = take(4 * (size_t)n), o_right = take(4 * (size_t)n);
    const size_t o_first = take(4 * (size_t)n), o_last = take(4 * (size_t)n);
    const size_t o_ip = take(4 * (size_t)n), o_lp = take(4 * (size_t)n);
    const size_t o_depth = take(4 * (size_t)n), o_depth2 = take(4 * (size_t)n);
    const size_t o_lt = take(4 * (size_t)n);
    const size_t o_ids = take(4 * (size_t)n), o_ids2 = take(4 * (size_t)n);
    const size_t o_small = take(4 * (kMaxDepth + 16 + kUMaxDepth + 8));
    const size_t o_leaf_tri = d_first ? take(4 * (size_t)n) : 0;
    size_t sort_bytes = 0, sort2_bytes = 0;
    BUILD_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (unsigned long long *)nullptr,
                                                   (unsigned long long *)nullptr, (uint32_t *)nullptr,
                                                   (uint32_t *)nullptr, (int)n, 0, 63, st));
    BUILD_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, sort2_bytes, (uint32_t *)nullptr,
                                                   (uint32_t *)nullptr, (uint32_t *)nullptr,
                                                   (uint32_t *)nullptr, (int)n, 0, 8, st));
    size_t scan_bytes = 0;
    BUILD_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (uint32_t *)nullptr,
                                                 (uint32_t *)nullptr, (int)n, st));
    const size_t o_sort = take(std::max(std::max(sort_bytes, sort2_bytes), scan_bytes));
    // clustering and optimisation: node arrays for 2n-1 nodes (the other arrays above are reused)
    const size_t total_nodes = 2 * (size_t)n - 1;
    const bool unified = ploc_radius > 0 || optimize_passes > 0;
    const size_t o_pbox = unified ? take(32 * total_nodes) : 0;
    const size_t o_pleft = unified ? take(4 * total_nodes) : 0;
    const size_t o_pright = unified ? take(4 * total_nodes) : 0;
    const size_t o_psize = unified ? take(4 * total_nodes) : 0;
    const size_t o_ppos = unified ? take(4 * total_nodes) : 0;
    const size_t o_pbefore = unified ? take(4 * total_nodes) : 0;
    // optimisation only
    const bool opt = optimize_passes > 0;
    const size_t o_uparent = opt ? take(4 * total_nodes) : 0;
    const size_t o_uheight = opt ? take(4 * total_nodes) : 0;
    const size_t o_udepth = opt ? take(4 * total_nodes) : 0, o_udepth2 = opt ? take(4 * total_nodes) : 0;
    const size_t o_uids = opt ? take(4 * total_nodes) : 0, o_uids2 = opt ? take(4 * total_nodes) : 0;
    const size_t o_utarget = opt ? take(4 * total_nodes) : 0, o_ugain = opt ? take(4 * total_nodes) : 0;
    const size_t o_ulock = opt ? take(8 * total_nodes) : 0;
    size_t usort_bytes = 0;
    if (opt)
        BUILD_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, usort_bytes, (uint32_t *)nullptr,
                                                       (uint32_t *)nullptr, (uint32_t *)nullptr,
                                                       (uint32_t *)nullptr, (int)total_nodes, 0, 8, st));
    const size_t o_usort = opt ? take(usort_bytes) : 0;
    if (scratch.reserve(off) != PT_OK) {
        res.no_memory = true;
        return res;
    }
    char *base = (char *)scratch.ptr;
    float4 *boxes = (float4 *)(base + o_boxes), *ibox = (float4 *)(base + o_ibox);
    unsigned long long *k0 = (unsigned long long *)(base + o_k0), *k1 = (unsigned long long *)(base + o_k1);
    uint32_t *v0 = (uint32_t *)(base + o_v0), *v1 = (uint32_t *)(base + o_v1);
    int32_t *left = (int32_t *)(base + o_left), *right = (int32_t *)(base + o_right);
    uint32_t *first = (uint32_t *)(base + o_first), *last = (uint32_t *)(base + o_last);
    uint32_t *iparent = (uint32_t *)(base + o_ip), *lparent = (uint32_t *)(base + o_lp);
    uint32_t *depth = (uint32_t *)(base + o_depth), *depth2 = (uint32_t *)(base + o_depth2);
    uint32_t *lt = (uint32_t *)(base + o_lt);
    uint32_t *ids = (uint32_t *)(base + o_ids), *ids2 = (uint32_t *)(base + o_ids2);
    unsigned int *small = (unsigned int *)(base + o_small);       // [0..6] bounds, [16..] histogram
    unsigned int *histogram = small + 16;
    uint32_t *leaf_tri = d_first ? (uint32_t *)(base + o_leaf_tri) : nullptr;

    // Optimisation passes over the tree in the common numbering, then positions and output.
    auto optimise_and_emit = [&](PlocNodes t, int32_t root) -> DeviceBuildResult {
        UTree u;
        u.t = t;
        u.parent = (int32_t *)(base + o_uparent);
        u.height = (uint32_t *)(base + o_uheight);
        u.n = n;
        u.total = (uint32_t)total_nodes;
        u.root = root;
        uint32_t *udepth = (uint32_t *)(base + o_udepth), *udepth2 = (uint32_t *)(base + o_udepth2);
        uint32_t *uids = (uint32_t *)(base + o_uids), *uids2 = (uint32_t *)(base + o_uids2);
        int32_t *target = (int32_t *)(base + o_utarget);
        float *gain = (float *)(base + o_ugain);
        unsigned long long *lock = (unsigned long long *)(base + o_ulock);
        unsigned int *uhist = small + 16 + kMaxDepth, *uflags = uhist + kUMaxDepth;   // [0] too deep, [1] moves
        const unsigned int g_all = grid_for(total_nodes);
        const int32_t minus1 = -1;
        BUILD_CHECK(hipMemcpyAsync(u.parent + root, &minus1, 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_u_parents, dim3(grid_for(n_inner)), dim3(kB), 0, st, u);
        std::vector<uint32_t> begin(kUMaxDepth + 1, 0);
        int deepest = 0;
        // depths, level lists, then boxes and leaf counts bottom-up
        auto levels_and_refit = [&]() -> hipError_t {
            hipError_t e = hipMemsetAsync(uhist, 0, 4 * (kUMaxDepth + 8), st);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k_u_depth, dim3(g_all), dim3(kB), 0, st, u, udepth, uids, uhist, uflags);
            e = hipcub::DeviceRadixSort::SortPairs(base + o_usort, usort_bytes, udepth + n, udepth2, uids + n,
                                                   uids2, (int)n_inner, 0, 8, st);
            if (e != hipSuccess) return e;
            unsigned int h[kUMaxDepth + 8];
            e = hipMemcpyAsync(h, uhist, sizeof(h), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) return e;
            if (h[kUMaxDepth]) return hipErrorInvalidValue;        // deeper than the level lists
            deepest = 0;
            for (int d = 0; d < kUMaxDepth; ++d) {
                begin[d + 1] = begin[d] + h[d];
                if (h[d]) deepest = d;
            }
            if (deepest + 1 >= max_levels) return hipErrorInvalidValue;     // the leaves lie one level further down
            for (int d = deepest; d >= 0; --d) {
                const uint32_t cnt = begin[d + 1] - begin[d];
                if (cnt)
                    hipLaunchKernelGGL(k_u_refit_level, dim3(grid_for(cnt)), dim3(kB), 0, st,
                                       uids2 + begin[d], cnt, u);
            }
            return hipGetLastError();
        };
        std::string schedule;                                      // I: insertion pass, T: treelet pass
        for (int i = 1; i <= optimize_passes; ++i) {
            schedule += 'I';
            if (treelet_every > 0 && i % treelet_every == 0) schedule += 'T';
        }
        for (size_t pass = 0; pass < schedule.size(); ++pass) {
            const hipError_t e = levels_and_refit();
            if (e == hipErrorInvalidValue) break;                  // keep the tree as it is now
            BUILD_CHECK(e);
            if (schedule[pass] == 'T') {
                for (int d = deepest; d >= 0; --d) {
                    const uint32_t cnt = begin[d + 1] - begin[d];
                    if (cnt)
                        hipLaunchKernelGGL(k_u_treelet, dim3((cnt + kTreeletBlock - 1) / kTreeletBlock),
                                           dim3(kTreeletBlock), 0, st, uids2 + begin[d], cnt, (uint32_t)d, u, uflags + 2);
                }
                continue;
            }
            hipLaunchKernelGGL(k_u_search, dim3(g_all), dim3(kB), 0, st, u, udepth, target, gain, udepth2);
            BUILD_CHECK(hipMemsetAsync(lock, 0, 8 * total_nodes, st));
            for (int round = 0; round < kURounds; ++round) {
                hipLaunchKernelGGL(k_u_lock, dim3(g_all), dim3(kB), 0, st, u, target, gain, udepth2, round, lock);
                hipLaunchKernelGGL(k_u_verify, dim3(g_all), dim3(kB), 0, st, u, target, gain, lock, round, udepth2);
                hipLaunchKernelGGL(k_u_settle, dim3(g_all), dim3(kB), 0, st, u, target, udepth2, lock);
            }
            hipLaunchKernelGGL(k_u_accept, dim3(g_all), dim3(kB), 0, st, u, target, gain, udepth2, udepth,
                               uflags + 1);
            hipLaunchKernelGGL(k_u_apply, dim3(g_all), dim3(kB), 0, st, u, target, udepth);
        }
        {
            const hipError_t e = levels_and_refit();
            if (e == hipErrorInvalidValue) {
                res.hip_error = e;
                res.what = "tree deeper than the level lists";
                res.too_deep = true;
                return res;
            }
            BUILD_CHECK(e);
        }
        const uint32_t zero = 0;
        BUILD_CHECK(hipMemcpyAsync(t.pos + root, &zero, 4, hipMemcpyHostToDevice, st));
        BUILD_CHECK(hipMemcpyAsync(t.before + root, &zero, 4, hipMemcpyHostToDevice, st));
        for (int d = 0; d <= deepest; ++d) {
            const uint32_t cnt = begin[d + 1] - begin[d];
            if (cnt)
                hipLaunchKernelGGL(k_u_place_level, dim3(grid_for(cnt)), dim3(kB), 0, st, uids2 + begin[d],
                                   cnt, u);
        }
        PlocEmitArgs ea;
        ea.t = t; ea.vals = v1; ea.vtx = d_vtx; ea.idx = d_idx; ea.mat_ids = d_mat_ids;
        ea.nodes = d_nodes; ea.wnodes = d_wnodes; ea.wtris = d_wtris; ea.stris = d_stris;
        ea.leaf_tri = leaf_tri; ea.tri_first = d_first;
        ea.n = n; ea.enc = enc;
        hipLaunchKernelGGL(k_ploc_emit, dim3(g_all), dim3(kB), 0, st, ea);
        BUILD_CHECK(hipGetLastError());
        BUILD_CHECK(hipStreamSynchronize(st));
        res.max_depth = (uint32_t)deepest + 2;
        return res;
    };
    PlocNodes unified_nodes{};
    if (unified) {
        unified_nodes.box = (float4 *)(base + o_pbox);
        unified_nodes.left = (int32_t *)(base + o_pleft);
        unified_nodes.right = (int32_t *)(base + o_pright);
        unified_nodes.size = (uint32_t *)(base + o_psize);
        unified_nodes.pos = (uint32_t *)(base + o_ppos);
        unified_nodes.before = (uint32_t *)(base + o_pbefore);
    }

    // 1. boxes + centre bounds
    {
        unsigned int init[7] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u};
        BUILD_CHECK(hipMemcpyAsync(small, init, sizeof(init), hipMemcpyHostToDevice, st));
        BUILD_CHECK(hipMemsetAsync(histogram, 0, 4 * kMaxDepth, st));
        const unsigned int g = std::min<unsigned int>(grid_for(n_tris), 4096);
        hipLaunchKernelGGL(k_leaf_boxes, dim3(g), dim3(kB), 0, st, d_vtx, d_idx, n_tris, d_first, leaf_tri,
                           boxes, small);
    }
    // 2. codes, 3. sort
    hipLaunchKernelGGL(k_morton, dim3(grid_for(n)), dim3(kB), 0, st, boxes, n, small, k0, v0);
    BUILD_CHECK(hipcub::DeviceRadixSort::SortPairs(base + o_sort, sort_bytes, k0, k1, v0, v1, (int)n, 0,
                                                   63, st));
    if (ploc_radius > 0) {
        PlocNodes t = unified_nodes;
        // the radix-tree arrays double as cluster lists, flags and scans
        uint32_t *cl = (uint32_t *)left, *cl_next = (uint32_t *)right, *nearest = first;
        uint32_t *merge = last, *keep = iparent, *merge_idx = lparent, *new_pos = depth;
        hipLaunchKernelGGL(k_ploc_init, dim3(grid_for(n)), dim3(kB), 0, st, n, v1, boxes, t, cl);
        unsigned int h_small[8];
        BUILD_CHECK(hipMemcpyAsync(h_small, small, sizeof(h_small), hipMemcpyDeviceToHost, st));
        BUILD_CHECK(hipStreamSynchronize(st));
        if (h_small[6]) {
            res.bad_input = true;
            return res;
        }
        std::vector<uint32_t> round_first, round_count;
        uint32_t m = n, next_id = n;
        while (m > 1) {
            hipLaunchKernelGGL(k_ploc_nearest, dim3(grid_for(m)), dim3(kB), 0, st, m, cl, t,
                               ploc_radius, nearest);
            hipLaunchKernelGGL(k_ploc_flags, dim3(grid_for(m)), dim3(kB), 0, st, m, nearest, merge, keep);
            BUILD_CHECK(hipcub::DeviceScan::ExclusiveSum(base + o_sort, scan_bytes, merge, merge_idx, (int)m, st));
            BUILD_CHECK(hipcub::DeviceScan::ExclusiveSum(base + o_sort, scan_bytes, keep, new_pos, (int)m, st));
            hipLaunchKernelGGL(k_ploc_count, dim3(1), dim3(1), 0, st, m, merge, merge_idx, small + 8);
            hipLaunchKernelGGL(k_ploc_apply, dim3(grid_for(m)), dim3(kB), 0, st, m, cl, nearest, merge,
                               merge_idx, keep, new_pos, next_id, t, cl_next);
            unsigned int merges = 0;
            BUILD_CHECK(hipMemcpyAsync(&merges, small + 8, sizeof(merges), hipMemcpyDeviceToHost, st));
            BUILD_CHECK(hipStreamSynchronize(st));
            if (merges == 0 || merges > m / 2) {          // cannot happen (see the ranking above)
                res.hip_error = hipErrorUnknown;
                res.what = "clustering round made no progress";
                return res;
            }
            round_first.push_back(next_id);
            round_count.push_back(merges);
            next_id += merges;
            m -= merges;
            std::swap(cl, cl_next);
        }
        // root = the last node made; place everything top-down, then write the outputs
        const uint32_t root = next_id - 1;
        if (opt) return optimise_and_emit(t, (int32_t)root);
        const uint32_t zero2[2] = {0u, 0u};
        BUILD_CHECK(hipMemcpyAsync(t.pos + root, &zero2[0], 4, hipMemcpyHostToDevice, st));
        BUILD_CHECK(hipMemcpyAsync(t.before + root, &zero2[1], 4, hipMemcpyHostToDevice, st));
        for (size_t k = round_first.size(); k-- > 0;)
            hipLaunchKernelGGL(k_ploc_place, dim3(grid_for(round_count[k])), dim3(kB), 0, st,
                               round_first[k], round_count[k], t, 0u, small + 9);
        PlocEmitArgs ea;
        ea.t = t; ea.vals = v1; ea.vtx = d_vtx; ea.idx = d_idx; ea.mat_ids = d_mat_ids;
        ea.nodes = d_nodes; ea.wnodes = d_wnodes; ea.wtris = d_wtris; ea.stris = d_stris;
        ea.leaf_tri = leaf_tri; ea.tri_first = d_first;
        ea.n = n; ea.enc = enc;
        hipLaunchKernelGGL(k_ploc_emit, dim3(grid_for(2 * (uint64_t)n - 1)), dim3(kB), 0, st, ea);
        BUILD_CHECK(hipGetLastError());
        BUILD_CHECK(hipStreamSynchronize(st));
        res.max_depth = (uint32_t)round_first.size() + 1;     // a path gains at most one node per round
        return res;
    }
    // 4. tree, depths
    hipLaunchKernelGGL(k_radix_tree, dim3(grid_for(n_inner)), dim3(kB), 0, st, k1, (int)n, left, right,
                       first, last, iparent, lparent);
    hipLaunchKernelGGL(k_depth, dim3(grid_for(n_inner)), dim3(kB), 0, st, iparent, (int)n_inner, depth,
                       lt, ids, histogram);
    BUILD_CHECK(hipcub::DeviceRadixSort::SortPairs(base + o_sort, sort2_bytes, depth, depth2, ids, ids2,
                                                   (int)n_inner, 0, 8, st));
    unsigned int h_small[16 + kMaxDepth];
    BUILD_CHECK(hipMemcpyAsync(h_small, small, sizeof(h_small), hipMemcpyDeviceToHost, st));
    BUILD_CHECK(hipStreamSynchronize(st));
    if (h_small[6]) {
        res.bad_input = true;
        return res;
    }
    // 5. boxes level by level, deepest first
    std::vector<uint32_t> begin(kMaxDepth + 1, 0);
    for (int d = 0; d < kMaxDepth; ++d) begin[d + 1] = begin[d] + h_small[16 + d];
    int max_depth = 0;
    for (int d = kMaxDepth - 1; d >= 0; --d) {
        const uint32_t cnt = begin[d + 1] - begin[d];
        if (!cnt) continue;
        max_depth = std::max(max_depth, d);
        hipLaunchKernelGGL(k_refit_level, dim3(grid_for(cnt)), dim3(kB), 0, st, ids2 + begin[d], cnt,
                           left, right, v1, boxes, ibox);
    }
    if (opt) {
        hipLaunchKernelGGL(k_u_from_radix, dim3(grid_for(total_nodes)), dim3(kB), 0, st, n, left, right, v1,
                           boxes, ibox, unified_nodes);
        return optimise_and_emit(unified_nodes, (int32_t)n);      // inner node 0, the radix tree's root
    }
    // 6. output
    EmitArgs a;
    a.left = left; a.right = right; a.first = first; a.last = last; a.lt = lt;
    a.leaf_parent = lparent; a.vals = v1; a.boxes = boxes; a.ibox = ibox;
    a.vtx = d_vtx; a.idx = d_idx;
    a.nodes = d_nodes; a.wnodes = d_wnodes; a.wtris = d_wtris;
    a.stris = d_stris; a.mat_ids = d_mat_ids;
    a.leaf_tri = leaf_tri; a.tri_first = d_first;
    a.n = n; a.enc = enc;
    hipLaunchKernelGGL(k_emit_inner, dim3(grid_for(n_inner)), dim3(kB), 0, st, a);
    hipLaunchKernelGGL(k_emit_leaf, dim3(grid_for(n)), dim3(kB), 0, st, a);
    BUILD_CHECK(hipGetLastError());
    BUILD_CHECK(hipStreamSynchronize(st));
    res.max_depth = (uint32_t)max_depth + 2;      // inner levels + the leaf level
    return res;
}

} // namespace

DeviceBuildResult plan_triangle_splits(hipStream_t st, const float *d_vtx, const int32_t *d_idx,
                                       uint32_t n, float ratio, int max_pieces, float budget,
                                       DeviceBuffer &first, uint32_t &n_leaves) {
    DeviceBuildResult res;
    n_leaves = n;
    max_pieces = std::min(std::max(max_pieces, 1), 256);
    if (!(ratio > 0.f) || max_pieces < 2 || !(budget > 0.f)) return res;
    size_t scan_bytes = 0;
    BUILD_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (uint32_t *)nullptr,
                                                 (uint32_t *)nullptr, (int)(n + 1), st));
    DeviceBuffer tmp;
    const size_t counts_bytes = (4 * ((size_t)n + 1) + 255) / 256 * 256;
    if (tmp.reserve(counts_bytes + 256 + scan_bytes) != PT_OK || first.reserve(4 * ((size_t)n + 1)) != PT_OK) {
        res.no_memory = true;
        return res;
    }
    uint32_t *counts = (uint32_t *)tmp.ptr;
    unsigned long long *extra = (unsigned long long *)((char *)tmp.ptr + counts_bytes);
    void *scan_tmp = (char *)tmp.ptr + counts_bytes + 256;
    // All the leaves the criterion asks for, or, if that is more than `budget` extra leaves per
    // triangle, every triangle's surplus scaled down alike (counted in integers: deterministic;
    // the scaled counts meet the budget to within their rounding).
    float scale = 1.f;
    unsigned long long h_extra = 0;
    for (int round = 0; round < 2; ++round) {
        BUILD_CHECK(hipMemsetAsync(extra, 0, 8, st));
        hipLaunchKernelGGL(k_split_count, dim3(grid_for((uint64_t)n + 1)), dim3(kB), 0, st, d_vtx, d_idx, n, ratio,
                           max_pieces, scale, counts, extra);
        BUILD_CHECK(hipMemcpyAsync(&h_extra, extra, 8, hipMemcpyDeviceToHost, st));
        BUILD_CHECK(hipStreamSynchronize(st));
        const double allowed = (double)budget * (double)n;
        if (round == 1 || (double)h_extra <= allowed) break;
        scale = (float)(allowed / (double)h_extra);
    }
    if (h_extra == 0) {
        tmp.release();
        return res;
    }
    BUILD_CHECK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, counts, (uint32_t *)first.ptr,
                                                 (int)(n + 1), st));
    BUILD_CHECK(hipStreamSynchronize(st));
    tmp.release();
    n_leaves = n + (uint32_t)h_extra;
    return res;
}

DeviceBuildResult build_scene_on_device(hipStream_t st, const float *d_vtx, const int32_t *d_idx,
                                        uint32_t n_tris, uint32_t n_leaves, const uint32_t *d_first,
                                        uint32_t enc, DeviceBuffer &scratch,
                                        const int32_t *d_mat_ids, float4 *d_nodes,
                                        float4 *d_wnodes, float4 *d_wtris, float4 *d_stris,
                                        int ploc_radius, int optimize_passes, int treelet_every,
                                        int max_levels) {
    max_levels = std::min(std::max(max_levels, 2), kUMaxDepth);
    if (n_leaves == n_tris) d_first = nullptr;
    DeviceBuildResult res = build_once(st, d_vtx, d_idx, n_tris, d_first, n_leaves, enc, scratch, d_mat_ids,
                                       d_nodes, d_wnodes, d_wtris, d_stris, ploc_radius, optimize_passes,
                                       treelet_every, max_levels);
    // The optimisation passes can, on contrived input, make a tree deeper than their level
    // lists reach; the radix tree cannot (its depth is bounded by the key length), so it
    // is the answer then.  The inputs are untouched and every output is written again.
    if (res.too_deep && (ploc_radius > 0 || optimize_passes > 0)) {
        (void)hipGetLastError();
        res = build_once(st, d_vtx, d_idx, n_tris, d_first, n_leaves, enc, scratch, d_mat_ids, d_nodes,
                         d_wnodes, d_wtris, d_stris, 0, 0, 0, max_levels);
    }
    return res;
}

} // namespace mipt
