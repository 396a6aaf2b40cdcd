// Per-ray BVH traversal for gfx950 (wave64), shared by the stand-alone trace kernels and
// the path tracer's extend kernel.
//
//  traverse_skiplink : IntersectSceneClosest exactly as written
//                      (shaders/RadeonRays/bvh.glslh:417-456): stackless, fixed order,
//                      on the 32-byte skip-link nodes the reference uploads.
//  traverse_ordered  : the production traversal.  Same boxes, same triangle test, same
//                      arithmetic, but nearer child first with a per-lane stack in LDS,
//                      on 64-byte two-child nodes and 48-byte triangle records.
//  closest_hit       : traverse_ordered, plus traverse_skiplink for the rare ray whose
//                      answer could depend on the visiting order.
//
// Why the ordered traversal returns what the fixed-order shader returns.  A triangle can
// become the shader's answer only if its own leaf box passes the part of the slab test
// that does not depend on the running nearest distance (an ancestor's box contains the
// leaf box and rounding is monotonic, so the ancestors then pass too) and its triangle
// test passes.  That candidate set does not depend on visiting order; only the choice
// among candidates whose distances are within a few ulp of each other does (the shader
// accepts an equal-distance triangle visited later, bvh.glslh:268-273).  The ordered
// traversal keeps its culling distance a factor 1+PT_TIE_WINDOW above its nearest hit,
// so it sees every candidate in that window; a ray with two or more (a few per million on
// ordinary scenes), or with a hit within the window of its own reach, is traversed again in the
// shader's fixed order.
// A ray with one gets that one from either order.
#pragma once

#include <limits.h>

#include "kernels.h"
#include "pt_device_math.h"

namespace mipt {

// Minimum resident waves per SIMD the persistent traversal kernels are compiled for (second
// argument of __launch_bounds__): bounds their VGPR allocation.
#ifndef PT_EXTEND_MIN_WAVES
#define PT_EXTEND_MIN_WAVES 1
#endif
// The closest-hit loop over both levels of a two-level scene needs 106 registers left to itself,
// i.e. four waves per SIMD; held to five (96 registers, six spilled outside the loops) it traces
// 4096 placements of a 65 K-triangle mesh at 3.08 instead of 2.85 Grays/s.
#ifndef PT_TWO_LEVEL_MIN_WAVES
#define PT_TWO_LEVEL_MIN_WAVES 5
#endif

constexpr int kTraceBlock = PT_TRACE_BLOCK;
constexpr int kTraceStack = PT_LDS_STACK_ENTRIES;
constexpr int32_t kSentinel = INT_MIN;

// The dynamic LDS of a persistent kernel's workgroup (kPersistentLdsBytes at launch).
extern __shared__ __align__(16) unsigned char g_persistent_lds[];
__device__ __forceinline__ int32_t *persistent_stack() {      // this lane's column
    return reinterpret_cast<int32_t *>(g_persistent_lds) + threadIdx.x;
}
// The node copy is read through a pointer that says "LDS" in its type: given a generic pointer the
// compiler folds the two sources of a node (LDS or global memory) into one flat load, which
// goes through the vector memory pipeline like the global loads it was meant to replace.
typedef float NativeFloat4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) NativeFloat4 LdsFloat4;
__device__ __forceinline__ float4 lds_load4(const LdsFloat4 *p) {
    const NativeFloat4 v = *p;
    return make_float4(v.x, v.y, v.z, v.w);
}
typedef int NativeInt2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) NativeInt2 LdsInt2;
__device__ __forceinline__ int2 lds_load_int2(const LdsFloat4 *p) {
    const NativeInt2 v = *reinterpret_cast<const LdsInt2 *>(p);
    return make_int2(v.x, v.y);
}
__device__ __forceinline__ void lds_store4(LdsFloat4 *p, const float4 v) {
    NativeFloat4 w;
    w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
    *p = w;
}
__device__ __forceinline__ LdsFloat4 *persistent_cache(int stack_cap) {   // behind the stacks
    return (LdsFloat4 *)(g_persistent_lds + (size_t)(stack_cap + 1) * kTraceBlock * 4);
}

__device__ __forceinline__ void store_hit(pt_hit *hits, uint64_t i, int32_t shapeid, int32_t primid,
                                          float b0, float b1, float b2, float t) {
    int4 *p = reinterpret_cast<int4 *>(hits + i);
    p[0] = make_int4(shapeid, primid, 667, 0);   // bvh.glslh:429-430
    reinterpret_cast<float4 *>(p)[1] = make_float4(b0, b1, b2, t);
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// IntersectSceneClosest, shaders/RadeonRays/bvh.glslh:417-456, for one ray.
template <bool MT, bool COUNT>
__device__ __forceinline__ void traverse_skiplink(const DeviceScene &s, const float4 ro,
                                                  const float4 rd, int32_t &shapeid,
                                                  int32_t &primid, float &b0, float &b1, float &b2,
                                                  float &tbest, unsigned long long &nb,
                                                  unsigned long long &nt,
                                                  float limit = INFINITY) {
    const RayPre r = precompute_ray(ro.x, ro.y, ro.z, rd.x, rd.y, rd.z);
    b0 = 0.f; b1 = 0.f; b2 = 0.f; tbest = ro.w;           // :423
    shapeid = -1; primid = -1;                              // :424-425
    int32_t idx = 0;
    // limit: a distance beyond which the caller knows the answer does not lie (the nearest hit
    // of the ordered traversal, widened by the tie window and a margin for rounding).  The walk starts culling there
    // instead of at r.o.w: hits beyond it would be replaced before the walk ends, so skipping
    // them and the boxes only they need leaves the result as it is.
    if (!MT && limit < tbest) {
        float cull = limit;
        while (idx != -1) {
            const float4 lo = s.nodes[2 * (int64_t)idx + 0];
            const float4 hi = s.nodes[2 * (int64_t)idx + 1];
            if (COUNT) ++nb;
            float t0;
            const int32_t link = decode_link(hi.w, s.link_enc);
            const int32_t payload = decode_link(lo.w, s.link_enc);
            if (!intersect_box(r, lo.x, hi.x, lo.y, hi.y, lo.z, hi.z, cull, t0)) {
                idx = link;
            } else if (payload == -1) {
                ++idx;
            } else {
                if (COUNT) ++nt;
                const int64_t start = payload;
                const float *p0 = s.vtx + 3 * (int64_t)s.idx[3 * start + 0];
                const float *p1 = s.vtx + 3 * (int64_t)s.idx[3 * start + 1];
                const float *p2 = s.vtx + 3 * (int64_t)s.idx[3 * start + 2];
                float h0, h1, h2, ht;
                if (intersect_triangle_watertight(r, p0[0], p0[1], p0[2], p1[0], p1[1], p1[2], p2[0],
                                                  p2[1], p2[2], cull, h0, h1, h2, ht)) {
                    b0 = h0; b1 = h1; b2 = h2; tbest = ht; cull = ht;
                    primid = payload;
                    shapeid = 0;
                }
                idx = link;
            }
        }
        return;
    }
    while (idx != -1) {
        const float4 lo = s.nodes[2 * (int64_t)idx + 0];
        const float4 hi = s.nodes[2 * (int64_t)idx + 1];
        if (COUNT) ++nb;
        float t0;
        const int32_t link = decode_link(hi.w, s.link_enc);
        if (intersect_box(r, lo.x, hi.x, lo.y, hi.y, lo.z, hi.z, tbest, t0)) {
            const int32_t payload = decode_link(lo.w, s.link_enc);
            if (payload != -1) {                            // LEAFNODE, :154
                if (COUNT) ++nt;
                const int64_t start = payload;              // STARTIDX, :153
                const int32_t i0 = s.idx[3 * start + 0], i1 = s.idx[3 * start + 1],
                              i2 = s.idx[3 * start + 2];
                const float *p0 = s.vtx + 3 * (int64_t)i0, *p1 = s.vtx + 3 * (int64_t)i1,
                            *p2 = s.vtx + 3 * (int64_t)i2;
                const float v0x = p0[0], v0y = p0[1], v0z = p0[2];
                const float v1x = p1[0], v1y = p1[1], v1z = p1[2];
                const float v2x = p2[0], v2y = p2[1], v2z = p2[2];
                if (MT) {
                    float m1, m2, temp;
                    moller_trumbore(r.ox, r.oy, r.oz, rd.x, rd.y, rd.z, v0x, v0y, v0z, v1x, v1y,
                                    v1z, v2x, v2y, v2z, m1, m2, temp);
                    temp = (0.0f < temp) ? temp : 0.0f;     // :191, max() as GLSL defines it (+0 for -0)
                    if (!(m1 < -PT_MT_EPS || m1 > 1.f + PT_MT_EPS || m2 < -PT_MT_EPS ||
                          m1 + m2 > 1.f + PT_MT_EPS || temp < -PT_MT_EPS ||
                          temp > tbest + PT_MT_EPS)) {       // :193
                        b0 = m1; b1 = m2; b2 = 0.f; tbest = temp;   // :199
                        primid = payload; shapeid = 0;
                    }
                } else {
                    float h0, h1, h2, ht;
                    if (intersect_triangle_watertight(r, v0x, v0y, v0z, v1x, v1y, v1z, v2x, v2y,
                                                      v2z, tbest, h0, h1, h2, ht)) {
                        b0 = h0; b1 = h1; b2 = h2; tbest = ht;
                        primid = payload;                   // :344
                        shapeid = 0;                        // :345
                    }
                }
                idx = link;                                 // :443
            } else {
                ++idx;                                      // :448
            }
        } else {
            idx = link;                                     // :453
        }
    }
}

// IntersectSceneClosest restricted to the leaves of two triangles, pa and pb, taken in the
// loop's own order (ascending node index).  This is what the whole loop returns for a ray whose
// only candidates are those two: every other triangle is either rejected by its own test or
// lies farther than both by more than rounding can reverse, so it neither ends up as the answer
// nor changes what the two are compared against; and the loop reaches a leaf exactly when the
// leaf's own box passes the slab test at the current distance (the boxes above it contain it and
// rounding is monotonic).  Needs s.tri_leaf, or s.tri_leaf_first / s.tri_leaf_list where
// triangles have several leaves.
//
// What the candidates' leaves alone cannot show: the loop tests a leaf's box against whatever it
// has accepted by then, which may be a third triangle farther than both candidates.  Such a
// hit lies beyond the tie window of the answer, so it cannot keep the loop out of a box that the
// ray enters no later than the window's end -- but it can keep it out of one entered later.
// There are such boxes: a flat box whose entry distance was rounded up past the hit (a hit close
// to the origin on large coordinates), and, where a triangle lies in several leaves, every leaf
// of it that the ray passes through behind the hit point (the triangle test is the whole
// triangle's, so the loop accepts the hit through any leaf it gets into).  A candidate leaf that
// the ray enters at all, but beyond the window of the final distance, therefore makes the replay
// say "undecided" (primid -1), and the caller walks the tree as the reference does.
__device__ __forceinline__ void replay_candidates(const DeviceScene &s, const float4 ro,
                                                  const float4 rd, int32_t pa, int32_t pb,
                                                  int32_t &shapeid, int32_t &primid, float &b0,
                                                  float &b1, float &b2, float &tbest) {
    const RayPre r = precompute_ray(ro.x, ro.y, ro.z, rd.x, rd.y, rd.z);
    b0 = 0.f; b1 = 0.f; b2 = 0.f; tbest = ro.w;
    shapeid = -1; primid = -1;
    float latest_entry = 0.f;            // over the leaves the ray enters within its reach
    auto leaf = [&](int32_t node, int32_t tri) {
        const float4 lo = s.nodes[2 * (int64_t)node + 0];
        const float4 hi = s.nodes[2 * (int64_t)node + 1];
        // entered within the ray's reach at all?  then also within the current distance iff the
        // entry lies before it (the slab test's far side does not depend on the distance)
        float e0;
        if (!intersect_box(r, lo.x, hi.x, lo.y, hi.y, lo.z, hi.z, ro.w, e0)) return;
        latest_entry = fmaxf(latest_entry, e0);
        if (!(tbest >= e0)) return;
        const int32_t i0 = s.idx[3 * (int64_t)tri + 0], i1 = s.idx[3 * (int64_t)tri + 1],
                      i2 = s.idx[3 * (int64_t)tri + 2];
        const float *p0 = s.vtx + 3 * (int64_t)i0, *p1 = s.vtx + 3 * (int64_t)i1,
                    *p2 = s.vtx + 3 * (int64_t)i2;
        const float v0x = p0[0], v0y = p0[1], v0z = p0[2];
        const float v1x = p1[0], v1y = p1[1], v1z = p1[2];
        const float v2x = p2[0], v2y = p2[1], v2z = p2[2];
        float h0, h1, h2, ht;
        if (intersect_triangle_watertight(r, v0x, v0y, v0z, v1x, v1y, v1z, v2x, v2y, v2z, tbest,
                                          h0, h1, h2, ht)) {
            b0 = h0; b1 = h1; b2 = h2; tbest = ht;
            primid = tri;
            shapeid = 0;
        }
    };
    if (s.tri_leaf) {
        int32_t na = s.tri_leaf[pa], nb = s.tri_leaf[pb];
        if (nb < na) {
            const int32_t tp = pa, tn = na;
            pa = pb; na = nb;
            pb = tp; nb = tn;
        }
        leaf(na, pa);
        leaf(nb, pb);
    } else {
        uint32_t ia = s.tri_leaf_first[pa], ib = s.tri_leaf_first[pb];
        const uint32_t ea = s.tri_leaf_first[pa + 1], eb = s.tri_leaf_first[pb + 1];
        while (ia < ea || ib < eb) {                 // the two ascending lists, merged
            const int32_t na = ia < ea ? s.tri_leaf_list[ia] : INT_MAX;
            const int32_t nb = ib < eb ? s.tri_leaf_list[ib] : INT_MAX;
            const bool take_a = na < nb;
            ia += take_a ? 1u : 0u;
            ib += take_a ? 0u : 1u;
            leaf(take_a ? na : nb, take_a ? pa : pb);
        }
    }
    if (!(latest_entry <= tbest + tbest * PT_TIE_WINDOW)) {
        primid = -1;
        shapeid = -1;
    }
}

// Records of the re-laid-out scene.  BIG = false: by 32-bit byte offset from a uniform base (one
// scalar base + one vector offset per load), for scenes whose arrays stay below 4 GiB (2^26
// two-child nodes, 89 M triangle records); BIG = true: 64-bit address arithmetic per lane, for
// scenes beyond that.  The persistent kernels are compiled both ways and the scene says which
// to launch (DeviceScene::big); the one-thread-per-ray forms always take the 64-bit form.
template <bool BIG = true>
__device__ __forceinline__ const float4 *wide_node(const DeviceScene &s, int32_t node) {
    if (BIG) return s.wnodes + 4 * (size_t)(uint32_t)node;
    return reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s.wnodes) +
                                            ((uint32_t)node << 6));
}
template <bool BIG = true>
__device__ __forceinline__ const float4 *wide_tri(const DeviceScene &s, int32_t leaf) {
    if (BIG) return s.wtris + (size_t)(uint32_t)(~leaf);
    return reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s.wtris) +
                                            ((uint32_t)(~leaf) << 4));
}

// Records of the four-child form, addressed like the two-child ones (BIG: 64-bit arithmetic).
template <bool BIG = false>
__device__ __forceinline__ const float4 *quad_node(const DeviceScene &s, int32_t node) {
    if (BIG) return s.qnodes + 4 * (size_t)(uint32_t)node;
    return reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s.qnodes) +
                                            ((uint32_t)node << 6));
}
template <bool BIG = false>
__device__ __forceinline__ const float4 *quad_tri(const DeviceScene &s, int32_t leaf) {
    if (BIG) return s.qtris + (size_t)(uint32_t)(~leaf);
    return reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s.qtris) +
                                            ((uint32_t)(~leaf) << 4));
}

// Ordered closest-hit traversal of one ray.  `stack` points at this lane's column of the
// workgroup's LDS stack (entry e at stack[e * kTraceBlock]).  Sets `tie` when the result
// could depend on the visiting order; the caller then re-resolves the ray with
// traverse_skiplink.
template <bool COUNT>
__device__ __forceinline__ void traverse_ordered(const DeviceScene &s, const float4 ro,
                                                 const float4 rd, int32_t *stack,
                                                 int32_t &primid, float &b0, float &b1, float &b2,
                                                 float &tbest, bool &tie, unsigned long long &nb,
                                                 unsigned long long &nt, int32_t root = 0) {
    const RayPre r = precompute_ray(ro.x, ro.y, ro.z, rd.x, rd.y, rd.z);
    const float tmax = ro.w;
    b0 = 0.f; b1 = 0.f; b2 = 0.f;
    tbest = tmax;                // nearest accepted distance (tmax while there is none)
    // culling distance: tbest widened by the tie window, and so is the reach while there is no
    // hit (see the note on hits near the reach at the triangle step of ordered_advance)
    float bound = tmax + tmax * PT_TIE_WINDOW;
    float second = INFINITY;     // nearest distance among the other candidates seen
    primid = -1;
    tie = false;

    int sp = 0;
    int32_t cur = root;          // >= 0: two-child node, < 0: a triangle record (see wide_tri)
    // Non-finite rays and zero directions follow whatever the shader's arithmetic makes of
    // them (a NaN distance is "accepted" once and then culls everything): that depends on
    // the visiting order, so they are left to the fixed-order traversal.
    {
        const float chk = ((ro.x + ro.y) + (ro.z + rd.x)) + (rd.y + rd.z);
        if (!(fabsf(chk) < INFINITY) || !(tmax == tmax) ||
            (rd.x == 0.f && rd.y == 0.f && rd.z == 0.f)) {
            tie = true;
            cur = kSentinel;
        }
    }
    while (cur != kSentinel) {
        while (cur >= 0) {
            const float4 *np = wide_node(s, cur);
            const float4 a = np[0], b = np[1], c = np[2];
            const int4 ch = reinterpret_cast<const int4 *>(np)[3];
            if (COUNT) nb += 2;
            float ta, tb;
            const bool ha = intersect_box(r, a.x, a.y, a.z, a.w, c.x, c.y, bound, ta);
            const bool hb = intersect_box(r, b.x, b.y, b.z, b.w, c.z, c.w, bound, tb);
            if (ha && hb) {
                const bool swap = tb < ta;
                const int32_t near_c = swap ? ch.y : ch.x;
                const int32_t far_c = swap ? ch.x : ch.y;
                if (sp < kTraceStack) {
                    stack[sp * kTraceBlock] = far_c;
                    ++sp;
                    cur = near_c;
                } else {           // out of stack: leave this ray to the stackless traversal
                    tie = true;
                    cur = kSentinel;
                }
            } else if (ha || hb) {
                cur = ha ? ch.x : ch.y;
            } else if (sp > 0) {
                --sp;
                cur = stack[sp * kTraceBlock];
            } else {
                cur = kSentinel;
            }
        }
        while (cur < 0 && cur != kSentinel) {
            const float4 *tp = wide_tri(s, cur);
            const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
            if (COUNT) ++nt;
            float h0, h1, h2, ht;
            if (intersect_triangle_watertight(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w,
                                              q2.x, bound, h0, h1, h2, ht)) {
                if (!(ht == ht)) {   // NaN distance out of finite inputs: as above
                    tie = true;
                    cur = kSentinel;
                    break;
                }
                if (primid < 0 || ht < tbest) {
                    if (primid >= 0) second = fminf(second, tbest);
                    tbest = ht; b0 = h0; b1 = h1; b2 = h2;
                    primid = __float_as_int(q2.y);
                    bound = tbest + tbest * PT_TIE_WINDOW;
                    tie |= bound >= tmax;
                } else if (__float_as_int(q2.y) != primid) {
                    second = fminf(second, ht);
                }   // else: another leaf of the same triangle (a tree built with spatial splits)
            }
            if (sp > 0) {
                --sp;
                cur = stack[sp * kTraceBlock];
            } else {
                cur = kSentinel;
            }
        }
    }
    if (primid >= 0 && second <= tbest + tbest * PT_TIE_WINDOW) tie = true;
}

// ---- the same traversal as a resumable state machine (persistent-wave kernels) ---------
// Where a lane's stack goes on once its LDS entries are used up: `entries` more slots in global
// memory, slot e of all the grid's lanes side by side.  entries == 0: there is none.
struct StackSpill {
    int32_t *lane;
    uint32_t stride;
    int entries;
};
__device__ __forceinline__ StackSpill lane_spill(int32_t *base, int entries) {
    StackSpill x;
    x.stride = gridDim.x * kTraceBlock;
    x.lane = base ? base + (blockIdx.x * kTraceBlock + threadIdx.x) : nullptr;
    x.entries = base ? entries : 0;
    return x;
}
constexpr int kSpillBlock = 8;      // entries that move between LDS and global memory at a time

struct OrderedState {
    RayPre r;
    float tmax, tbest, bound, second, b0, b1, b2;
    int32_t primid, cur;     // cur == kSentinel: no traversal in progress
    // The other candidates seen within the tie window of the nearest hit: -1 none, >= 0 exactly
    // one (its triangle id), -2 several.
    int32_t other;
    int sp;
    bool tie;
    // the tie flag came from a value that is not finite (the ray, or a hit distance): what the
    // fixed order makes of such a ray is not bounded by any hit found here
    bool wild;
    // Four-child walk only (see coarse_ray_terms): per axis, 1/d made safe for the coarse slab
    // test (0 where the direction component is zero or next to it) and the allowance by which
    // that test widens each slab (infinite on such an axis).
    float qax, qay, qaz, qmx, qmy, qmz;
};

// The coarse boxes of a four-child node need no more than to contain the exact ones, so their
// slab test need not be the reference's expression: distance of a decoded plane
//   fl(fl(fl(q * scale + org) - o) * inv)          three instructions per plane, as in the shader
// becomes
//   fl(q * A + B),  A = fl(scale * inv),  B = fl(fl(org - o) * inv) -/+ m
// one instruction per plane and fifteen per node.  The two differ by rounding, by less than
// 9 u |inv| (C + |o|) with u = 2^-24 and C the largest coordinate of the scene (each of the
// eight roundings involved acts on a term bounded by the node's extent, the plane's coordinate or
// its distance from the origin), so with m = 2^-20 |inv| (C + |o|) the near planes come out no
// later and the far planes no earlier than in the reference's form: a ray that passes the
// reference's test on a child's exact box passes this one.  Relative to a slab's width m is
// about 1e-6 C / (box size): nothing is let in that matters.  An axis along which the ray does
// not move (1 / d infinite, or too large for the products to stay finite) gets A = 0 and
// B = -/+ infinity: no constraint from that axis, which is again only wider.
__device__ __forceinline__ void coarse_ray_terms(OrderedState &st, float cmax) {
    const float kAllow = 9.5367431640625e-07f;      // 2^-20
    const float ax = fabsf(st.r.ix), ay = fabsf(st.r.iy), az = fabsf(st.r.iz);
    const bool okx = ax < 1.0e30f, oky = ay < 1.0e30f, okz = az < 1.0e30f;
    st.qax = okx ? st.r.ix : 0.f;
    st.qay = oky ? st.r.iy : 0.f;
    st.qaz = okz ? st.r.iz : 0.f;
    st.qmx = okx ? kAllow * (ax * (cmax + fabsf(st.r.ox))) : INFINITY;
    st.qmy = oky ? kAllow * (ay * (cmax + fabsf(st.r.oy))) : INFINITY;
    st.qmz = okz ? kAllow * (az * (cmax + fabsf(st.r.oz))) : INFINITY;
}
// Largest coordinate of the scene, a little generous (the decoded boxes may stick out of the
// root box by a rounding step).
__device__ __forceinline__ float scene_max_coordinate(const DeviceScene &s) {
    const float4 lo = s.nodes[0], hi = s.nodes[1];
    const float c = fmaxf(fmaxf(fmaxf(fabsf(lo.x), fabsf(hi.x)), fmaxf(fabsf(lo.y), fabsf(hi.y))),
                          fmaxf(fabsf(lo.z), fabsf(hi.z)));
    return c * 1.001f;
}

__device__ __forceinline__ void ordered_begin(OrderedState &st, const float4 ro, const float4 rd,
                                              int32_t *stack, int32_t root = 0, int base = 0) {
    st.r = precompute_ray(ro.x, ro.y, ro.z, rd.x, rd.y, rd.z);
    st.tmax = ro.w;
    st.b0 = 0.f; st.b1 = 0.f; st.b2 = 0.f;
    st.tbest = st.tmax;
    st.bound = st.tmax + st.tmax * PT_TIE_WINDOW;     // see the triangle step of ordered_advance
    st.second = INFINITY;
    st.primid = -1;
    st.other = -1;
    st.tie = false;
    st.wild = false;
    stack[base * kTraceBlock] = kSentinel;   // bottom of the stack: popping it ends the ray
    st.sp = base + 1;
    st.cur = root;
    const float chk = ((ro.x + ro.y) + (ro.z + rd.x)) + (rd.y + rd.z);
    if (!(fabsf(chk) < INFINITY) || !(st.tmax == st.tmax) ||
        (rd.x == 0.f && rd.y == 0.f && rd.z == 0.f)) {
        st.tie = true;
        st.wild = true;
        st.cur = kSentinel;
    }
}

// Start of an occlusion query (IntersectSceneAny, shaders/RadeonRays/bvh.glslh:378-414) in
// the same state machine: 1/d as at :380, the culling distance stays at r.o.w, and the shear
// slots of the ray, which the Moller-Trumbore test has no use for, carry the direction.
// exact: leave the ray to the skip-link walk (see occluded_wide for when that is needed).
__device__ __forceinline__ void any_begin(OrderedState &st, const float4 ro, const float4 rd,
                                          int32_t *stack, bool exact, int32_t root = 0,
                                          int base = 0) {
    st.r.ox = ro.x; st.r.oy = ro.y; st.r.oz = ro.z;
    st.r.ix = 1.0f / rd.x; st.r.iy = 1.0f / rd.y; st.r.iz = 1.0f / rd.z;
    st.r.sx = rd.x; st.r.sy = rd.y; st.r.sz = rd.z;
    st.r.axis = 2;
    st.tmax = ro.w;
    st.bound = ro.w;
    st.tbest = ro.w;
    st.second = INFINITY;
    st.b0 = 0.f; st.b1 = 0.f; st.b2 = 0.f;
    st.primid = -1;                // >= 0 once an occluder has been found
    st.other = -1;
    st.tie = false;
    st.wild = false;
    stack[base * kTraceBlock] = kSentinel;
    st.sp = base + 1;
    st.cur = root;
    const float chk = ((ro.x + ro.y) + (ro.z + rd.x)) + (rd.y + rd.z);
    if (exact || !(fabsf(chk) < INFINITY) || !(ro.w == ro.w) ||
        (rd.x == 0.f && rd.y == 0.f && rd.z == 0.f)) {
        st.tie = true;
        st.cur = kSentinel;
    }
}

// One step at a four-child node whose record (f0, f1, f2: grid and child bytes, ch: the children)
// the caller has read: tests the four decoded boxes, takes the nearest hit child and pushes the
// others, farthest first; with no room for them the lane stops with the node itself on top of its
// stack, as in the two-child step.
template <bool COUNT>
__device__ __forceinline__ void quad_node_step(OrderedState &st, int32_t *stack, int stack_cap,
                                               const float4 f0, const float4 f1, const float4 f2,
                                               const int4 ch, unsigned long long &nb) {
    if (COUNT) nb += 4;
    const uint32_t qlx = __float_as_uint(f1.z), qly = __float_as_uint(f1.w),
                   qlz = __float_as_uint(f2.x), qhx = __float_as_uint(f2.y),
                   qhy = __float_as_uint(f2.z), qhz = __float_as_uint(f2.w);
    float key[4];
    int32_t ref[4] = {ch.x, ch.y, ch.z, ch.w};
    int nh = 0;
    // The slab test of bvh.glslh:165-177 on the decoded boxes, with the min / max of
    // the two planes of an axis taken by the sign of 1/d instead (one select per axis
    // for the four children together): the same numbers wherever no product is a NaN
    // (a decoded lo is never above its hi, and rounding is monotonic), and where one is
    // (a ray inside a box plane with no extent along that axis) the plane drops out
    // of the max3 / min3 -- which can only let the ray into a coarse box the
    // reference's arithmetic would have kept it out of.  Not so for a leaf's exact
    // box, whose test is the reference's as written (triangle step).
    const bool sgx = __float_as_int(st.r.ix) < 0, sgy = __float_as_int(st.r.iy) < 0,
               sgz = __float_as_int(st.r.iz) < 0;
    const uint32_t qnx = sgx ? qhx : qlx, qfx = sgx ? qlx : qhx;
    const uint32_t qny = sgy ? qhy : qly, qfy = sgy ? qly : qhy;
    const uint32_t qnz = sgz ? qhz : qlz, qfz = sgz ? qlz : qhz;
#ifndef PT_QUAD_REFERENCE_FORM
    // see coarse_ray_terms
    const float cax = f0.w * st.qax, cay = f1.x * st.qay, caz = f1.y * st.qaz;
    const float cbx = (f0.x - st.r.ox) * st.qax, cby = (f0.y - st.r.oy) * st.qay,
                cbz = (f0.z - st.r.oz) * st.qaz;
    const float bnx = cbx - st.qmx, bny = cby - st.qmy, bnz = cbz - st.qmz;
    const float bfx = cbx + st.qmx, bfy = cby + st.qmy, bfz = cbz + st.qmz;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float nx = __builtin_fmaf((float)((qnx >> (8 * k)) & 0xffu), cax, bnx);
        const float ny = __builtin_fmaf((float)((qny >> (8 * k)) & 0xffu), cay, bny);
        const float nz = __builtin_fmaf((float)((qnz >> (8 * k)) & 0xffu), caz, bnz);
        const float fx = __builtin_fmaf((float)((qfx >> (8 * k)) & 0xffu), cax, bfx);
        const float fy = __builtin_fmaf((float)((qfy >> (8 * k)) & 0xffu), cay, bfy);
        const float fz = __builtin_fmaf((float)((qfz >> (8 * k)) & 0xffu), caz, bfz);
        const float t1 = fminf(fminf(fx, fminf(fy, fz)), st.bound);
        const float t0 = fmaxf(fmaxf(nx, fmaxf(ny, nz)), 0.0f);
        const bool h = (t1 >= t0) && ref[k] != kNoChild;
        key[k] = h ? fminf(t0, 3.0e38f) : INFINITY;
        nh += h ? 1 : 0;
    }
#else
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float pnx = __builtin_fmaf((float)((qnx >> (8 * k)) & 0xffu), f0.w, f0.x);
        const float pny = __builtin_fmaf((float)((qny >> (8 * k)) & 0xffu), f1.x, f0.y);
        const float pnz = __builtin_fmaf((float)((qnz >> (8 * k)) & 0xffu), f1.y, f0.z);
        const float pfx = __builtin_fmaf((float)((qfx >> (8 * k)) & 0xffu), f0.w, f0.x);
        const float pfy = __builtin_fmaf((float)((qfy >> (8 * k)) & 0xffu), f1.x, f0.y);
        const float pfz = __builtin_fmaf((float)((qfz >> (8 * k)) & 0xffu), f1.y, f0.z);
        const float nx = (pnx - st.r.ox) * st.r.ix, fx = (pfx - st.r.ox) * st.r.ix;
        const float ny = (pny - st.r.oy) * st.r.iy, fy = (pfy - st.r.oy) * st.r.iy;
        const float nz = (pnz - st.r.oz) * st.r.iz, fz = (pfz - st.r.oz) * st.r.iz;
        const float t1 = fminf(fminf(fx, fminf(fy, fz)), st.bound);
        const float t0 = fmaxf(fmaxf(nx, fmaxf(ny, nz)), 0.0f);
        const bool h = (t1 >= t0) && ref[k] != kNoChild;
        key[k] = h ? fminf(t0, 3.0e38f) : INFINITY;
        nh += h ? 1 : 0;
    }
#endif
#ifdef PT_EXP_QUAD_EXTRA
    // experiment: PT_EXP_QUAD_EXTRA more vector instructions per four-child step (what
    // the launch pays -- or would save -- per instruction of this step)
    {
        float acc = key[0];
#pragma unroll
        for (int e = 0; e < PT_EXP_QUAD_EXTRA; ++e)
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(f0.x), "v"(f0.y));
        if (acc == 12345.678f) key[0] = acc;
    }
#endif
    // children in order of entry distance (misses last): five compare-exchanges.  (Singling
    // out only the nearest child and pushing the others in child order saves a fifth of
    // the step's instructions and was measured 3-5 % slower on the 28 M-triangle scenes:
    // the order in which the pushed siblings are taken up matters.)
#define PT_CE(a, b)                                                        \
    {                                                                      \
        const bool sw = key[b] < key[a];                                   \
        const float ka = sw ? key[b] : key[a], kb = sw ? key[a] : key[b];  \
        const int32_t ra = sw ? ref[b] : ref[a], rb = sw ? ref[a] : ref[b];\
        key[a] = ka; key[b] = kb; ref[a] = ra; ref[b] = rb;                \
    }
    PT_CE(0, 1) PT_CE(2, 3) PT_CE(0, 2) PT_CE(1, 3) PT_CE(1, 2)
#undef PT_CE
    // No room for the farther children: as in the two-child step, the lane stops with
    // the node itself on top of its stack.
    const bool full = nh > 1 && st.sp + nh - 1 > stack_cap;
    int32_t next = ref[0];
    if (full) {
        stack[st.sp * kTraceBlock] = st.cur;
        ++st.sp;
        next = kSentinel;
    } else if (nh > 1) {
        if (nh > 3) { stack[st.sp * kTraceBlock] = ref[3]; ++st.sp; }
        if (nh > 2) { stack[st.sp * kTraceBlock] = ref[2]; ++st.sp; }
        stack[st.sp * kTraceBlock] = ref[1];
        ++st.sp;
    } else if (nh == 0) {
        --st.sp;
        next = stack[st.sp * kTraceBlock];
    }
    st.cur = next;
}

// Advances the traversing lanes of the wave until at most `min_active` of them are still
// traversing (0: run the wave's rays to completion).  Each round the wave votes on one step:
// the lanes standing at a triangle take a triangle step once `leaf_lanes` of them have
// gathered (or no lane stands at a node), otherwise the lanes standing at a node take a node
// step.  Lanes of the other kind wait, so neither step runs for a handful of stragglers.
// Stack slot 0 holds kSentinel (ordered_begin puts it there), so a pop needs no emptiness
// test: popping the bottom entry ends the ray.  Both steps are written with selects; the
// only divergent regions are the stack accesses and the work inside the triangle test.
// ANY: occlusion query -- the triangle step is the Moller-Trumbore test of
// shaders/RadeonRays/bvh.glslh:289-310 and the first hit ends the ray.
// QUAD: the node step reads a four-child node (DeviceScene::qnodes; closest hits and occlusion
// queries alike): child boxes from 8-bit
// offsets on the node's grid, never inside the exact boxes, so a ray that reaches a leaf in the
// two-child tree reaches it here too; the slab arithmetic on them is the reference's (rounding
// is monotonic in the box).  The leaf's exact box, the one the reference tests, comes with the
// triangle record and is tested at the triangle step, against the nearest hit at that time.
template <bool COUNT, bool ANY = false, bool CACHE = false, bool BIG = false, bool QUAD = false>
__device__ __forceinline__ void ordered_advance(const DeviceScene &s, OrderedState &st,
                                                int32_t *stack, int min_active, int leaf_lanes,
                                                unsigned long long &nb, unsigned long long &nt,
                                                int stack_cap = kTraceStack,
                                                const LdsFloat4 *cache = nullptr,
                                                unsigned long long *nl = nullptr,
                                                int *budget = nullptr) {
    for (;;) {
        // budget (wave-uniform, optional): rounds of this loop -- a run of node steps and the
        // triangle step after it -- the wave may still make; see the tail of persistent_traverse
        if (budget) {
            if (*budget <= 0) return;
            --*budget;
        }
        // Node steps run in a loop of their own: only the position and the stack pointer
        // change in it, so the rest of the lane's state stays put in its registers.
        // The vote: 0 node step, 1 triangle step, 2 done (the same value in every lane).
        auto vote = [&](bool &at_node, bool &at_leaf) {
            at_node = st.cur >= 0;
            at_leaf = (st.cur < 0) & (st.cur != kSentinel);
            const int n_node = (int)__popcll(__ballot(at_node));
            const int n_leaf = (int)__popcll(__ballot(at_leaf));
            return (n_node + n_leaf <= min_active) ? 2 : ((n_node != 0 && n_leaf < leaf_lanes) ? 0 : 1);
        };
        bool at_node, at_leaf;
        int mode = vote(at_node, at_leaf);
        while (mode == 0) {
            if constexpr (QUAD) {
              if (at_node) {
                float4 f0, f1, f2;
                int4 ch;
                if (CACHE && st.cur >= kLdsNodeBase) {
                    const LdsFloat4 *np = cache + ((st.cur - kLdsNodeBase) << 2);
                    f0 = lds_load4(np); f1 = lds_load4(np + 1); f2 = lds_load4(np + 2);
                    const float4 cw = lds_load4(np + 3);
                    ch = make_int4(__float_as_int(cw.x), __float_as_int(cw.y), __float_as_int(cw.z),
                                   __float_as_int(cw.w));
                    if (COUNT && nl) *nl += 4;
                } else {
                    const float4 *np = quad_node<BIG>(s, st.cur);
                    f0 = np[0]; f1 = np[1]; f2 = np[2];
                    ch = reinterpret_cast<const int4 *>(np)[3];
                }
                quad_node_step<COUNT>(st, stack, stack_cap, f0, f1, f2, ch, nb);
              }
            } else
            if (at_node) {
                float4 a, b, c;
                int2 ch;                                    // the two children: 8 of the last 16 bytes
                if (CACHE && st.cur >= kLdsNodeBase) {      // a node of the top of the tree, in LDS
                    const LdsFloat4 *np = cache + ((st.cur - kLdsNodeBase) << 2);
                    a = lds_load4(np); b = lds_load4(np + 1); c = lds_load4(np + 2);
                    ch = lds_load_int2(np + 3);
#ifndef PT_WAVE_ITERS
                    if (COUNT && nl) *nl += 2;
#endif
                } else {
                    const float4 *np = wide_node<BIG>(s, st.cur);
                    a = np[0]; b = np[1]; c = np[2];
                    ch = reinterpret_cast<const int2 *>(np + 3)[0];
                }
#ifdef PT_WAVE_ITERS
                // instrumented build: node steps per wave (x128, in place of the slab tests), and of
                // those the steps in which every lane of the step stands at one and the same node
                // of the LDS copy (x128, in place of the slab tests on LDS nodes)
                if (COUNT && (__ffsll(__ballot(1)) - 1) == (int)(threadIdx.x & 63)) nb += 128;
                {
                    const int32_t first_node = __builtin_amdgcn_readfirstlane(st.cur);
                    const bool uniform = __ballot(st.cur == first_node) == __ballot(1) &&
                                         first_node >= kLdsNodeBase;
                    if (COUNT && nl && uniform && (__ffsll(__ballot(1)) - 1) == (int)(threadIdx.x & 63)) *nl += 128;
                }
#else
                if (COUNT) nb += 2;
#endif
                float ta, tb;
                const bool ha = intersect_box(st.r, a.x, a.y, a.z, a.w, c.x, c.y, st.bound, ta);
                const bool hb = intersect_box(st.r, b.x, b.y, b.z, b.w, c.z, c.w, st.bound, tb);
                const bool take_b = hb && (!ha || (tb < ta));   // the nearer hit child is c1
                const bool both = ha && hb, none = !ha && !hb;
                // No room for the farther child: the lane stops here with the node itself on top
                // of its stack (the last slot is kept for that), and whoever drives the loop
                // makes room and repeats the step (stack_service) or gives the ray up.
                const bool full = both && (st.sp >= stack_cap);
                int32_t next = take_b ? ch.y : ch.x;
                const int32_t far_c = take_b ? ch.x : ch.y;
                if (both) {
                    stack[st.sp * kTraceBlock] = full ? st.cur : far_c;
                    ++st.sp;
                }
                if (none) {
                    --st.sp;
                    next = stack[st.sp * kTraceBlock];
                }
                st.cur = full ? kSentinel : next;
            }
            mode = vote(at_node, at_leaf);
        }
        if (mode == 2) return;
        if (at_leaf) {
            const float4 *tp = QUAD ? quad_tri<BIG>(s, st.cur) : wide_tri<BIG>(s, st.cur);
            const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
            bool in_leaf_box = true;
            if constexpr (QUAD) {
                // the leaf's own box, the one test of the reference's that the coarse boxes on the
                // way here stand in for
                const float4 q3 = tp[3];
                float t;
                in_leaf_box = intersect_box(st.r, q2.z, q3.y, q2.w, q3.z, q3.x, q3.w, st.bound, t);
                if (COUNT) ++nb;
            }
#ifdef PT_WAVE_ITERS
            if (COUNT && (__ffsll(__ballot(1)) - 1) == (int)(threadIdx.x & 63)) nt += 64;
#else
            if (COUNT) nt += in_leaf_box ? 1 : 0;
#endif
            if (ANY) {
                float m1, m2, temp;
                moller_trumbore(st.r.ox, st.r.oy, st.r.oz, st.r.sx, st.r.sy, st.r.sz, q0.x, q0.y,
                                q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, m1, m2, temp);
                // (four-child nodes: the reference gets to this triangle only through its leaf's own
                // box, which the coarse boxes on the way here stand in for)
                const bool hit = !((m1 < -PT_MT_EPS) | (m1 > 1.f + PT_MT_EPS) | (m2 < -PT_MT_EPS) |
                                   (m1 + m2 > 1.f + PT_MT_EPS) | (temp < -PT_MT_EPS) |
                                   (temp > st.tmax + PT_MT_EPS)) &               // :301
                                 in_leaf_box;
                st.primid = hit ? 0 : st.primid;
                --st.sp;
                const int32_t popped = stack[st.sp * kTraceBlock];
                st.cur = hit ? kSentinel : popped;
                continue;
            }
            float h0 = 0.f, h1 = 0.f, h2 = 0.f, ht = 0.f;
#ifdef PT_EXP_TRI_TWICE
            // experiment: the triangle test a second time on inputs the compiler cannot tell are
            // the same -- what the launch pays per triangle step's worth of vector instructions
            float zero;
            asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
            float g0, g1, g2, gt;
            const bool hit_again = intersect_triangle_watertight(st.r, q0.x + zero, q0.y, q0.z, q0.w, q1.x, q1.y,
                                                                 q1.z, q1.w, q2.x, st.bound, g0, g1, g2, gt);
#endif
            bool hit = intersect_triangle_watertight(st.r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y,
                                                     q1.z, q1.w, q2.x, st.bound, h0, h1, h2, ht) &
                       in_leaf_box;
#ifdef PT_EXP_TRI_TWICE
            hit &= (hit_again == hit) | !in_leaf_box | true;
            if (hit_again && !(gt == ht)) ht = gt;          // never taken for finite distances
#endif
            const bool bad = hit & !(ht == ht);
            const bool had = st.primid >= 0;
            const bool closer = hit & !bad & (!had | (ht < st.tbest));
            // another leaf of the triangle already held (a tree built with spatial splits) is no
            // competing candidate
            const bool farther = hit & !bad & !closer & (__float_as_int(q2.y) != st.primid);
            const float s_old = fminf(st.second, st.tbest), s_new = fminf(st.second, ht);
            st.second = (closer & had) ? s_old : (farther ? s_new : st.second);
            // Hits near the ray's reach: the test compares products, so a distance an ulp beyond
            // the reach can be accepted -- and the shader then culls at that distance, which lets
            // in a triangle it had turned down at the reach itself, or would have, had it come
            // first.  The reach is therefore widened like every culling distance (ordered_begin),
            // and a ray with a hit within the window of its reach is left to the fixed order.
            const float nbound = ht + ht * PT_TIE_WINDOW;
            // Who else is within the window of the nearest hit: a hit that is not nearer joins;
            // a nearer one keeps the old nearest (and whoever was with it) if that is within the
            // new window, and stands alone if not (the others were no nearer than the old one).
            const int32_t joined = st.other == -1 ? (closer ? st.primid : __float_as_int(q2.y)) : -2;
            st.other = farther ? joined
                               : ((closer & had) ? (st.tbest <= nbound ? joined : -1) : st.other);
            st.tbest = closer ? ht : st.tbest;
            st.b0 = closer ? h0 : st.b0;
            st.b1 = closer ? h1 : st.b1;
            st.b2 = closer ? h2 : st.b2;
            st.primid = closer ? __float_as_int(q2.y) : st.primid;
            st.bound = closer ? nbound : st.bound;
            st.tie |= bad | (hit & !bad & (nbound >= st.tmax));
            st.wild |= bad;
            --st.sp;
            const int32_t popped = stack[st.sp * kTraceBlock];
            st.cur = bad ? kSentinel : popped;
        }
    }
}

// For a lane whose run has stopped (st.cur == kSentinel): true if it goes on.  A run that
// stopped for lack of stack room (entries left above the sentinel at `base`, no hit that ended
// an occlusion query, no tie flag) moves its oldest kSpillBlock entries to the lane's global
// slots, closes the gap and repeats the step of the node it stopped at; a run whose LDS entries
// have run out while blocks are in global memory brings the newest block back.  `spilled`
// counts the lane's blocks out there (0 at the start of a run).  Without room out there the
// run gets its tie flag, i.e. the ray is left to the fixed-order walk.
template <bool ANY>
__device__ __forceinline__ bool stack_service(OrderedState &st, int32_t *stack, const StackSpill &x,
                                              int base, int &spilled) {
    if (st.tie || (ANY && st.primid >= 0)) return false;
    if (st.sp > base) {
        // Above `base` lie the run's pending entries and, on top, the node that found no room.  A
        // block can only go out if a whole block of pending entries is there: with a small LDS
        // stack and a four-child step (which stops up to two entries short of the end), or deep
        // inside the stack of a two-level walk (`base` near the end), there may be fewer -- the
        // ray is then left to the fixed-order walk, as when the global slots are used up.
        if (st.sp - base - 2 < kSpillBlock || (spilled + 1) * kSpillBlock > x.entries) {
            st.tie = true;
            return false;
        }
        for (int e = 0; e < kSpillBlock; ++e)
            x.lane[(size_t)(spilled * kSpillBlock + e) * x.stride] = stack[(base + 1 + e) * kTraceBlock];
        for (int i = base + 1 + kSpillBlock; i < st.sp; ++i)
            stack[(i - kSpillBlock) * kTraceBlock] = stack[i * kTraceBlock];
        st.sp -= kSpillBlock + 1;
        st.cur = stack[st.sp * kTraceBlock];                      // the node that found no room
        ++spilled;
        return true;
    }
    if (spilled > 0) {
        --spilled;
        for (int e = 0; e < kSpillBlock; ++e)
            stack[(base + 1 + e) * kTraceBlock] = x.lane[(size_t)(spilled * kSpillBlock + e) * x.stride];
        st.sp = base + kSpillBlock;
        st.cur = stack[st.sp * kTraceBlock];
        return true;
    }
    return false;
}

// Call once for a lane whose traversal has ended; true if the ray must be re-resolved.
__device__ __forceinline__ bool ordered_is_tie(const OrderedState &st) {
    return st.tie || (st.primid >= 0 && st.second <= st.tbest + st.tbest * PT_TIE_WINDOW);
}

// Closest hit of one ray with the reference's result.  Returns whether the fixed-order
// traversal had to be run (or would have been, when fixup is off).
template <bool COUNT>
__device__ __forceinline__ bool closest_hit(const DeviceScene &s, const float4 ro, const float4 rd,
                                            int32_t *stack, bool fixup, int32_t &shapeid,
                                            int32_t &primid, float &b0, float &b1, float &b2,
                                            float &t, unsigned long long &nb,
                                            unsigned long long &nt, int32_t root = 0) {
    bool tie;
    traverse_ordered<COUNT>(s, ro, rd, stack, primid, b0, b1, b2, t, tie, nb, nt, root);
    shapeid = primid >= 0 ? 0 : -1;
    if (tie && fixup) {
        unsigned long long xb = 0, xt = 0;
        traverse_skiplink<false, false>(s, ro, rd, shapeid, primid, b0, b1, b2, t, xb, xt);
    }
    return tie;
}

// IntersectSceneAny, shaders/RadeonRays/bvh.glslh:378-414, for one ray: any triangle whose
// Moller-Trumbore test (IntersectTriangleP, :289-310) passes within r.o.w.
__device__ __forceinline__ bool occluded_skiplink(const DeviceScene &s, const float4 ro,
                                                  const float4 rd) {
    RayPre r;
    r.ox = ro.x; r.oy = ro.y; r.oz = ro.z;
    r.ix = 1.0f / rd.x; r.iy = 1.0f / rd.y; r.iz = 1.0f / rd.z;   // :380
    r.sx = r.sy = r.sz = 0.f;
    r.axis = 2;
    int32_t idx = 0;
    while (idx != -1) {
        const float4 lo = s.nodes[2 * (int64_t)idx + 0];
        const float4 hi = s.nodes[2 * (int64_t)idx + 1];
        const int32_t link = decode_link(hi.w, s.link_enc);
        float t0;
        if (intersect_box(r, lo.x, hi.x, lo.y, hi.y, lo.z, hi.z, ro.w, t0)) {
            const int32_t payload = decode_link(lo.w, s.link_enc);
            if (payload != -1) {
                const int64_t start = payload;
                const int32_t i0 = s.idx[3 * start + 0], i1 = s.idx[3 * start + 1],
                              i2 = s.idx[3 * start + 2];
                const float *p0 = s.vtx + 3 * (int64_t)i0, *p1 = s.vtx + 3 * (int64_t)i1,
                            *p2 = s.vtx + 3 * (int64_t)i2;
                float m1, m2, temp;
                moller_trumbore(r.ox, r.oy, r.oz, rd.x, rd.y, rd.z, p0[0], p0[1], p0[2], p1[0],
                                p1[1], p1[2], p2[0], p2[1], p2[2], m1, m2, temp);
                if (!(m1 < -PT_MT_EPS || m1 > 1.f + PT_MT_EPS || m2 < -PT_MT_EPS ||
                      m1 + m2 > 1.f + PT_MT_EPS || temp < -PT_MT_EPS ||
                      temp > ro.w + PT_MT_EPS))              // :301
                    return true;
                idx = link;
            } else {
                ++idx;
            }
        } else {
            idx = link;
        }
    }
    return false;
}

// The same predicate over the two-child nodes, nearer child first.  Nothing shrinks during
// an any-hit query, so which triangles get tested does not depend on the visiting order: a
// triangle is tested exactly when every box on the way to it passes the slab test against
// r.o.w, in the skip-link order and in this one alike (a child's box lies inside its
// parent's, and the slab arithmetic is monotonic in the plane positions).  Rays the
// argument does not cover (non-finite, zero direction) and rays that overflow the stack
// take the skip-link walk.  Not for single-triangle scenes: their filler record is only
// safe under the watertight test.
__device__ __forceinline__ bool occluded_wide(const DeviceScene &s, const float4 ro,
                                              const float4 rd, int32_t *stack, int32_t root = 0) {
    const float chk = ((ro.x + ro.y) + (ro.z + rd.x)) + (rd.y + rd.z);
    if (!(fabsf(chk) < INFINITY) || !(ro.w == ro.w) ||
        (rd.x == 0.f && rd.y == 0.f && rd.z == 0.f))
        return occluded_skiplink(s, ro, rd);
    RayPre r;
    r.ox = ro.x; r.oy = ro.y; r.oz = ro.z;
    r.ix = 1.0f / rd.x; r.iy = 1.0f / rd.y; r.iz = 1.0f / rd.z;   // :380
    r.sx = r.sy = r.sz = 0.f;
    r.axis = 2;
    int sp = 0;
    int32_t cur = root;
    for (;;) {
        if (cur >= 0) {
            const float4 *np = wide_node(s, cur);
            const float4 a = np[0], b = np[1], c = np[2];
            const int4 ch = reinterpret_cast<const int4 *>(np)[3];
            float ta, tb;
            const bool ha = intersect_box(r, a.x, a.y, a.z, a.w, c.x, c.y, ro.w, ta);
            const bool hb = intersect_box(r, b.x, b.y, b.z, b.w, c.z, c.w, ro.w, tb);
            if (ha && hb) {
                if (sp >= kTraceStack) return occluded_skiplink(s, ro, rd);
                const bool swap = tb < ta;
                stack[sp * kTraceBlock] = swap ? ch.x : ch.y;
                ++sp;
                cur = swap ? ch.y : ch.x;
                continue;
            }
            if (ha || hb) {
                cur = ha ? ch.x : ch.y;
                continue;
            }
        } else {
            const float4 *tp = wide_tri(s, cur);
            const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
            float m1, m2, temp;
            moller_trumbore(r.ox, r.oy, r.oz, rd.x, rd.y, rd.z, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y,
                            q1.z, q1.w, q2.x, m1, m2, temp);
            if (!(m1 < -PT_MT_EPS || m1 > 1.f + PT_MT_EPS || m2 < -PT_MT_EPS ||
                  m1 + m2 > 1.f + PT_MT_EPS || temp < -PT_MT_EPS ||
                  temp > ro.w + PT_MT_EPS))                  // :301
                return true;
        }
        if (sp == 0) return false;
        --sp;
        cur = stack[sp * kTraceBlock];
    }
}

// ---- two-level scenes ------------------------------------------------------------------
// The top level is walked in the reference's fixed skip-link order on the world-space ray
// (same loop as IntersectSceneClosest, shaders/RadeonRays/bvh.glslh:431-455); a top-level leaf
// names an instance (src/bvh/bvh_builder.h:31), whose mesh is then searched with the ray taken
// to object space and the running nearest distance as its limit.  The ray parameter is the
// same in both spaces (the direction is transformed, not renormalised).  Meshes are searched
// with closest_hit / occluded_wide, i.e. with the reference's per-mesh result, so the whole
// query equals the fixed-order walk of both levels.

// One instance's mesh as a scene of its own; `ray` goes to its object space.  The two-child
// nodes and triangle records of all meshes share one index space (a mesh's child indices are
// stored scene-wide), so only the arrays of the fixed-order walk are rebased; root receives
// the mesh's first two-child node.
__device__ __forceinline__ DeviceScene enter_instance(const DeviceScene &s, int32_t instance,
                                                      float4 &ro, float4 &rd, int32_t &root) {
    const float4 *ip = reinterpret_cast<const float4 *>(s.instances + instance);
    const float4 r0 = ip[0], r1 = ip[1], r2 = ip[2];
    const uint4 o0 = reinterpret_cast<const uint4 *>(ip)[3];
    const uint4 o1 = reinterpret_cast<const uint4 *>(ip)[4];
    DeviceScene m = s;
    m.nodes = s.nodes + 2 * (size_t)o0.x;
    root = (int32_t)o0.y;
    m.vtx = s.vtx + 3 * (size_t)o0.w;
    m.idx = s.idx + 3 * (size_t)o1.x;
    m.n_tri = o1.y;
    const float ox = ro.x, oy = ro.y, oz = ro.z, dx = rd.x, dy = rd.y, dz = rd.z;
    ro.x = ((r0.x * ox + r0.y * oy) + r0.z * oz) + r0.w;
    ro.y = ((r1.x * ox + r1.y * oy) + r1.z * oz) + r1.w;
    ro.z = ((r2.x * ox + r2.y * oy) + r2.z * oz) + r2.w;
    rd.x = (r0.x * dx + r0.y * dy) + r0.z * dz;
    rd.y = (r1.x * dx + r1.y * dy) + r1.z * dz;
    rd.z = (r2.x * dx + r2.y * dy) + r2.z * dz;
    return m;
}

__device__ __forceinline__ RayPre top_level_ray(const float4 ro, const float4 rd) {
    RayPre r;
    r.ox = ro.x; r.oy = ro.y; r.oz = ro.z;
    r.ix = 1.0f / rd.x; r.iy = 1.0f / rd.y; r.iz = 1.0f / rd.z;   // bvh.glslh:421
    r.sx = r.sy = r.sz = 0.f;
    r.axis = 2;
    return r;
}

// Closest hit: shapeid = instance, primid = triangle of the instance's mesh.  exact: meshes
// by traverse_skiplink instead of closest_hit.  Returns whether any mesh search was a tie.
template <bool COUNT>
__device__ __forceinline__ bool closest_hit_two_level(const DeviceScene &s, const float4 ro,
                                                      const float4 rd, int32_t *stack, bool fixup,
                                                      bool exact, int32_t &shapeid,
                                                      int32_t &primid, float &b0, float &b1,
                                                      float &b2, float &t, unsigned long long &nb,
                                                      unsigned long long &nt) {
    const RayPre r = top_level_ray(ro, rd);
    b0 = 0.f; b1 = 0.f; b2 = 0.f; t = ro.w;
    shapeid = -1; primid = -1;
    bool tie = false;
    int32_t idx = 0;
    while (idx != -1) {
        const float4 lo = s.tlas[2 * (int64_t)idx + 0];
        const float4 hi = s.tlas[2 * (int64_t)idx + 1];
        if (COUNT) ++nb;
        const int32_t link = decode_link(hi.w, s.link_enc);
        float t0;
        if (intersect_box(r, lo.x, hi.x, lo.y, hi.y, lo.z, hi.z, t, t0)) {
            const int32_t payload = decode_link(lo.w, s.link_enc);
            if (payload != -1) {
                float4 oo = ro, od = rd;
                int32_t root;
                const DeviceScene m = enter_instance(s, payload, oo, od, root);
                oo.w = t;
                int32_t sid, pid;
                float c0, c1, c2, ct;
                if (exact)
                    traverse_skiplink<false, COUNT>(m, oo, od, sid, pid, c0, c1, c2, ct, nb, nt);
                else
                    tie |= closest_hit<COUNT>(m, oo, od, stack, fixup, sid, pid, c0, c1, c2, ct, nb,
                                              nt, root);
                if (pid >= 0) {
                    shapeid = payload; primid = pid;
                    b0 = c0; b1 = c1; b2 = c2; t = ct;
                }
                idx = link;
            } else {
                ++idx;
            }
        } else {
            idx = link;
        }
    }
    return tie;
}

// Occlusion (IntersectSceneAny, bvh.glslh:378-414, over both levels).
__device__ __forceinline__ bool occluded_two_level(const DeviceScene &s, const float4 ro,
                                                   const float4 rd, int32_t *stack, bool exact) {
    const RayPre r = top_level_ray(ro, rd);
    int32_t idx = 0;
    while (idx != -1) {
        const float4 lo = s.tlas[2 * (int64_t)idx + 0];
        const float4 hi = s.tlas[2 * (int64_t)idx + 1];
        const int32_t link = decode_link(hi.w, s.link_enc);
        float t0;
        if (intersect_box(r, lo.x, hi.x, lo.y, hi.y, lo.z, hi.z, ro.w, t0)) {
            const int32_t payload = decode_link(lo.w, s.link_enc);
            if (payload != -1) {
                float4 oo = ro, od = rd;
                int32_t root;
                const DeviceScene m = enter_instance(s, payload, oo, od, root);
                const bool occ = (exact || m.n_tri < 2) ? occluded_skiplink(m, oo, od)
                                                        : occluded_wide(m, oo, od, stack, root);
                if (occ) return true;
                idx = link;
            } else {
                ++idx;
            }
        } else {
            idx = link;
        }
    }
    return false;
}

__device__ __forceinline__ void count_ties(bool tie, unsigned int *tie_count) {
    const unsigned long long m = __ballot(tie);
    if (m && (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1)
        atomicAdd(tie_count, (unsigned int)__popcll(m));
}

// Where the persistent closest-hit loop puts a finished ray.
struct HitSink {
    pt_hit *hits;
    __device__ __forceinline__ void operator()(uint32_t ray, int32_t shapeid, int32_t primid,
                                               float b0, float b1, float b2, float t) const {
        store_hit(hits, ray, shapeid, primid, b0, b1, b2, t);
    }
};

// The workgroup's copy of the top of the tree (DeviceScene::top_cache, or q_top_cache for the
// four-child nodes; for a two-level scene the top of the top-level tree): the first
// pp.cache_nodes nodes go to LDS, and the children beyond the workgroup's share get their global
// names back.  All threads of the workgroup call it; true if anything was copied.
template <bool QUAD>
__device__ __forceinline__ bool copy_top_to_lds(const DeviceScene &s, const PersistentParams &pp,
                                                LdsFloat4 *cache) {
    const float4 *const top = QUAD ? s.q_top_cache : s.top_cache;
    const int32_t *const top_global = QUAD ? s.q_top_global : s.top_global;
    const uint32_t n_cached = min((uint32_t)pp.cache_nodes, QUAD ? s.n_q_top_cache : s.n_top_cache);
    for (uint32_t i = threadIdx.x; i < 4u * n_cached; i += kTraceBlock) {
        float4 v = top[i];
        if ((i & 3u) == 3u) {
            int32_t c0 = __float_as_int(v.x), c1 = __float_as_int(v.y);
            if (c0 >= kLdsNodeBase + (int32_t)n_cached) c0 = top_global[c0 - kLdsNodeBase];
            if (c1 >= kLdsNodeBase + (int32_t)n_cached) c1 = top_global[c1 - kLdsNodeBase];
            v.x = __int_as_float(c0);
            v.y = __int_as_float(c1);
            if constexpr (QUAD) {
                int32_t c2 = __float_as_int(v.z), c3 = __float_as_int(v.w);
                if (c2 >= kLdsNodeBase + (int32_t)n_cached) c2 = top_global[c2 - kLdsNodeBase];
                if (c3 >= kLdsNodeBase + (int32_t)n_cached) c3 = top_global[c3 - kLdsNodeBase];
                v.z = __int_as_float(c2);
                v.w = __int_as_float(c3);
            }
        }
        lds_store4(cache + i, v);
    }
    __syncthreads();
    return n_cached != 0;
}

// Rays[0..n) by persistent waves: the grid is sized to the machine, not to the queue.
// Chunks of 64 rays are dealt round-robin to the waves (chunk c belongs to wave c % n_waves:
// static, so no work-distribution atomics), and a wave tops its idle lanes up from its own
// stream whenever `refill` or more of them have finished, instead of idling them until its
// slowest ray ends.  Every wave reaches the end of its finite stream, so the grid always
// drains.  ANY = false: closest hits, sink(ray, shapeid, primid, b0, b1, b2, t).
// ANY = true: occlusion queries, sink(ray, occluded).  perm, if not null, is the order in
// which to take the rays (queue position -> ray index); sinks always get the ray index.
template <bool COUNT, bool ANY, bool CACHE, bool BIG, class Sink, bool QUAD = false>
__device__ __forceinline__ void persistent_traverse(const DeviceScene &s, const pt_ray *rays,
                                                    uint32_t n, bool fixup, const PersistentParams pp,
                                                    unsigned int *tie_count,
                                                    DeviceTraceCounters *work, Sink sink,
                                                    const uint32_t *perm = nullptr) {
    const int refill = pp.refill, leaf_lanes = pp.leaf_lanes, stack_cap = pp.stack_cap;
    int32_t *const stack = persistent_stack();
    const StackSpill spill = lane_spill(pp.spill, pp.spill_entries);
    // The top of the tree goes to LDS once per workgroup, and rays then start at its first node.
    // The scene's array names a child inside it kLdsNodeBase + position; a workgroup that keeps
    // fewer nodes than the array holds gives the children beyond its share their global names.
    LdsFloat4 *const cache = CACHE ? persistent_cache(stack_cap) : nullptr;
    int32_t root = 0;
    if (CACHE && copy_top_to_lds<QUAD>(s, pp, cache)) root = kLdsNodeBase;
    const float cmax = QUAD ? scene_max_coordinate(s) : 0.f;
    const uint32_t lane = threadIdx.x & 63u;
    // readfirstlane: the wave index is the same in all 64 lanes, which the compiler cannot see
    // from threadIdx; with it the stream bookkeeping and the loop control stay scalar
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(
        (int)(blockIdx.x * (kTraceBlock / 64) + (threadIdx.x >> 6)));
    const uint32_t n_waves = gridDim.x * (kTraceBlock / 64);
    const uint32_t n_chunks = (n + 63u) >> 6;
    // Which chunk of every round of n_waves chunks is this wave's.  Plainly its own index; or,
    // with xcd_slots, such that the workgroups of one XCD (blockIdx % 8) take one contiguous
    // eighth of the round: neighbouring chunks of an ordered queue then meet in one L2 instead
    // of all eight, while every round is still spread over the whole machine.
    uint32_t slot = wave;
    if (pp.xcd_slots && (gridDim.x & 7u) == 0u) {
        constexpr uint32_t wpb = kTraceBlock / 64;
        const uint32_t block = wave / wpb;
        slot = (block & 7u) * (n_waves >> 3) + (block >> 3) * wpb + (wave % wpb);
    }
    const uint32_t owned = slot < n_chunks ? (n_chunks - slot + n_waves - 1) / n_waves : 0;
    const uint32_t stream_len = owned << 6;     // positions in this wave's stream
    uint32_t pos = 0;                           // next unread position (wave-uniform)
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    uint32_t my_ray = kNone;
    OrderedState st;
    st.cur = kSentinel;
    st.tie = false;
    st.wild = false;
    st.primid = -1;
    st.other = -1;
    st.second = INFINITY;
    st.tbest = 0.f;
    unsigned long long nb = 0, nt = 0, nl = 0;  // slab tests, triangle tests, slab tests on LDS nodes
    unsigned int n_ties = 0;
    const bool any_exact = s.n_tri < 2;         // see occluded_wide
    int spilled = 0;                            // blocks of this lane's stack in global memory
    // Once a wave's stream has ended, its remaining rays are normally done within a few rounds.
    // A ray that is not (one that runs along a fan of slivers makes thousands of steps, one after
    // the other in its lane) would hold the whole launch: after tail_budget more rounds the wave
    // hands what it still has to the list of long rays, which a second kernel searches with a
    // whole wave per ray (long_ray_search), and ends.
    int tail_left = pp.tail_budget;
    // (Compiled into the four-child kernels only, the ones scenes beyond the caches get: there a
    // step is a trip to memory and such a ray costs milliseconds; the cache-resident kernel,
    // where it would cost a fraction of that, loses 1.5 % of its rate with the hand-over and the
    // second launch in place, and stays as it was.)
    constexpr bool hand_over_allowed = QUAD && !ANY;
    bool hand_over = hand_over_allowed && pp.long_list != nullptr && pp.tail_budget > 0;
    // A ray whose answer may depend on the visiting order is walked again in the reference's
    // fixed order.  That walk is serial and long, so a lane does not start it on the spot with
    // the other 63 waiting: it parks the ray here and goes on, and the parked rays of a wave
    // are walked side by side once kTieBatch lanes hold one, when a lane with both of its two
    // places taken has a third to park, and at the end of the wave's stream.
    constexpr int kTieBatch = PT_TIE_BATCH;
    uint32_t parked = kNone, parked2 = kNone;   // two places per lane: batches get larger
    // Top bit of a parked ray index (indices stay below 2^31): the ray has exactly two candidates
    // and the two floats parked with it hold their triangle ids instead of distances.
    constexpr uint32_t kReplay = 0x80000000u;
    float parked_limit = INFINITY, parked2_limit = INFINITY;   // see traverse_skiplink
    float parked_bound = INFINITY, parked2_bound = INFINITY;   // the ordered hit, widened by the window
    auto walk_parked = [&]() {
        while (parked != kNone) {
            int32_t shapeid, primid;
            float b0, b1, b2, t;
            unsigned long long xb = 0, xt = 0;
            const uint32_t ray = parked & ~kReplay;
            const float4 ro = reinterpret_cast<const float4 *>(rays + ray)[0];
            const float4 rd = reinterpret_cast<const float4 *>(rays + ray)[1];
            if (parked & kReplay) {
                // two candidates (their ids are in the two floats): decided at their two leaves;
                // should neither be accepted there, the candidates were not what they seemed and
                // the ray gets the reference's loop as written
                replay_candidates(s, ro, rd, __float_as_int(parked_limit),
                                  __float_as_int(parked_bound), shapeid, primid, b0, b1, b2, t);
                if (primid < 0)
                    traverse_skiplink<false, false>(s, ro, rd, shapeid, primid, b0, b1, b2, t, xb, xt);
            } else {
                traverse_skiplink<false, false>(s, ro, rd, shapeid, primid, b0, b1, b2, t, xb, xt,
                                                parked_limit);
                // The bounded walk starts from a hit the ordered traversal has found, so it must
                // come back with one at most the tie window beyond it.  If it does not, the
                // rounding margin in its starting distance was too small for this ray after all:
                // the walk is repeated from the ray's own reach, which is the reference's loop as
                // written.
                if (parked_limit < INFINITY && (primid < 0 || !(t <= parked_bound)))
                    traverse_skiplink<false, false>(s, ro, rd, shapeid, primid, b0, b1, b2, t, xb, xt);
            }
            if constexpr (!ANY) sink(ray, shapeid, primid, b0, b1, b2, t);
            parked = parked2;
            parked_limit = parked2_limit;
            parked_bound = parked2_bound;
            parked2 = kNone;
        }
    };
    for (;;) {
        const bool finished = my_ray != kNone && st.cur == kSentinel &&
                              !stack_service<ANY>(st, stack, spill, 0, spilled);
        if constexpr (!ANY) {
            const bool to_park = finished && fixup && ordered_is_tie(st);
            const unsigned long long waiting = __ballot(parked != kNone);
            if (waiting && ((int)__popcll(waiting) >= kTieBatch || __ballot(to_park && parked2 != kNone)))
                walk_parked();
        }
        if (finished) {                                         // retire a finished ray
            if constexpr (ANY) {
                bool occ = st.primid >= 0;
                if (st.tie) {
                    const float4 ro = reinterpret_cast<const float4 *>(rays + my_ray)[0];
                    const float4 rd = reinterpret_cast<const float4 *>(rays + my_ray)[1];
                    occ = occluded_skiplink(s, ro, rd);
                }
                sink(my_ray, occ);
            } else {
                const bool tie = ordered_is_tie(st);
                n_ties += tie ? 1u : 0u;
                if (tie && fixup) {
                    // Any hit found here bounds the answer -- also the nearest so far of a search cut
                    // short by a full stack: the shader ends no farther than a triangle it would
                    // accept -- unless the search was cut short by a value that is not finite.
                    float lim = INFINITY, lim_bound = INFINITY;
                    uint32_t mark = my_ray;
                    if ((s.tri_leaf || s.tri_leaf_first) && !st.tie && st.primid >= 0 && st.other >= 0) {
                        mark |= kReplay;
                        lim = __int_as_float(st.primid);
                        lim_bound = __int_as_float(st.other);
                    } else if (s.tri_leaf && !st.wild && st.primid >= 0) {
                        // (Only where every triangle has one leaf: a triangle in several leaves
                        // can be accepted through a leaf that the ray enters behind the hit, if
                        // the loop gets there while it still holds something farther -- a walk
                        // that starts culling at the hit would pass such a leaf by.)
                        lim_bound = st.tbest + st.tbest * PT_TIE_WINDOW;
                        // The walk must enter every box on the way to that hit.  A box's computed
                        // entry distance can lie past the computed hit distance by rounding (a flat
                        // triangle's box has no depth): by a few units in the last place of the
                        // coordinates involved, times 1/d.  64 of them, taken at the scene's
                        // largest coordinates, are added; a ray along an axis gets no bound.
                        const float4 lo = s.nodes[0], hi = s.nodes[1];
                        const float ex = (fabsf(st.r.ox) + fmaxf(fabsf(lo.x), fabsf(hi.x))) * fabsf(st.r.ix);
                        const float ey = (fabsf(st.r.oy) + fmaxf(fabsf(lo.y), fabsf(hi.y))) * fabsf(st.r.iy);
                        const float ez = (fabsf(st.r.oz) + fmaxf(fabsf(lo.z), fabsf(hi.z))) * fabsf(st.r.iz);
                        lim = lim_bound + 3.814697265625e-06f * fmaxf(ex, fmaxf(ey, ez));
                    }
                    if (parked == kNone) {
                        parked = mark;
                        parked_limit = lim;
                        parked_bound = lim_bound;
                    } else {
                        parked2 = mark;
                        parked2_limit = lim;
                        parked2_bound = lim_bound;
                    }
                } else
                    sink(my_ray, st.primid >= 0 ? 0 : -1, st.primid, st.b0, st.b1, st.b2, st.tbest);
            }
            my_ray = kNone;
        }
        const bool idle = my_ray == kNone;
        const unsigned long long m = __ballot(idle);
        if (pos < stream_len && m) {                            // top up idle lanes
            const uint32_t p = pos + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (idle && p < stream_len) {
                const uint32_t idx = (((p >> 6) * n_waves + slot) << 6) + (p & 63u);
                if (idx < n) {
                    const uint32_t ri = perm ? perm[idx] : idx;
                    const float4 ro = reinterpret_cast<const float4 *>(rays + ri)[0];
                    const float4 rd = reinterpret_cast<const float4 *>(rays + ri)[1];
                    // A ray that does not move along an axis (1/d infinite) and lies exactly in a bounding plane
                    // of a box misses that box in the reference's slab test (0 * inf is a NaN on
                    // one side of the slab only) but not a flat leaf box in the same plane inside it
                    // (NaN on both sides: the slab is skipped), and the Moller-Trumbore test then
                    // reports a hit for a ray in the triangle's plane (every comparison with its
                    // NaNs is false).  The coarse boxes of the four-child walk cannot tell: such
                    // occlusion rays go to the fixed order.  (The watertight test turns a ray in
                    // the triangle's plane down, so closest hits are not affected.)
                    const bool axis_ray = QUAD && ANY && !(fabsf(1.0f / rd.x) < INFINITY && fabsf(1.0f / rd.y) < INFINITY &&
                                                   fabsf(1.0f / rd.z) < INFINITY);
                    if constexpr (ANY) any_begin(st, ro, rd, stack, any_exact || axis_ray, root);
                    else ordered_begin(st, ro, rd, stack, root);
                    if constexpr (QUAD) coarse_ray_terms(st, cmax);
                    my_ray = ri;
                    spilled = 0;
                }
            }
            pos += (uint32_t)__popcll(m);
        }
        if (__ballot(my_ray != kNone) == 0ull) {
            if (pos >= stream_len) break;
            continue;                                           // tail positions past the queue end
        }
        const bool tail = pos >= stream_len;
        if (hand_over && tail && tail_left <= 0) {
            hand_over = false;                                  // one attempt: a full list leaves the rest here
            if (my_ray != kNone) {
                const uint32_t at = atomicAdd(&pp.long_list[0], 1u);
                if (at < pp.long_cap) {
                    pp.long_list[kLongListHeader + at] = my_ray;
                    my_ray = kNone;
                    st.cur = kSentinel;
                    spilled = 0;
                }
            }
            continue;
        }
        int rounds = hand_over && tail ? tail_left : INT_MAX;   // (a plain local: stays in a register)
        ordered_advance<COUNT, ANY, CACHE, BIG, QUAD>(s, st, stack, tail ? 0 : 64 - refill, leaf_lanes, nb, nt,
                                                stack_cap, cache, &nl, hand_over_allowed ? &rounds : nullptr);
        if (hand_over && tail) tail_left = rounds;
    }
    if constexpr (!ANY) walk_parked();
    if (!ANY && n_ties) atomicAdd(tie_count, n_ties);
    if (COUNT) {
        nb = wave_sum(nb);
        nt = wave_sum(nt);
        nl = wave_sum(nl);
        if (lane == 0) {
            atomicAdd(&work->box_tests, nb);
            atomicAdd(&work->tri_tests, nt);
            atomicAdd(&work->lds_box_tests, nl);
        }
    }
}

// ---- one wave per ray --------------------------------------------------------------------------
// For the rays the persistent loop hands over (PersistentParams::long_list): the closest hit of
// one ray searched by all 64 lanes of a wave.  The pending nodes and leaves sit on one stack in
// LDS; each round the lanes take the top 64 entries, one each -- a node's two child boxes are
// tested and the hit children pushed, a leaf's triangle is tested -- and the hits of the round
// are merged into the nearest and the second nearest triangle seen so far.  Boxes and triangles
// are culled with the nearest hit widened by the tie window, as in ordered_advance, so whatever
// lies within the window of the final nearest hit is seen whatever the order, and the rules that
// send a ray to the reference's fixed order are the same: a second triangle within the window,
// a hit within the window of the ray's own reach, a value that is not finite; also a stack that
// outgrows its LDS.  Such a ray gets traverse_skiplink as it is (by every lane alike: same
// addresses, one answer).  The triangle tests are the same code on the same numbers as anywhere
// else, so the record written is the one the persistent loop would have written.
// The two-child nodes are walked, also where the persistent loop walked the four-child ones.
constexpr int kLongStack = 2048;            // entries per wave
__device__ __forceinline__ bool ray_is_ordinary(const float4 ro, const float4 rd);
template <bool BIG, class Sink>
__device__ __forceinline__ void long_ray_search(const DeviceScene &s, const pt_ray *rays, uint32_t ray,
                                                int32_t *stk, bool fixup, unsigned int *tie_count,
                                                DeviceTraceCounters *work, Sink sink) {
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const float4 ro = reinterpret_cast<const float4 *>(rays + ray)[0];
    const float4 rd = reinterpret_cast<const float4 *>(rays + ray)[1];
    int32_t shapeid = -1, primid = -1;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, t = ro.w;
    unsigned long long nb = 0, nt = 0;
    bool exact = !ray_is_ordinary(ro, rd) || s.n_tri < 2;
    if (!exact) {
        const RayPre r = precompute_ray(ro.x, ro.y, ro.z, rd.x, rd.y, rd.z);
        const float tmax = ro.w;
        float t1 = tmax, t2 = INFINITY, c0 = 0.f, c1 = 0.f, c2 = 0.f;     // nearest, second nearest
        int32_t p1 = -1, p2 = -1;
        float bound = tmax + tmax * PT_TIE_WINDOW;
        bool tie = false, overflow = false;
        int top = 1;
        if (lane == 0) stk[0] = 0;                                      // the root
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        while (top > 0) {
            const int take = top < 64 ? top : 64;
            const int32_t e = lane < take ? stk[top - 1 - lane] : kSentinel;
            top -= take;
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            bool push_a = false, push_b = false, hit = false;
            int32_t child_a = 0, child_b = 0, hp = -1;
            float h0 = 0.f, h1 = 0.f, h2 = 0.f, ht = 0.f;
            if (lane < take && e >= 0) {
                const float4 *np = wide_node<BIG>(s, e);
                const float4 a = np[0], b = np[1], c = np[2];
                const int2 ch = reinterpret_cast<const int2 *>(np + 3)[0];
                float ta, tb;
                push_a = intersect_box(r, a.x, a.y, a.z, a.w, c.x, c.y, bound, ta);
                push_b = intersect_box(r, b.x, b.y, b.z, b.w, c.z, c.w, bound, tb);
                child_a = ch.x;
                child_b = ch.y;
                nb += 2;
            } else if (lane < take) {
                const float4 *tp = wide_tri<BIG>(s, e);
                const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
                hit = intersect_triangle_watertight(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w,
                                                    q2.x, bound, h0, h1, h2, ht);
                hp = __float_as_int(q2.y);
                nt += 1;
            }
            // the round's hits, one after the other, into (t1, p1) and (t2, p2): the two nearest
            // distinct triangles (a triangle met again through another of its leaves is no second)
            unsigned long long hm = __ballot(hit);
            while (hm) {
                const int src = __ffsll((long long)hm) - 1;
                hm &= hm - 1ull;
                const float tt = __shfl(ht, src, 64);
                const int32_t pp = __shfl(hp, src, 64);
                if (!(tt == tt)) {
                    tie = true;
                } else if (pp == p1) {
                } else if (p1 < 0 || tt < t1) {
                    if (p1 >= 0) { t2 = t1; p2 = p1; }
                    t1 = tt; p1 = pp;
                    c0 = __shfl(h0, src, 64); c1 = __shfl(h1, src, 64); c2 = __shfl(h2, src, 64);
                } else if (pp != p2 && tt < t2) {
                    t2 = tt; p2 = pp;
                } else if (pp == p2) {
                    t2 = fminf(t2, tt);
                }
            }
            if (p1 >= 0) {
                bound = t1 + t1 * PT_TIE_WINDOW;
                tie |= bound >= tmax;
            }
            const unsigned long long ma = __ballot(push_a), mb = __ballot(push_b);
            const int na = (int)__popcll(ma), n_push = na + (int)__popcll(mb);
            if (top + n_push > kLongStack) {
                overflow = true;
                break;
            }
            if (push_a) stk[top + (int)__popcll(ma & below)] = child_a;
            if (push_b) stk[top + na + (int)__popcll(mb & below)] = child_b;
            top += n_push;
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
        }
        exact = overflow || tie || (p1 >= 0 && t2 <= t1 + t1 * PT_TIE_WINDOW);
        primid = p1;
        shapeid = p1 >= 0 ? 0 : -1;
        b0 = c0; b1 = c1; b2 = c2;
        t = p1 >= 0 ? t1 : tmax;
        if (exact && lane == 0) atomicAdd(tie_count, 1u);
        exact = exact && fixup;
    }
    if (exact) {
        unsigned long long xb = 0, xt = 0;
        traverse_skiplink<false, false>(s, ro, rd, shapeid, primid, b0, b1, b2, t, xb, xt);
    }
    if (lane == 0) sink(ray, shapeid, primid, b0, b1, b2, t);
    if (work) {
        nb = wave_sum(nb);
        nt = wave_sum(nt);
        if (lane == 0) {
            atomicAdd(&work->box_tests, nb);
            atomicAdd(&work->tri_tests, nt);
        }
    }
}

// ---- the persistent loop for two-level scenes ------------------------------------------------
// One voted loop over both levels.  A lane is either in world space, walking the top-level
// tree's two-child nodes (from s.top_root on; a negative child is the complement of an instance
// index), or
// inside an instance, walking that mesh's nodes and triangle records with the ray taken to the
// instance's object space.  The top level's nodes lie in the same array as the meshes', so a
// node step is the same code on either level.  Besides the node step and the triangle step there is a third kind of
// step, voted like the other two: entering an instance (transform the ray, put an exit mark on
// the stack, go to the mesh's root) and leaving one (the exit mark has been popped: back to the
// world-space ray).  Everything below the mark on a lane's stack belongs to the top level,
// everything above it to the mesh, so one stack serves both.
//
// Why this returns the fixed-order result of both levels: as one level down.  The ray parameter
// is the same in both spaces, every box (top-level or mesh) is culled with the nearest hit
// widened by the tie window, so any second candidate within the window of the nearest -- in the
// same mesh or in another instance -- is seen, and such a ray (also a non-finite one, or one
// that outgrows stack and spill slots) is traced again by its lane in the fixed order of both
// levels (closest_hit_two_level / occluded_two_level with exact = true, which need no stack).
constexpr int32_t kExitMark = INT_MIN + 1;

struct InstanceRay {
    float4 o, d;          // the ray in the instance's object space
    int32_t root;         // first two-child node of the instance's mesh
    uint32_t n_tri;
    float cmax;           // largest coordinate of the mesh's boxes (see coarse_ray_terms)
};
__device__ __forceinline__ InstanceRay instance_ray(const DeviceScene &s, int32_t instance,
                                                    const float4 ro, const float4 rd) {
    const float4 *ip = reinterpret_cast<const float4 *>(s.instances + instance);
    const float4 r0 = ip[0], r1 = ip[1], r2 = ip[2];
    const uint4 o0 = reinterpret_cast<const uint4 *>(ip)[3];
    const uint4 o1 = reinterpret_cast<const uint4 *>(ip)[4];
    InstanceRay x;
    x.root = (int32_t)o0.y;
    x.n_tri = o1.y;
    x.cmax = __uint_as_float(o1.w);
    x.o.x = ((r0.x * ro.x + r0.y * ro.y) + r0.z * ro.z) + r0.w;      // as enter_instance
    x.o.y = ((r1.x * ro.x + r1.y * ro.y) + r1.z * ro.z) + r1.w;
    x.o.z = ((r2.x * ro.x + r2.y * ro.y) + r2.z * ro.z) + r2.w;
    x.o.w = ro.w;
    x.d.x = (r0.x * rd.x + r0.y * rd.y) + r0.z * rd.z;
    x.d.y = (r1.x * rd.x + r1.y * rd.y) + r1.z * rd.z;
    x.d.z = (r2.x * rd.x + r2.y * rd.y) + r2.z * rd.z;
    x.d.w = rd.w;
    return x;
}

// The per-space part of a lane's state: what the slab and triangle tests read of the ray.
template <bool ANY>
__device__ __forceinline__ void set_space_ray(OrderedState &st, const float4 ro, const float4 rd) {
    if (ANY) {                      // as any_begin: 1/d, and the direction in the shear slots
        st.r.ox = ro.x; st.r.oy = ro.y; st.r.oz = ro.z;
        st.r.ix = 1.0f / rd.x; st.r.iy = 1.0f / rd.y; st.r.iz = 1.0f / rd.z;
        st.r.sx = rd.x; st.r.sy = rd.y; st.r.sz = rd.z;
        st.r.axis = 2;
    } else {
        st.r = precompute_ray(ro.x, ro.y, ro.z, rd.x, rd.y, rd.z);
    }
}
__device__ __forceinline__ bool ray_is_ordinary(const float4 ro, const float4 rd) {
    const float chk = ((ro.x + ro.y) + (ro.z + rd.x)) + (rd.y + rd.z);
    return (fabsf(chk) < INFINITY) && (ro.w == ro.w) && !(rd.x == 0.f && rd.y == 0.f && rd.z == 0.f);
}

// QUAD: inside the instances the four-child nodes are walked (DeviceScene::qnodes / qtris over
// the meshes' records, built like a single-level scene's); the top-level tree stays two-child.
template <bool COUNT, bool ANY, bool BIG, class Sink, bool CACHE = false, bool QUAD = false>
__device__ __forceinline__ void persistent_traverse_two_level(
    const DeviceScene &s, const pt_ray *rays, uint32_t n, bool fixup, const PersistentParams pp,
    unsigned int *tie_count, DeviceTraceCounters *work, Sink sink) {
    const int refill = pp.refill, leaf_lanes = pp.leaf_lanes, stack_cap = pp.stack_cap;
    int32_t *const stack = persistent_stack();
    const StackSpill spill = lane_spill(pp.spill, pp.spill_entries);
    // the top of the top-level tree, which every ray walks first, from LDS (as one level down)
    LdsFloat4 *const cache = CACHE ? persistent_cache(stack_cap) : nullptr;
    int32_t top_root = (int32_t)s.top_root;
    if (CACHE && copy_top_to_lds<false>(s, pp, cache)) top_root = kLdsNodeBase;
    const uint32_t lane = threadIdx.x & 63u;
    // readfirstlane: the wave index is the same in all 64 lanes, which the compiler cannot see
    // from threadIdx; with it the stream bookkeeping and the loop control stay scalar
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(
        (int)(blockIdx.x * (kTraceBlock / 64) + (threadIdx.x >> 6)));
    const uint32_t n_waves = gridDim.x * (kTraceBlock / 64);
    const uint32_t n_chunks = (n + 63u) >> 6;
    const uint32_t owned = wave < n_chunks ? (n_chunks - wave + n_waves - 1) / n_waves : 0;
    const uint32_t stream_len = owned << 6;
    uint32_t pos = 0;
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    uint32_t my_ray = kNone;
    OrderedState st;
    st.cur = kSentinel;
    st.tie = false;
    st.wild = false;
    st.primid = -1;
    st.other = -1;
    st.second = INFINITY;
    st.tbest = 0.f;
    st.tmax = 0.f;
    st.sp = 0;
    int32_t cur_inst = -1;         // instance the lane is inside (-1: world space)
    int32_t best_shape = -1;       // instance of the nearest hit so far
    unsigned long long nb = 0, nt = 0;
    unsigned int n_ties = 0;
    int spilled = 0;
    for (;;) {
        const bool finished = my_ray != kNone && st.cur == kSentinel &&
                              !stack_service<ANY>(st, stack, spill, 0, spilled);
        if (finished) {                                         // retire a finished ray
            const float4 ro = reinterpret_cast<const float4 *>(rays + my_ray)[0];
            const float4 rd = reinterpret_cast<const float4 *>(rays + my_ray)[1];
            if constexpr (ANY) {
                bool occ = st.primid >= 0;
                if (st.tie) occ = occluded_two_level(s, ro, rd, stack, true);
                sink(my_ray, occ);
            } else {
                int32_t shape = st.primid >= 0 ? best_shape : -1, prim = st.primid;
                float b0 = st.b0, b1 = st.b1, b2 = st.b2, t = st.tbest;
                if (ordered_is_tie(st)) {
                    ++n_ties;
                    if (fixup) {
                        unsigned long long xb = 0, xt = 0;
                        closest_hit_two_level<false>(s, ro, rd, stack, true, true, shape, prim, b0, b1,
                                                     b2, t, xb, xt);
                    }
                }
                sink(my_ray, shape, prim, b0, b1, b2, t);
            }
            my_ray = kNone;
        }
        const bool idle = my_ray == kNone;
        const unsigned long long m = __ballot(idle);
        if (pos < stream_len && m) {                            // hand idle lanes their next rays
            const uint32_t p = pos + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (idle && p < stream_len) {
                const uint32_t idx = (((p >> 6) * n_waves + wave) << 6) + (p & 63u);
                if (idx < n) {
                    const float4 ro = reinterpret_cast<const float4 *>(rays + idx)[0];
                    const float4 rd = reinterpret_cast<const float4 *>(rays + idx)[1];
                    my_ray = idx;
                    set_space_ray<ANY>(st, ro, rd);
                    st.tmax = ro.w;
                    st.tbest = ro.w;
                    st.bound = ANY ? ro.w : ro.w + ro.w * PT_TIE_WINDOW;
                    st.second = INFINITY;
                    st.b0 = 0.f; st.b1 = 0.f; st.b2 = 0.f;
                    st.primid = -1;
                    best_shape = -1;
                    cur_inst = -1;
                    spilled = 0;
                    stack[0] = kSentinel;
                    st.sp = 1;
                    st.cur = top_root;                          // the top-level root
                    st.tie = !ray_is_ordinary(ro, rd);          // whatever the fixed order makes of it
                    if (st.tie) st.cur = kSentinel;
                }
            }
            pos += (uint32_t)__popcll(m);
        }
        if (__ballot(my_ray != kNone) == 0ull) {
            if (pos >= stream_len) break;
            continue;                                           // tail positions past the queue end
        }
        // ---- voted steps until enough lanes have finished for a refill to pay ----------------
        const int min_active = pos < stream_len ? 64 - refill : 0;
        for (;;) {
            // The vote: 0 node step, 1 triangle step, 2 instance step (enter or leave), 3 enough
            // lanes are idle.  The same value in every lane.
            bool at_node, at_tri, at_inst;
            auto vote = [&]() {
                const bool in_obj = cur_inst >= 0;
                at_node = st.cur >= 0;
                at_tri = in_obj & (st.cur < 0) & (st.cur > kExitMark);
                at_inst = (st.cur != kSentinel) & !at_node & !at_tri;   // an instance to enter, or the exit mark
                const int n_node = (int)__popcll(__ballot(at_node));
                const int n_tri = (int)__popcll(__ballot(at_tri));
                const int n_inst = (int)__popcll(__ballot(at_inst));
                if (n_node + n_tri + n_inst <= min_active) return 3;
                if (n_node != 0 && n_tri < leaf_lanes && n_inst < leaf_lanes) return 0;
                return (n_inst != 0 && n_inst >= n_tri) ? 2 : 1;
            };
            int mode = vote();
            // node steps in a loop of their own: only position and stack pointer change in it
            while (mode == 0) {
                if (QUAD && at_node && cur_inst >= 0) {         // a four-child node of a mesh
                    const float4 *np = quad_node<BIG>(s, st.cur);
                    const float4 f0 = np[0], f1 = np[1], f2 = np[2];
                    const int4 ch = reinterpret_cast<const int4 *>(np)[3];
                    quad_node_step<COUNT>(st, stack, stack_cap, f0, f1, f2, ch, nb);
                } else if (at_node) {
                    float4 a, b, c;
                    int2 ch;
                    if (CACHE && st.cur >= kLdsNodeBase) {      // a node of the top of the top level, in LDS
                        const LdsFloat4 *np = cache + ((st.cur - kLdsNodeBase) << 2);
                        a = lds_load4(np); b = lds_load4(np + 1); c = lds_load4(np + 2);
                        ch = lds_load_int2(np + 3);
                    } else {
                        const float4 *np = wide_node<BIG>(s, st.cur);
                        a = np[0]; b = np[1]; c = np[2];
                        ch = reinterpret_cast<const int2 *>(np + 3)[0];
                    }
                    if (COUNT) nb += 2;
                    float ta, tb;
                    const bool ha = intersect_box(st.r, a.x, a.y, a.z, a.w, c.x, c.y, st.bound, ta);
                    const bool hb = intersect_box(st.r, b.x, b.y, b.z, b.w, c.z, c.w, st.bound, tb);
                    const bool take_b = hb && (!ha || (tb < ta));
                    const bool both = ha && hb, none = !ha && !hb;
                    const bool full = both && (st.sp >= stack_cap);  // see ordered_advance
                    int32_t next = take_b ? ch.y : ch.x;
                    const int32_t far_c = take_b ? ch.x : ch.y;
                    if (both) {
                        stack[st.sp * kTraceBlock] = full ? st.cur : far_c;
                        ++st.sp;
                    }
                    if (none) {
                        --st.sp;
                        next = stack[st.sp * kTraceBlock];
                    }
                    st.cur = full ? kSentinel : next;
                }
                mode = vote();
            }
            if (mode == 3) break;
            if (mode == 2) {
                if (at_inst) {
                    const float4 ro = reinterpret_cast<const float4 *>(rays + my_ray)[0];
                    const float4 rd = reinterpret_cast<const float4 *>(rays + my_ray)[1];
                    if (st.cur == kExitMark) {                  // leave the instance
                        cur_inst = -1;
                        set_space_ray<ANY>(st, ro, rd);
                        --st.sp;
                        st.cur = stack[st.sp * kTraceBlock];
                    } else if (st.sp >= stack_cap) {            // no room for the exit mark: as a full node step
                        stack[st.sp * kTraceBlock] = st.cur;
                        ++st.sp;
                        st.cur = kSentinel;
                    } else {                                    // enter the instance ~cur
                        const int32_t inst = ~st.cur;
                        const InstanceRay x = instance_ray(s, inst, ro, rd);
                        stack[st.sp * kTraceBlock] = kExitMark;
                        ++st.sp;
                        cur_inst = inst;
                        set_space_ray<ANY>(st, x.o, x.d);
                        if constexpr (QUAD) coarse_ray_terms(st, x.cmax);
                        st.cur = x.root;
                        // an object-space ray that is not finite, and (occlusion only) a mesh of a
                        // single triangle, whose filler record is safe under the watertight test
                        // alone, or with four-child nodes a ray along an axis (see persistent_traverse):
                        // left to the fixed order
                        const bool axis_ray = QUAD && ANY &&
                                              !(fabsf(st.r.ix) < INFINITY && fabsf(st.r.iy) < INFINITY &&
                                                fabsf(st.r.iz) < INFINITY);
                        if (!ray_is_ordinary(x.o, x.d) || (ANY && x.n_tri < 2) || axis_ray) {
                            st.tie = true;
                            st.cur = kSentinel;
                        }
                    }
                }
            } else if (at_tri) {                                // triangle step, inside an instance
                const float4 *tp = QUAD ? quad_tri<BIG>(s, st.cur) : wide_tri<BIG>(s, st.cur);
                const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
                bool in_leaf_box = true;
                if constexpr (QUAD) {                           // the leaf's exact box (see ordered_advance)
                    const float4 q3 = tp[3];
                    float t;
                    in_leaf_box = intersect_box(st.r, q2.z, q3.y, q2.w, q3.z, q3.x, q3.w, st.bound, t);
                    if (COUNT) ++nb;
                }
                if (COUNT) nt += in_leaf_box ? 1 : 0;
                if constexpr (ANY) {
                    float m1, m2, temp;
                    moller_trumbore(st.r.ox, st.r.oy, st.r.oz, st.r.sx, st.r.sy, st.r.sz, q0.x, q0.y,
                                    q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, m1, m2, temp);
                    const bool hit = !((m1 < -PT_MT_EPS) | (m1 > 1.f + PT_MT_EPS) | (m2 < -PT_MT_EPS) |
                                       (m1 + m2 > 1.f + PT_MT_EPS) | (temp < -PT_MT_EPS) |
                                       (temp > st.tmax + PT_MT_EPS)) &          // bvh.glslh:301
                                     in_leaf_box;
                    st.primid = hit ? 0 : st.primid;
                    --st.sp;
                    const int32_t popped = stack[st.sp * kTraceBlock];
                    st.cur = hit ? kSentinel : popped;
                } else {
                    float h0 = 0.f, h1 = 0.f, h2 = 0.f, ht = 0.f;
                    const bool hit = intersect_triangle_watertight(st.r, q0.x, q0.y, q0.z, q0.w, q1.x,
                                                                   q1.y, q1.z, q1.w, q2.x, st.bound, h0,
                                                                   h1, h2, ht) &
                                     in_leaf_box;
                    const int32_t tri = __float_as_int(q2.y);
                    const bool bad = hit & !(ht == ht);
                    const bool had = st.primid >= 0;
                    const bool closer = hit & !bad & (!had | (ht < st.tbest));
                    // the triangle already held, met again (another leaf of it in a tree built with
                    // spatial splits, or its instance reached through another top-level leaf), is no
                    // competing candidate
                    const bool same = (tri == st.primid) & (cur_inst == best_shape);
                    const bool farther = hit & !bad & !closer & !same;
                    const float s_old = fminf(st.second, st.tbest), s_new = fminf(st.second, ht);
                    st.second = (closer & had) ? s_old : (farther ? s_new : st.second);
                    st.tbest = closer ? ht : st.tbest;
                    st.b0 = closer ? h0 : st.b0;
                    st.b1 = closer ? h1 : st.b1;
                    st.b2 = closer ? h2 : st.b2;
                    st.primid = closer ? tri : st.primid;
                    best_shape = closer ? cur_inst : best_shape;
                    const float nbound = ht + ht * PT_TIE_WINDOW;   // see ordered_advance
                    st.bound = closer ? nbound : st.bound;
                    st.tie |= bad | (hit & !bad & (nbound >= st.tmax));
                    --st.sp;
                    const int32_t popped = stack[st.sp * kTraceBlock];
                    st.cur = bad ? kSentinel : popped;
                }
            }
        }
    }
    if (!ANY && n_ties) atomicAdd(tie_count, n_ties);
    if (COUNT) {
        nb = wave_sum(nb);
        nt = wave_sum(nt);
        if (lane == 0) {
            atomicAdd(&work->box_tests, nb);
            atomicAdd(&work->tri_tests, nt);
        }
    }
}

template <bool COUNT, bool CACHE, bool BIG, bool QUAD = false>
__device__ __forceinline__ void persistent_closest_hits(const DeviceScene &s, const pt_ray *rays,
                                                        pt_hit *hits, uint32_t n, bool fixup,
                                                        const PersistentParams pp,
                                                        unsigned int *tie_count,
                                                        DeviceTraceCounters *work,
                                                        const uint32_t *perm = nullptr) {
    persistent_traverse<COUNT, false, CACHE, BIG, HitSink, QUAD>(s, rays, n, fixup, pp, tie_count, work,
                                                                HitSink{hits}, perm);
}

// Compacting append to a queue: every lane with `keep` gets a distinct slot.  One atomic
// per wave (ballot + prefix count), as in the reference author's plan of a sum-scan over
// per-ray spawn counts (src/main_granite.cpp:409-411).
__device__ __forceinline__ uint32_t queue_append(bool keep, unsigned int *count) {
    const unsigned long long m = __ballot(keep);
    if (!m) return 0;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(count, (unsigned int)__popcll(m));
    base = __shfl(base, leader, 64);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// Workgroup-wide form: one atomic per workgroup instead of one per wave (a single counter
// word sustains only ~90 returning atomics per microsecond chip-wide, which at one atomic
// per 64 rays would cap a queue at ~6 Grays/s of appends and serialise the launch).
// Every thread of the workgroup must call it; `scratch` is >= 17 words of LDS.
__device__ __forceinline__ uint32_t block_queue_append(bool keep, unsigned int *count,
                                                       unsigned int *scratch) {
    const unsigned long long m = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_waves = (blockDim.x + 63) >> 6;
    if (lane == 0) scratch[wave] = (unsigned int)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int total = 0;
        for (int w = 0; w < n_waves; ++w) {
            const unsigned int c = scratch[w];
            scratch[w] = total;          // exclusive prefix over the waves
            total += c;
        }
        scratch[16] = total ? atomicAdd(count, total) : 0u;
    }
    __syncthreads();
    const uint32_t slot = scratch[16] + scratch[wave] +
                          (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();                     // scratch may be reused by the caller
    return slot;
}

} // namespace mipt
