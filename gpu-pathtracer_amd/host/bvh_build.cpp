// Deterministic host BVH builder (binned SAH or spatial median), built
// directly into pre-order storage.
//
// Stands in for RadeonRays::Bvh::Build as called from
// src/bvh/bvh_builder.cpp:61-62; that class is not part of the reference
// checkout, so the topology it would produce is not reproducible.  What is
// kept are the properties the rest of the reference depends on: binary tree,
// exactly one primitive per leaf (the shader intersects one triangle per leaf,
// shaders/RadeonRays/bvh.glslh:326-333), node bounds equal to the union of the
// inflated leaf boxes below, and the options of src/bvh/bvh_builder.cpp:18-32.
#include "bvh.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <thread>

namespace mipt {

namespace {

struct Aabb {
    float lo[3], hi[3];
    void clear() {
        lo[0] = lo[1] = lo[2] = FLT_MAX;
        hi[0] = hi[1] = hi[2] = -FLT_MAX;
    }
    void grow(const pt_bbox &b) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::fmin(lo[k], b.pmin[k]);
            hi[k] = std::fmax(hi[k], b.pmax[k]);
        }
    }
    void grow(const Aabb &b) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::fmin(lo[k], b.lo[k]);
            hi[k] = std::fmax(hi[k], b.hi[k]);
        }
    }
    void grow_point(const float *p) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::fmin(lo[k], p[k]);
            hi[k] = std::fmax(hi[k], p[k]);
        }
    }
    float half_area() const {
        const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        return ex * ey + ex * ez + ey * ez;
    }
};

struct Builder {
    const pt_bbox *leaf;
    std::vector<float> cent;      // 3 per primitive
    std::vector<uint32_t> prims;  // permutation being partitioned
    BvhTree *tree;
    bool use_sah;
    int num_bins;
    float traversal_cost;
    std::atomic<int> free_threads{0};

    static constexpr uint32_t kSpawnThreshold = 1u << 15;

    void emit_leaf(uint32_t node, uint32_t p) {
        pt_bbox b = leaf[p];
        b.pmin[3] = 0.f;
        b.pmax[3] = 0.f;
        tree->bounds[node] = b;
        tree->right[node] = -1;
        tree->prim[node] = (int32_t)p;
    }

    // Returns the split position in [begin+1, end-1]; partitions prims in place.
    uint32_t split(uint32_t begin, uint32_t end, const Aabb &cb) {
        const uint32_t n = end - begin;
        int axis = 0;
        {
            const float e[3] = {cb.hi[0] - cb.lo[0], cb.hi[1] - cb.lo[1], cb.hi[2] - cb.lo[2]};
            if (e[1] > e[axis]) axis = 1;
            if (e[2] > e[axis]) axis = 2;
            if (!(e[axis] > 0.f)) return begin + n / 2;  // all centroids coincide
        }
        int best_bin = -1;
        int nb = use_sah ? num_bins : 2;
        float k1 = 0.f;

        if (use_sah && n > 2) {
            float best_cost = FLT_MAX;
            std::vector<Aabb> bin_box((size_t)nb), right_box((size_t)nb);
            std::vector<uint32_t> bin_cnt((size_t)nb);
            for (int a = 0; a < 3; ++a) {
                const float ext = cb.hi[a] - cb.lo[a];
                if (!(ext > 0.f)) continue;
                const float scale = (float)nb / ext;
                for (int b = 0; b < nb; ++b) {
                    bin_box[b].clear();
                    bin_cnt[b] = 0;
                }
                for (uint32_t i = begin; i < end; ++i) {
                    const uint32_t p = prims[i];
                    int b = (int)(scale * (cent[3 * (size_t)p + a] - cb.lo[a]));
                    b = std::min(std::max(b, 0), nb - 1);
                    bin_cnt[b]++;
                    bin_box[b].grow(leaf[p]);
                }
                Aabb acc;
                acc.clear();
                for (int b = nb - 1; b > 0; --b) {
                    acc.grow(bin_box[b]);
                    right_box[b] = acc;
                }
                acc.clear();
                uint32_t lc = 0;
                for (int b = 0; b < nb - 1; ++b) {
                    acc.grow(bin_box[b]);
                    lc += bin_cnt[b];
                    const uint32_t rc = n - lc;
                    if (lc == 0 || rc == 0) continue;
                    const float cost =
                        (float)lc * acc.half_area() + (float)rc * right_box[b + 1].half_area();
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_bin = b;
                        axis = a;
                        k1 = scale;
                    }
                }
            }
        }
        if (best_bin < 0) {
            // Spatial median of the centroid bounds on their longest axis.
            nb = 2;
            best_bin = 0;
            k1 = 2.f / (cb.hi[axis] - cb.lo[axis]);
        }
        const float lo = cb.lo[axis];
        const int nbm1 = nb - 1;
        auto mid_it = std::partition(prims.begin() + begin, prims.begin() + end, [&](uint32_t p) {
            int b = (int)(k1 * (cent[3 * (size_t)p + axis] - lo));
            b = std::min(std::max(b, 0), nbm1);
            return b <= best_bin;
        });
        uint32_t mid = (uint32_t)(mid_it - prims.begin());
        if (mid == begin || mid == end) {
            // Degenerate partition: order by centroid then primitive id and cut in the middle.
            mid = begin + n / 2;
            std::nth_element(prims.begin() + begin, prims.begin() + mid, prims.begin() + end,
                             [&](uint32_t a, uint32_t b) {
                                 const float ca = cent[3 * (size_t)a + axis],
                                             cbv = cent[3 * (size_t)b + axis];
                                 return ca < cbv || (ca == cbv && a < b);
                             });
        }
        return mid;
    }

    // Builds the subtree over prims[begin,end) into slots [node, node + 2n-1).
    void build(uint32_t node, uint32_t begin, uint32_t end) {
        const uint32_t n = end - begin;
        if (n == 1) {
            emit_leaf(node, prims[begin]);
            return;
        }
        Aabb nb, cb;
        nb.clear();
        cb.clear();
        for (uint32_t i = begin; i < end; ++i) {
            const uint32_t p = prims[i];
            nb.grow(leaf[p]);
            cb.grow_point(&cent[3 * (size_t)p]);
        }
        pt_bbox b;
        for (int k = 0; k < 3; ++k) {
            b.pmin[k] = nb.lo[k];
            b.pmax[k] = nb.hi[k];
        }
        b.pmin[3] = b.pmax[3] = 0.f;
        const uint32_t mid = split(begin, end, cb);
        const uint32_t nl = mid - begin;
        const uint32_t right_node = node + 2 * nl;  // left subtree: 2*nl-1 slots after `node`
        tree->bounds[node] = b;
        tree->right[node] = (int32_t)right_node;
        tree->prim[node] = -1;

        bool spawned = false;
        std::thread worker;
        if (n >= kSpawnThreshold && free_threads.load(std::memory_order_relaxed) > 0) {
            if (free_threads.fetch_sub(1) > 0) {
                spawned = true;
                worker = std::thread([this, right_node, mid, end] {
                    build(right_node, mid, end);
                    free_threads.fetch_add(1);
                });
            } else {
                free_threads.fetch_add(1);
            }
        }
        build(node + 1, begin, mid);
        if (spawned)
            worker.join();
        else
            build(right_node, mid, end);
    }
};

} // namespace

void inflate_leaf_bounds(pt_bbox *boxes, uint32_t n) {
    const float kBoundsScale = 1.f + 1e-4f;
    for (uint32_t i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) {
            const float c = 0.5f * (boxes[i].pmax[k] + boxes[i].pmin[k]);
            const float lo = c + (boxes[i].pmin[k] - c) * kBoundsScale;
            const float hi = c + (boxes[i].pmax[k] - c) * kBoundsScale;
            boxes[i].pmin[k] = lo;
            boxes[i].pmax[k] = hi;
        }
    }
}

void mesh_leaf_bounds(const float *vtx, const int32_t *idx, uint32_t n_tri, pt_bbox *out) {
    for (uint32_t t = 0; t < n_tri; ++t) {
        pt_bbox b;
        for (int k = 0; k < 3; ++k) {
            b.pmin[k] = FLT_MAX;
            b.pmax[k] = -FLT_MAX;
        }
        b.pmin[3] = b.pmax[3] = 0.f;
        for (int j = 0; j < 3; ++j) {
            const float *p = vtx + 3 * (size_t)idx[3 * (size_t)t + j];
            for (int k = 0; k < 3; ++k) {
                b.pmin[k] = std::fmin(b.pmin[k], p[k]);
                b.pmax[k] = std::fmax(b.pmax[k], p[k]);
            }
        }
        out[t] = b;
    }
}

void build_tree(const pt_bbox *leaf_bounds, uint32_t n, const pt_bvh_options &opt, BvhTree &out) {
    if (n == 0) throw std::runtime_error("pt_bvh_build: no primitives");
    if (n > (1u << 30)) throw std::runtime_error("pt_bvh_build: too many primitives");
    if (opt.use_splits)
        throw std::runtime_error(
            "pt_bvh_build: bvh.sah.use_splits (spatial splits) is not implemented");
    if (opt.num_bins < 2 || opt.num_bins > 4096)
        throw std::runtime_error("pt_bvh_build: bvh.sah.num_bins out of range [2,4096]");
    for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k)
            if (!(leaf_bounds[i].pmin[k] <= leaf_bounds[i].pmax[k]) ||
                !std::isfinite(leaf_bounds[i].pmin[k]) || !std::isfinite(leaf_bounds[i].pmax[k]))
                throw std::runtime_error("pt_bvh_build: leaf box " + std::to_string(i) +
                                         " is empty, inverted or not finite");

    const size_t n_nodes = 2 * (size_t)n - 1;
    out.bounds.assign(n_nodes, pt_bbox{});
    out.right.assign(n_nodes, -1);
    out.prim.assign(n_nodes, -1);
    out.n_prims = n;

    Builder b;
    b.leaf = leaf_bounds;
    b.tree = &out;
    b.use_sah = opt.use_sah != 0;
    b.num_bins = opt.num_bins;
    b.traversal_cost = opt.traversal_cost;
    b.cent.resize(3 * (size_t)n);
    b.prims.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        b.prims[i] = i;
        for (int k = 0; k < 3; ++k)
            b.cent[3 * (size_t)i + k] = 0.5f * (leaf_bounds[i].pmax[k] + leaf_bounds[i].pmin[k]);
    }
    int threads = opt.num_threads > 0 ? opt.num_threads : (int)std::thread::hardware_concurrency();
    threads = std::max(1, std::min(threads, 64));
    b.free_threads.store(threads - 1);
    b.build(0, 0, n);
}

bool set_option(pt_bvh_options &opt, const std::string &key, const std::string &value,
                std::string &err) {
    auto as_float = [&](float &dst) {
        char *end = nullptr;
        const float v = std::strtof(value.c_str(), &end);
        if (end == value.c_str() || !std::isfinite(v)) {
            err = "option '" + key + "': '" + value + "' is not a number";
            return false;
        }
        dst = v;
        return true;
    };
    float f = 0.f;
    if (key == "bvh.builder") {
        opt.use_sah = (value == "sah") ? 1 : 0;  // any other value selects the median builder
        return true;
    } else if (key == "bvh.sah.use_splits") {
        if (!as_float(f)) return false;
        opt.use_splits = f > 0.f ? 1 : 0;
    } else if (key == "bvh.sah.max_split_depth") {
        if (!as_float(f)) return false;
        opt.max_split_depth = (int)f;
    } else if (key == "bvh.sah.min_overlap") {
        if (!as_float(f)) return false;
        opt.min_overlap = f;
    } else if (key == "bvh.sah.traversal_cost") {
        if (!as_float(f)) return false;
        opt.traversal_cost = f;
    } else if (key == "bvh.sah.extra_node_budget") {
        if (!as_float(f)) return false;
        opt.extra_node_budget = f;
    } else if (key == "bvh.sah.num_bins") {
        if (!as_float(f)) return false;
        opt.num_bins = (int)f;
    } else if (key == "mipt.link_encoding") {
        if (value == "float") opt.link_encoding = PT_LINKS_FLOAT_VALUE;
        else if (value == "int") opt.link_encoding = PT_LINKS_INT_BITS;
        else { err = "option 'mipt.link_encoding' must be 'float' or 'int'"; return false; }
    } else if (key == "mipt.num_threads") {
        if (!as_float(f)) return false;
        opt.num_threads = (int)f;
    } else {
        err = "unknown option '" + key + "'";
        return false;
    }
    return true;
}

} // namespace mipt
