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

// ---- spatial splits -----------------------------------------------------------------------
// Stands in for RadeonRays::SplitBvh, which the reference selects with bvh.sah.use_splits
// (src/bvh/bvh_builder.cpp:38-44) and which, like Bvh, is external to the checkout.  The
// builder is handed boxes only (bvh->Build(bounds, n), :62), so a "split" cuts a primitive's
// box at a plane into two references to the same primitive (Stich et al. 2009, restricted to
// boxes): the tree then has more leaves than primitives, each leaf still holding exactly one
// primitive, as the shader requires.  Options as passed at :43-44: a spatial split is tried
// down to max_split_depth, when the object split's children overlap by more than min_overlap
// of the root's area, while fewer than extra_node_budget * n extra references exist.
struct SplitBuilder {
    struct Ref {
        Aabb box;
        uint32_t prim;
    };
    BvhTree *tree;
    int num_bins;
    int max_split_depth;
    float min_overlap;
    float root_area = 0.f;
    size_t extra_left = 0;      // references that may still be added

    static float center(const Aabb &b, int k) { return 0.5f * (b.lo[k] + b.hi[k]); }

    uint32_t new_node() {
        tree->bounds.push_back(pt_bbox{});
        tree->right.push_back(-1);
        tree->prim.push_back(-1);
        return (uint32_t)(tree->bounds.size() - 1);
    }
    void set_box(uint32_t node, const Aabb &b) {
        pt_bbox &o = tree->bounds[node];
        for (int k = 0; k < 3; ++k) {
            o.pmin[k] = b.lo[k];
            o.pmax[k] = b.hi[k];
        }
        o.pmin[3] = o.pmax[3] = 0.f;
    }

    struct ObjectSplit { float cost = FLT_MAX; int axis = -1; int bin = -1; float scale = 0.f; float lo = 0.f; float overlap = 0.f; };
    struct SpatialSplit { float cost = FLT_MAX; int axis = -1; float plane = 0.f; uint32_t nl = 0, nr = 0; };

    ObjectSplit find_object_split(const std::vector<Ref> &refs, const Aabb &cb) const {
        ObjectSplit best;
        const uint32_t n = (uint32_t)refs.size();
        const int nb = num_bins;
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
            for (const Ref &r : refs) {
                int b = (int)(scale * (center(r.box, a) - cb.lo[a]));
                b = std::min(std::max(b, 0), nb - 1);
                bin_cnt[b]++;
                bin_box[b].grow(r.box);
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
                const Aabb &rb = right_box[b + 1];
                const float cost = (float)lc * acc.half_area() + (float)rc * rb.half_area();
                if (cost < best.cost) {
                    best.cost = cost;
                    best.axis = a;
                    best.bin = b;
                    best.scale = scale;
                    best.lo = cb.lo[a];
                    Aabb ov;                                  // intersection of the two sides
                    bool empty = false;
                    for (int k = 0; k < 3; ++k) {
                        ov.lo[k] = std::fmax(acc.lo[k], rb.lo[k]);
                        ov.hi[k] = std::fmin(acc.hi[k], rb.hi[k]);
                        if (!(ov.lo[k] <= ov.hi[k])) empty = true;
                    }
                    best.overlap = empty ? 0.f : ov.half_area();
                }
            }
        }
        return best;
    }

    SpatialSplit find_spatial_split(const std::vector<Ref> &refs, const Aabb &nbx) const {
        SpatialSplit best;
        const uint32_t n = (uint32_t)refs.size();
        const int nb = num_bins;
        std::vector<Aabb> bin_box((size_t)nb), right_box((size_t)nb);
        std::vector<uint32_t> enter((size_t)nb), leave((size_t)nb);
        for (int a = 0; a < 3; ++a) {
            const float lo = nbx.lo[a], ext = nbx.hi[a] - lo;
            if (!(ext > 0.f)) continue;
            const float scale = (float)nb / ext, width = ext / (float)nb;
            for (int b = 0; b < nb; ++b) {
                bin_box[b].clear();
                enter[b] = leave[b] = 0;
            }
            for (const Ref &r : refs) {
                int b0 = std::min(std::max((int)(scale * (r.box.lo[a] - lo)), 0), nb - 1);
                int b1 = std::min(std::max((int)(scale * (r.box.hi[a] - lo)), b0), nb - 1);
                enter[b0]++;
                leave[b1]++;
                for (int b = b0; b <= b1; ++b) {          // the box clipped to the bin's slab
                    Aabb c = r.box;
                    if (b > b0) c.lo[a] = std::fmax(c.lo[a], lo + width * (float)b);
                    if (b < b1) c.hi[a] = std::fmin(c.hi[a], lo + width * (float)(b + 1));
                    bin_box[b].grow(c);
                }
            }
            Aabb acc;
            acc.clear();
            for (int b = nb - 1; b > 0; --b) {
                acc.grow(bin_box[b]);
                right_box[b] = acc;
            }
            acc.clear();
            uint32_t nl = 0, nr = n;
            for (int b = 0; b < nb - 1; ++b) {
                acc.grow(bin_box[b]);
                nl += enter[b];
                nr -= leave[b];
                // a plane that leaves every reference on one side makes no progress
                if (nl == 0 || nr == 0 || nl >= n || nr >= n) continue;
                const float cost = (float)nl * acc.half_area() + (float)nr * right_box[b + 1].half_area();
                if (cost < best.cost) {
                    best.cost = cost;
                    best.axis = a;
                    best.plane = lo + width * (float)(b + 1);
                    best.nl = nl;
                    best.nr = nr;
                }
            }
        }
        return best;
    }

    void build(std::vector<Ref> &refs, int depth) {
        const uint32_t node = new_node();
        const uint32_t n = (uint32_t)refs.size();
        Aabb nbx, cb;
        nbx.clear();
        cb.clear();
        for (const Ref &r : refs) {
            nbx.grow(r.box);
            const float c[3] = {center(r.box, 0), center(r.box, 1), center(r.box, 2)};
            cb.grow_point(c);
        }
        set_box(node, nbx);
        if (n == 1) {
            tree->prim[node] = (int32_t)refs[0].prim;
            return;
        }
        const ObjectSplit os = n > 2 ? find_object_split(refs, cb) : ObjectSplit{};
        SpatialSplit ss;
        if (depth < max_split_depth && extra_left > 0 && os.axis >= 0 && root_area > 0.f &&
            os.overlap / root_area > min_overlap)
            ss = find_spatial_split(refs, nbx);
        std::vector<Ref> left, right;
        bool done = false;
        if (ss.axis >= 0 && ss.cost < os.cost) {
            const int a = ss.axis;
            size_t straddle = 0;
            for (const Ref &r : refs)
                if (r.box.lo[a] < ss.plane && r.box.hi[a] > ss.plane) ++straddle;
            if (straddle <= extra_left) {
                for (const Ref &r : refs) {
                    if (r.box.hi[a] <= ss.plane && r.box.lo[a] < ss.plane) {
                        left.push_back(r);
                    } else if (r.box.lo[a] >= ss.plane) {
                        right.push_back(r);
                    } else {                                   // cut the box at the plane
                        Ref l = r, rr = r;
                        l.box.hi[a] = ss.plane;
                        rr.box.lo[a] = ss.plane;
                        left.push_back(l);
                        right.push_back(rr);
                    }
                }
                if (!left.empty() && !right.empty() && left.size() < n && right.size() < n) {
                    extra_left -= straddle;
                    done = true;
                } else {
                    left.clear();
                    right.clear();
                }
            }
        }
        if (!done && os.axis >= 0) {
            const int nbm1 = num_bins - 1;
            for (const Ref &r : refs) {
                int b = (int)(os.scale * (center(r.box, os.axis) - os.lo));
                b = std::min(std::max(b, 0), nbm1);
                (b <= os.bin ? left : right).push_back(r);
            }
            done = !left.empty() && !right.empty();
            if (!done) {
                left.clear();
                right.clear();
            }
        }
        if (!done) {
            // No usable plane (two references, or coincident centres): order by centre on the
            // longest axis, then by primitive and box, and cut in the middle.
            int axis = 0;
            const float e[3] = {cb.hi[0] - cb.lo[0], cb.hi[1] - cb.lo[1], cb.hi[2] - cb.lo[2]};
            if (e[1] > e[axis]) axis = 1;
            if (e[2] > e[axis]) axis = 2;
            std::stable_sort(refs.begin(), refs.end(), [&](const Ref &x, const Ref &y) {
                const float cx = center(x.box, axis), cy = center(y.box, axis);
                return cx < cy || (cx == cy && x.prim < y.prim);
            });
            left.assign(refs.begin(), refs.begin() + n / 2);
            right.assign(refs.begin() + n / 2, refs.end());
        }
        refs.clear();
        refs.shrink_to_fit();
        build(left, depth + 1);
        left = std::vector<Ref>();
        tree->right[node] = (int32_t)tree->bounds.size();
        build(right, depth + 1);
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
    if (opt.num_bins < 2 || opt.num_bins > 4096)
        throw std::runtime_error("pt_bvh_build: bvh.sah.num_bins out of range [2,4096]");
    for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k)
            if (!(leaf_bounds[i].pmin[k] <= leaf_bounds[i].pmax[k]) ||
                !std::isfinite(leaf_bounds[i].pmin[k]) || !std::isfinite(leaf_bounds[i].pmax[k]))
                throw std::runtime_error("pt_bvh_build: leaf box " + std::to_string(i) +
                                         " is empty, inverted or not finite");

    if (opt.use_splits) {
        if (!(opt.extra_node_budget >= 0.f) || !(opt.min_overlap >= 0.f))
            throw std::runtime_error("pt_bvh_build: bvh.sah.extra_node_budget and "
                                     "bvh.sah.min_overlap must be >= 0");
        out.bounds.clear();
        out.right.clear();
        out.prim.clear();
        SplitBuilder sb;
        sb.tree = &out;
        sb.num_bins = opt.num_bins;
        sb.max_split_depth = opt.max_split_depth;
        sb.min_overlap = opt.min_overlap;
        sb.extra_left = (size_t)((double)opt.extra_node_budget * (double)n);
        std::vector<SplitBuilder::Ref> refs(n);
        Aabb root;
        root.clear();
        for (uint32_t i = 0; i < n; ++i) {
            for (int k = 0; k < 3; ++k) {
                refs[i].box.lo[k] = leaf_bounds[i].pmin[k];
                refs[i].box.hi[k] = leaf_bounds[i].pmax[k];
            }
            refs[i].prim = i;
            root.grow(refs[i].box);
        }
        sb.root_area = root.half_area();
        out.bounds.reserve(2 * (size_t)n + 2 * sb.extra_left);
        out.right.reserve(2 * (size_t)n + 2 * sb.extra_left);
        out.prim.reserve(2 * (size_t)n + 2 * sb.extra_left);
        sb.build(refs, 0);
        out.n_prims = (uint32_t)((out.bounds.size() + 1) / 2);      // leaves
        return;
    }

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
    } else if (key == "mipt.optimize_passes") {
        if (!as_float(f)) return false;
        if (f < 0.f || f > 64.f) { err = "option 'mipt.optimize_passes' must be in [0, 64]"; return false; }
        opt.optimize_passes = (int)f;
    } else {
        err = "unknown option '" + key + "'";
        return false;
    }
    return true;
}

} // namespace mipt
