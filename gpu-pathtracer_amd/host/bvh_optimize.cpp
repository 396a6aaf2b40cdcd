// Insertion-based optimisation of a finished binary BVH (after Bittner, Hapala, Havran 2013 and
// the per-node form of Meister & Bittner 2018): every subtree in turn is taken out of the tree
// and put back where it adds the least surface area, found by a best-first search with the
// area already added on the way down as the bound.  The leaf boxes are never touched and inner
// boxes stay exact unions of their children, so every query returns what it returned before;
// only the number of boxes a ray meets changes.
//
// The reference's builder (RadeonRays::Bvh / SplitBvh, src/bvh/bvh_builder.cpp:42-47) has no
// such pass; this is an extra of this project behind the option "mipt.optimize_passes".
#include "bvh.h"

#include <algorithm>
#include <cmath>
#include <queue>

namespace mipt {

namespace {

struct Box {
    float lo[3], hi[3];
};
inline Box unite(const Box &a, const Box &b) {
    Box r;
    for (int k = 0; k < 3; ++k) {
        r.lo[k] = std::fmin(a.lo[k], b.lo[k]);
        r.hi[k] = std::fmax(a.hi[k], b.hi[k]);
    }
    return r;
}
inline double area(const Box &b) {
    const double x = (double)b.hi[0] - b.lo[0], y = (double)b.hi[1] - b.lo[1], z = (double)b.hi[2] - b.lo[2];
    return x * y + y * z + z * x;
}

struct Node {
    Box box;
    int32_t parent, left, right;   // left < 0: leaf
    int32_t prim;
};

struct Optimizer {
    std::vector<Node> n;
    int32_t root = 0;

    void refit_up(int32_t i) {
        while (i >= 0) {
            n[i].box = unite(n[n[i].left].box, n[n[i].right].box);
            i = n[i].parent;
        }
    }

    // Where to hang subtree x (already detached) so that the tree's total area grows least.
    int32_t best_place(int32_t x) const {
        const Box &bx = n[x].box;
        const double ax = area(bx);
        struct Item {
            double induced;
            int32_t node;
            bool operator<(const Item &o) const { return induced > o.induced; }
        };
        std::priority_queue<Item> open;
        open.push({0.0, root});
        double best = INFINITY;
        int32_t best_node = root;
        while (!open.empty()) {
            const Item it = open.top();
            open.pop();
            if (it.induced + ax >= best) break;        // nothing below can beat the best found
            const Node &y = n[it.node];
            const double direct = area(unite(y.box, bx));
            const double total = it.induced + direct;
            if (total < best) {
                best = total;
                best_node = it.node;
            }
            if (y.left >= 0) {
                const double down = it.induced + (direct - area(y.box));
                if (down + ax < best) {
                    open.push({down, y.left});
                    open.push({down, y.right});
                }
            }
        }
        return best_node;
    }

    // Takes x out of the tree (its parent p goes with it, the sibling moves up) and hangs the
    // pair (y, x) under p at the best place y.  Returns false if x cannot be moved.
    bool reinsert(int32_t x) {
        const int32_t p = n[x].parent;
        if (p < 0 || p == root) return false;          // the root's children stay
        const int32_t g = n[p].parent;
        const int32_t s = n[p].left == x ? n[p].right : n[p].left;
        (n[g].left == p ? n[g].left : n[g].right) = s;
        n[s].parent = g;
        refit_up(g);
        const int32_t y = best_place(x);
        const int32_t yp = n[y].parent;
        n[p].left = y;
        n[p].right = x;
        n[p].parent = yp;
        n[y].parent = p;
        n[x].parent = p;
        if (yp < 0) root = p;
        else (n[yp].left == y ? n[yp].left : n[yp].right) = p;
        refit_up(p);
        return true;
    }
};

} // namespace

double tree_area_cost(const BvhTree &t) {
    double sum = 0.0, root = 0.0;
    for (size_t i = 0; i < t.size(); ++i) {
        Box b;
        for (int k = 0; k < 3; ++k) { b.lo[k] = t.bounds[i].pmin[k]; b.hi[k] = t.bounds[i].pmax[k]; }
        if (i == 0) root = area(b);
        if (t.right[i] >= 0) sum += area(b);
    }
    return root > 0.0 ? sum / root : 0.0;
}

void optimize_tree(BvhTree &t, int passes) {
    const size_t count = t.size();
    if (passes <= 0 || count < 7) return;
    Optimizer o;
    o.n.resize(count);
    for (size_t i = 0; i < count; ++i) {
        Node &d = o.n[i];
        for (int k = 0; k < 3; ++k) { d.box.lo[k] = t.bounds[i].pmin[k]; d.box.hi[k] = t.bounds[i].pmax[k]; }
        d.prim = t.prim[i];
        d.left = t.right[i] >= 0 ? (int32_t)i + 1 : -1;
        d.right = t.right[i];
        d.parent = -1;
    }
    for (size_t i = 0; i < count; ++i)
        if (o.n[i].left >= 0) {
            o.n[o.n[i].left].parent = (int32_t)i;
            o.n[o.n[i].right].parent = (int32_t)i;
        }
    std::vector<int32_t> order(count);
    std::vector<double> key(count);
    for (int pass = 0; pass < passes; ++pass) {
        // big boxes first: moving them pays most, and it settles the top before the bottom
        for (size_t i = 0; i < count; ++i) {
            order[i] = (int32_t)i;
            key[i] = area(o.n[i].box);
        }
        std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
            return key[a] != key[b] ? key[a] > key[b] : a < b;
        });
        for (int32_t x : order) o.reinsert(x);
    }
    // back to the pre-order arrays (left child = next node)
    BvhTree out;
    out.bounds.resize(count);
    out.right.assign(count, -1);
    out.prim.assign(count, -1);
    out.n_prims = t.n_prims;
    std::vector<int32_t> stack{o.root}, pos(count, -1);
    std::vector<int32_t> pending;                      // inner nodes whose right child is not placed yet
    int32_t next = 0;
    while (!stack.empty()) {
        const int32_t i = stack.back();
        stack.pop_back();
        const int32_t at = next++;
        pos[i] = at;
        const Node &s = o.n[i];
        pt_bbox b{};
        for (int k = 0; k < 3; ++k) { b.pmin[k] = s.box.lo[k]; b.pmax[k] = s.box.hi[k]; }
        out.bounds[at] = b;
        if (s.left >= 0) {
            stack.push_back(s.right);
            stack.push_back(s.left);
        } else {
            out.prim[at] = s.prim;
        }
    }
    for (size_t i = 0; i < count; ++i)
        if (o.n[i].left >= 0) out.right[pos[i]] = pos[o.n[i].right];
    t = std::move(out);
}

} // namespace mipt
