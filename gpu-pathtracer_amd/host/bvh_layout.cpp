// Skip-link linearizer, its inverse (with validation) and the 64-byte
// two-child re-layout read by the ordered GPU traversal.
#include "bvh.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <stdexcept>
#include <string>

namespace mipt {

namespace {

inline float encode_index(int64_t v, uint32_t enc) {
    if (enc == PT_LINKS_INT_BITS) {
        const int32_t i = (int32_t)v;
        float f;
        std::memcpy(&f, &i, 4);
        return f;
    }
    return (float)v;
}

inline int64_t decode_index(float f, uint32_t enc) {
    if (enc == PT_LINKS_INT_BITS) {
        int32_t i;
        std::memcpy(&i, &f, 4);
        return i;
    }
    // The shader reads the field with int(x) (shaders/RadeonRays/bvh.glslh:153,443).
    if (!(f >= -1.f) || !(f < 2147483648.f) || f != std::floor(f))
        throw std::runtime_error("skip-link array: field " + std::to_string(f) +
                                 " is not an integer-valued float >= -1");
    return (int64_t)f;
}

} // namespace

// Skip links as PlainBvhTranslator::Process computes them
// (src/bvh/RadeonRays/plain_bvh_translator.cpp:59-69): the root links to -1;
// for an internal node i with right child r, node i+1 links to r and r
// inherits i's link.  Leaf payload = primitive index (:88-89), -1 on internal
// nodes (:93).
void linearize_skip_links(const BvhTree &t, uint32_t enc, std::vector<pt_bbox> &out) {
    const size_t n = t.size();
    if (enc != PT_LINKS_FLOAT_VALUE && enc != PT_LINKS_INT_BITS)
        throw std::runtime_error("unknown link encoding");
    if (enc == PT_LINKS_FLOAT_VALUE && n > (1u << 24))
        throw std::runtime_error(
            "tree has " + std::to_string(n) +
            " nodes: indices above 2^24 are not exact as float values; use "
            "PT_LINKS_INT_BITS (option mipt.link_encoding=int)");
    out.resize(n);
    std::vector<int32_t> link(n, -1);
    for (size_t i = 0; i < n; ++i) {
        const int32_t r = t.right[i];
        if (r >= 0) {
            link[i + 1] = r;
            link[(size_t)r] = link[i];
        }
    }
    for (size_t i = 0; i < n; ++i) {
        pt_bbox b = t.bounds[i];
        b.pmin[3] = t.right[i] < 0 ? encode_index(t.prim[i], enc) : encode_index(-1, enc);
        b.pmax[3] = encode_index(link[i], enc);
        out[i] = b;
    }
}

void tree_from_skip_links(const pt_bbox *nodes, uint32_t n_nodes, uint32_t enc, uint32_t n_tri,
                          BvhTree &out) {
    if (enc != PT_LINKS_FLOAT_VALUE && enc != PT_LINKS_INT_BITS)
        throw std::runtime_error("unknown link encoding");
    if (n_nodes == 0) throw std::runtime_error("skip-link array is empty");
    if ((n_nodes & 1u) == 0)
        throw std::runtime_error("skip-link array: a binary tree with one primitive per leaf "
                                 "has an odd node count, got " + std::to_string(n_nodes));
    out.bounds.assign(nodes, nodes + n_nodes);
    out.right.assign(n_nodes, -1);
    out.prim.assign(n_nodes, -1);
    out.n_prims = (n_nodes + 1) / 2;

    std::vector<int64_t> link(n_nodes);
    for (uint32_t i = 0; i < n_nodes; ++i) {
        for (int k = 0; k < 3; ++k)
            if (std::isnan(nodes[i].pmin[k]) || std::isnan(nodes[i].pmax[k]))
                throw std::runtime_error("skip-link array: node " + std::to_string(i) +
                                         " has a NaN bound");
        const int64_t payload = decode_index(nodes[i].pmin[3], enc);
        link[i] = decode_index(nodes[i].pmax[3], enc);
        if (link[i] != -1 && (link[i] <= (int64_t)i || link[i] >= (int64_t)n_nodes))
            throw std::runtime_error("skip-link array: node " + std::to_string(i) +
                                     " links to " + std::to_string(link[i]) +
                                     ", expected -1 or a later node");
        if (payload != -1) {
            if (payload < 0 || (n_tri && payload >= (int64_t)n_tri))
                throw std::runtime_error("skip-link array: leaf " + std::to_string(i) +
                                         " references triangle " + std::to_string(payload) +
                                         " outside the index buffer");
            out.prim[i] = (int32_t)payload;
        }
    }
    if (link[0] != -1) throw std::runtime_error("skip-link array: the root must link to -1");

    // Pre-order walk with an explicit stack of pending right children: an
    // internal node i has its left child at i+1, whose link is the right child.
    std::vector<int64_t> pending_link;   // link each open subtree must end on
    std::vector<uint32_t> pending_right;
    uint32_t i = 0;
    int64_t expect_link = -1;
    uint32_t leaves = 0;
    for (;;) {
        if (link[i] != expect_link)
            throw std::runtime_error("skip-link array: node " + std::to_string(i) + " links to " +
                                     std::to_string(link[i]) + ", expected " +
                                     std::to_string(expect_link));
        if (out.prim[i] >= 0) {
            ++leaves;
            if (pending_right.empty()) {
                if (i + 1 != n_nodes)
                    throw std::runtime_error("skip-link array: nodes after the end of the tree");
                break;
            }
            const uint32_t r = pending_right.back();
            if (r != i + 1)
                throw std::runtime_error("skip-link array: subtree of node ending at " +
                                         std::to_string(i) + " is not contiguous");
            expect_link = pending_link.back();
            pending_right.pop_back();
            pending_link.pop_back();
            i = r;
        } else {
            if (i + 1 >= n_nodes)
                throw std::runtime_error("skip-link array: internal node " + std::to_string(i) +
                                         " has no left child");
            const int64_t r = link[i + 1];
            if (r <= (int64_t)i + 1 || r >= (int64_t)n_nodes)
                throw std::runtime_error("skip-link array: internal node " + std::to_string(i) +
                                         " has no valid right child");
            out.right[i] = (int32_t)r;
            pending_right.push_back((uint32_t)r);
            pending_link.push_back(expect_link);
            expect_link = r;
            i = i + 1;
        }
    }
    if (leaves != out.n_prims)
        throw std::runtime_error("skip-link array: leaf count does not match node count");
    // A child's box must lie within its parent's.  The ordered traversal visits another set of
    // nodes than the fixed-order loop and returns the same hit only because a leaf whose own box
    // passes the slab test is reached in either order, which takes boxes that nest.
    for (uint32_t p = 0; p < n_nodes; ++p) {
        if (out.right[p] < 0) continue;
        for (const uint32_t c : {p + 1, (uint32_t)out.right[p]})
            for (int k = 0; k < 3; ++k)
                if (nodes[c].pmin[k] < nodes[p].pmin[k] || nodes[c].pmax[k] > nodes[p].pmax[k])
                    throw std::runtime_error("skip-link array: the box of node " + std::to_string(c) +
                                             " is not inside the box of its parent, node " +
                                             std::to_string(p));
    }
}

void build_wide_layout(const BvhTree &t, const float *vtx, uint32_t n_vtx, const int32_t *idx,
                       uint32_t n_tri, WideBvh &out) {
    const size_t n = t.size();
    const uint32_t n_leaves = t.n_prims;
    const uint32_t n_inner = n_leaves > 1 ? n_leaves - 1 : 1;
    out.nodes.assign(n_inner, WideNode{});
    out.tris.assign(n_leaves, WideTriRecord{});
    out.max_depth = 0;

    // Inner nodes and leaves are numbered separately, each in pre-order.
    std::vector<int32_t> slot(n);  // inner index (>= 0) or ~leaf slot (< 0)
    {
        int32_t ni = 0, nl = 0;
        for (size_t i = 0; i < n; ++i) slot[i] = t.right[i] >= 0 ? ni++ : ~(nl++);
    }
    // Experiment knob (tools/exp_node_order.py): another order of the two-child nodes in memory.
    // MIPT_WIDE_ORDER = random | bfs | treelet:<nodes>.  The root stays node 0; no result depends
    // on the order.
    if (const char *order = std::getenv("MIPT_WIDE_ORDER")) {
        const std::string kind(order);
        std::vector<size_t> seq;                 // tree nodes (inner only) in their new order
        seq.reserve(n_inner);
        if (kind == "bfs" || kind == "random") {
            std::vector<size_t> q{0};
            for (size_t h = 0; h < q.size(); ++h) {
                const size_t i = q[h];
                if (t.right[i] < 0) continue;
                seq.push_back(i);
                q.push_back(i + 1);
                q.push_back((size_t)t.right[i]);
            }
            if (kind == "random") {
                uint64_t state = 0x9E3779B97F4A7C15ull;
                for (size_t k = seq.size(); k > 2; --k) {     // Fisher-Yates over seq[1..]
                    state = state * 6364136223846793005ull + 1442695040888963407ull;
                    const size_t j = 1 + (size_t)((state >> 33) % (k - 1));
                    std::swap(seq[k - 1], seq[j]);
                }
            }
        } else if (kind.rfind("treelet:", 0) == 0) {
            const size_t cap = std::max<size_t>(1, std::strtoull(kind.c_str() + 8, nullptr, 10));
            std::vector<size_t> roots{0};        // treelet roots still to be laid out (a stack)
            std::vector<size_t> q;
            while (!roots.empty()) {
                const size_t r = roots.back();
                roots.pop_back();
                if (t.right[r] < 0) continue;
                q.assign(1, r);                   // breadth first from r, at most cap inner nodes
                size_t taken = 0;
                for (size_t h = 0; h < q.size(); ++h) {
                    const size_t i = q[h];
                    if (t.right[i] < 0) continue;
                    if (taken == cap) {
                        roots.push_back(i);
                        continue;
                    }
                    seq.push_back(i);
                    ++taken;
                    q.push_back(i + 1);
                    q.push_back((size_t)t.right[i]);
                }
            }
        }
        if (seq.size() == n_inner && n_leaves > 1)
            for (size_t k = 0; k < seq.size(); ++k) slot[seq[k]] = (int32_t)k;
    }
    auto fill_tri = [&](size_t node) {
        const int32_t p = t.prim[node];
        if (p < 0 || (uint32_t)p >= n_tri) throw std::runtime_error("leaf triangle out of range");
        TriRecord &r = out.tris[(size_t)(~slot[node])];
        const float *v[3];
        for (int j = 0; j < 3; ++j) {
            const int32_t vi = idx[3 * (size_t)p + j];
            if (vi < 0 || (uint32_t)vi >= n_vtx)
                throw std::runtime_error("triangle " + std::to_string(p) +
                                         " references vertex " + std::to_string(vi) +
                                         " outside the vertex buffer");
            v[j] = vtx + 3 * (size_t)vi;
        }
        r.v0[0] = v[0][0]; r.v0[1] = v[0][1]; r.v0[2] = v[0][2];
        r.v1x = v[1][0]; r.v1yz[0] = v[1][1]; r.v1yz[1] = v[1][2];
        r.v2xy[0] = v[2][0]; r.v2xy[1] = v[2][1]; r.v2z = v[2][2];
        r.primid = p;
        r.pad[0] = r.pad[1] = 0;
    };
    auto set_child = [&](WideNode &w, int c, size_t node) {
        const pt_bbox &b = t.bounds[node];
        float *xy = c == 0 ? w.c0xy : w.c1xy;
        xy[0] = b.pmin[0]; xy[1] = b.pmax[0]; xy[2] = b.pmin[1]; xy[3] = b.pmax[1];
        w.cz[2 * c + 0] = b.pmin[2];
        w.cz[2 * c + 1] = b.pmax[2];
        // a leaf is named by the position of its record in 16-byte units, complemented
        w.child[c] = slot[node] >= 0 ? slot[node] : ~(PT_TRI_RECORD_16B * ~slot[node]);
    };

    if (n == 1) {
        // Single triangle: the root's second child is a record whose three vertices
        // coincide.  Its edge functions are exactly zero for every finite ray, so the
        // triangle test rejects it (det == 0) and it can never become a candidate.  (An
        // "inverted" box would not do: the slab test takes min/max of the two planes, so a
        // box with min > max looks unbounded, not empty.)
        WideNode &w = out.nodes[0];
        set_child(w, 0, 0);
        set_child(w, 1, 0);
        w.child[1] = ~PT_TRI_RECORD_16B;
        w.child[2] = w.child[3] = 0;
        out.tris.assign(2, WideTriRecord{});
        fill_tri(0);
        TriRecord &r = out.tris[1];
        r = out.tris[0];
        r.v1x = r.v0[0]; r.v1yz[0] = r.v0[1]; r.v1yz[1] = r.v0[2];
        r.v2xy[0] = r.v0[0]; r.v2xy[1] = r.v0[1]; r.v2z = r.v0[2];
        return;
    }
    for (size_t i = 0; i < n; ++i) {
        if (t.right[i] >= 0) {
            WideNode &w = out.nodes[(size_t)slot[i]];
            set_child(w, 0, i + 1);
            set_child(w, 1, (size_t)t.right[i]);
            w.child[2] = w.child[3] = 0;
        } else {
            fill_tri(i);
        }
    }
    // Depth, for sizing checks and statistics.
    std::vector<uint32_t> depth(n, 0);
    for (size_t i = 0; i < n; ++i) {
        if (t.right[i] >= 0) {
            depth[i + 1] = depth[i] + 1;
            depth[(size_t)t.right[i]] = depth[i] + 1;
        }
        if (depth[i] > out.max_depth) out.max_depth = depth[i];
    }
}

// Two-child nodes of a top-level tree: children numbered like build_wide_layout's inner nodes,
// a leaf child is the complement of its payload (an instance index).  A single-leaf tree gets
// one node naming that leaf twice.
void build_wide_top(const BvhTree &t, std::vector<WideNode> &out) {
    const size_t n = t.size();
    out.assign(t.n_prims > 1 ? t.n_prims - 1 : 1, WideNode{});
    std::vector<int32_t> slot(n);
    {
        int32_t ni = 0;
        for (size_t i = 0; i < n; ++i) slot[i] = t.right[i] >= 0 ? ni++ : ~t.prim[i];
    }
    auto set_child = [&](WideNode &w, int c, size_t node) {
        const pt_bbox &b = t.bounds[node];
        float *xy = c == 0 ? w.c0xy : w.c1xy;
        xy[0] = b.pmin[0]; xy[1] = b.pmax[0]; xy[2] = b.pmin[1]; xy[3] = b.pmax[1];
        w.cz[2 * c + 0] = b.pmin[2];
        w.cz[2 * c + 1] = b.pmax[2];
        w.child[c] = slot[node];
    };
    if (n == 1) {
        set_child(out[0], 0, 0);
        set_child(out[0], 1, 0);
        return;
    }
    for (size_t i = 0; i < n; ++i)
        if (t.right[i] >= 0) {
            WideNode &w = out[(size_t)slot[i]];
            set_child(w, 0, i + 1);
            set_child(w, 1, (size_t)t.right[i]);
        }
}

} // namespace mipt
