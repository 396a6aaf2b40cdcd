// Minimal glTF 2.0 reader (.gltf with external or base64 buffers, and binary .glb).
//
// The live application of the reference loads its scene with Granite's glTF loader
// (scene_loader_.load_scene("../assets/lucy/lucy_watertight.glb"), src/main_granite.cpp:265-268;
// Granite is external to the checkout) and hands the positions and 16- or 32-bit indices of
// the first mesh, untransformed, to build_mesh_bvh (:286-298, :115-172).  This reader
// supplies exactly that with PT_GLTF_FIRST_MESH_ONLY; without the flag it flattens the whole
// default scene -- every triangle primitive of every node, positions taken to world space by
// the node hierarchy -- which is the "all meshes" the reference leaves as a TODO (:284-285).
// Materials are reduced to what the shading kernels use: base colour, metallic and roughness
// factors and the emissive factor (textures are ignored).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mipt.h"
#include "../csrc/internal.h"
#include "mesh.h"

namespace {

// ---- JSON ---------------------------------------------------------------------------------
struct Json {
    enum Type { Null, Bool, Num, Str, Arr, Obj } type = Null;
    double num = 0.0;
    bool b = false;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    const Json *get(const char *key) const {
        if (type != Obj) return nullptr;
        for (const auto &kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    double number(const char *key, double dflt) const {
        const Json *v = get(key);
        return v && v->type == Num ? v->num : dflt;
    }
    long index(const char *key) const {             // -1 if absent
        const Json *v = get(key);
        return v && v->type == Num ? (long)v->num : -1;
    }
    size_t size() const { return type == Arr ? arr.size() : 0; }
};

struct JsonParser {
    const char *p, *end;
    int depth = 0;
    [[noreturn]] void fail(const char *what) const { throw std::runtime_error(std::string("JSON: ") + what); }
    void ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    bool lit(const char *s) {
        const size_t n = std::strlen(s);
        if ((size_t)(end - p) >= n && std::memcmp(p, s, n) == 0) {
            p += n;
            return true;
        }
        return false;
    }
    std::string string() {
        if (p >= end || *p != '"') fail("expected a string");
        ++p;
        std::string out;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) fail("unterminated escape");
                switch (*p) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        if (end - p < 5) fail("short \\u escape");
                        unsigned v = 0;
                        for (int k = 1; k <= 4; ++k) {
                            const char c = p[k];
                            v = v * 16 + (c >= '0' && c <= '9' ? c - '0'
                                          : c >= 'a' && c <= 'f' ? c - 'a' + 10
                                          : c >= 'A' && c <= 'F' ? c - 'A' + 10 : 0);
                        }
                        p += 4;
                        if (v < 0x80) out += (char)v;           // names here are ASCII; others
                        else out += '?';                        // only need to stay distinct-ish
                        break;
                    }
                    default: out += *p;
                }
                ++p;
            } else {
                out += *p++;
            }
        }
        if (p >= end) fail("unterminated string");
        ++p;
        return out;
    }
    Json value() {
        if (++depth > 64) fail("nesting too deep");
        ws();
        if (p >= end) fail("unexpected end");
        Json v;
        if (*p == '{') {
            ++p;
            v.type = Json::Obj;
            ws();
            if (p < end && *p == '}') { ++p; --depth; return v; }
            for (;;) {
                ws();
                std::string k = string();
                ws();
                if (p >= end || *p != ':') fail("expected ':'");
                ++p;
                v.obj.emplace_back(std::move(k), value());
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                fail("expected ',' or '}'");
            }
        } else if (*p == '[') {
            ++p;
            v.type = Json::Arr;
            ws();
            if (p < end && *p == ']') { ++p; --depth; return v; }
            for (;;) {
                v.arr.push_back(value());
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; break; }
                fail("expected ',' or ']'");
            }
        } else if (*p == '"') {
            v.type = Json::Str;
            v.str = string();
        } else if (lit("true")) {
            v.type = Json::Bool;
            v.b = true;
        } else if (lit("false")) {
            v.type = Json::Bool;
        } else if (lit("null")) {
        } else {
            char *e = nullptr;
            const std::string tmp(p, (size_t)std::min<ptrdiff_t>(end - p, 64));
            v.num = std::strtod(tmp.c_str(), &e);
            if (e == tmp.c_str()) fail("unexpected character");
            p += e - tmp.c_str();
            v.type = Json::Num;
        }
        --depth;
        return v;
    }
};

// ---- bytes ----------------------------------------------------------------------------------
bool read_file(const std::string &path, std::vector<unsigned char> &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    if (n < 0) return false;
    f.seekg(0);
    out.resize((size_t)n);
    if (n) f.read(reinterpret_cast<char *>(out.data()), n);
    return (bool)f;
}

std::vector<unsigned char> base64(const std::string &s, size_t from) {
    std::vector<unsigned char> out;
    unsigned acc = 0;
    int bits = 0;
    for (size_t i = from; i < s.size(); ++i) {
        const char c = s[i];
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A';
        else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
        else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62;
        else if (c == '/' || c == '_') v = 63;
        else continue;                                 // '=', whitespace
        acc = (acc << 6) | (unsigned)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out.push_back((unsigned char)((acc >> bits) & 0xFF));
        }
    }
    return out;
}

uint32_t rd32(const unsigned char *p) { uint32_t v; std::memcpy(&v, p, 4); return v; }

// ---- 4x4 matrices, column-major as in glTF ----------------------------------------------------
struct Mat4 { double m[16]; };
Mat4 identity() {
    Mat4 r{};
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0;
    return r;
}
Mat4 mul(const Mat4 &a, const Mat4 &b) {
    Mat4 r{};
    for (int c = 0; c < 4; ++c)
        for (int rr = 0; rr < 4; ++rr) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += a.m[k * 4 + rr] * b.m[c * 4 + k];
            r.m[c * 4 + rr] = s;
        }
    return r;
}
Mat4 node_matrix(const Json &node) {
    if (const Json *mj = node.get("matrix")) {
        if (mj->size() != 16) throw std::runtime_error("node.matrix must have 16 numbers");
        Mat4 r{};
        for (int k = 0; k < 16; ++k) r.m[k] = mj->arr[k].num;
        return r;
    }
    double t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
    auto read = [&](const char *key, double *dst, size_t n) {
        if (const Json *v = node.get(key)) {
            if (v->size() != n) throw std::runtime_error(std::string("node.") + key + " has the wrong length");
            for (size_t k = 0; k < n; ++k) dst[k] = v->arr[k].num;
        }
    };
    read("translation", t, 3);
    read("rotation", q, 4);
    read("scale", s, 3);
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    Mat4 r = identity();
    r.m[0] = (1 - 2 * (y * y + z * z)) * s[0];
    r.m[1] = (2 * (x * y + z * w)) * s[0];
    r.m[2] = (2 * (x * z - y * w)) * s[0];
    r.m[4] = (2 * (x * y - z * w)) * s[1];
    r.m[5] = (1 - 2 * (x * x + z * z)) * s[1];
    r.m[6] = (2 * (y * z + x * w)) * s[1];
    r.m[8] = (2 * (x * z + y * w)) * s[2];
    r.m[9] = (2 * (y * z - x * w)) * s[2];
    r.m[10] = (1 - 2 * (x * x + y * y)) * s[2];
    r.m[12] = t[0];
    r.m[13] = t[1];
    r.m[14] = t[2];
    return r;
}

// ---- the document -------------------------------------------------------------------------------
struct Gltf {
    Json doc;
    std::vector<std::vector<unsigned char>> buffers;
    pt_mesh *out = nullptr;             // mesh that primitive() appends to
    std::vector<pt_material> *mats = nullptr;   // material list that material() appends to
    pt_model *model = nullptr;          // set: keep meshes in object space and record instances
    std::vector<int> mesh_map;          // glTF mesh -> index in model->meshes, -1 = not yet, -2 = empty
    std::vector<int> material_map;      // glTF material -> index in *mats, -1 = not yet
    int default_material = -1;
    bool first_only = false, done = false;

    const Json &array(const char *key) const {
        static const Json empty;
        const Json *v = doc.get(key);
        return v && v->type == Json::Arr ? *v : empty;
    }
    const Json &at(const char *key, long i) const {
        const Json &a = array(key);
        if (i < 0 || (size_t)i >= a.size())
            throw std::runtime_error(std::string(key) + "[" + std::to_string(i) + "] does not exist");
        return a.arr[(size_t)i];
    }

    // Accessor i as a strided view of its buffer.
    struct View { const unsigned char *data; size_t stride, count; long component; int comps; };
    View view(long accessor) const {
        const Json &a = at("accessors", accessor);
        if (a.get("sparse")) throw std::runtime_error("sparse accessors are not supported");
        const long ct = a.index("componentType");
        // Sizes and offsets are JSON numbers (doubles): only non-negative integers that a double
        // holds exactly are taken, so that no cast or sum below can wrap.
        auto size_field = [](const Json &j, const char *key) -> size_t {
            const double v = j.number(key, 0);
            if (!(v >= 0.0) || v > 9007199254740992.0 || v != std::floor(v))
                throw std::runtime_error(std::string(key) + " is not a non-negative integer");
            return (size_t)v;
        };
        const size_t count = size_field(a, "count");
        const Json *ty = a.get("type");
        const std::string type = ty ? ty->str : "";
        const int comps = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : 0;
        const size_t csize = (ct == 5120 || ct == 5121) ? 1 : (ct == 5122 || ct == 5123) ? 2
                             : (ct == 5125 || ct == 5126) ? 4 : 0;
        if (!comps || !csize) throw std::runtime_error("accessor with an unsupported type");
        const Json &bv = at("bufferViews", a.index("bufferView"));
        const long b = bv.index("buffer");
        if (b < 0 || (size_t)b >= buffers.size()) throw std::runtime_error("bufferView.buffer out of range");
        const size_t buf_size = buffers[(size_t)b].size();
        const size_t view_off = size_field(bv, "byteOffset"), view_len = size_field(bv, "byteLength");
        const size_t acc_off = size_field(a, "byteOffset");
        const size_t elem = csize * (size_t)comps;
        size_t stride = size_field(bv, "byteStride");
        if (!stride) stride = elem;
        // the view lies inside its buffer, the accessor's elements inside the view
        if (view_off > buf_size || view_len > buf_size - view_off)
            throw std::runtime_error("bufferView reaches beyond its buffer");
        if (count) {
            if (acc_off > view_len || elem > view_len - acc_off ||
                count - 1 > (view_len - acc_off - elem) / stride)
                throw std::runtime_error("accessor reaches beyond its bufferView");
        } else if (acc_off > view_len) {
            throw std::runtime_error("accessor reaches beyond its bufferView");
        }
        const size_t off = view_off + acc_off;
        return View{buffers[(size_t)b].data() + off, stride, count, ct, comps};
    }

    int material(long gm) {
        if (gm < 0) {
            if (default_material < 0) {
                pt_material m{};
                m.kd[0] = m.kd[1] = m.kd[2] = 0.5f;
                m.roughness = 1.f;
                default_material = (int)mats->size();
                mats->push_back(m);
            }
            return default_material;
        }
        if ((size_t)gm >= material_map.size()) throw std::runtime_error("primitive.material out of range");
        if (material_map[(size_t)gm] >= 0) return material_map[(size_t)gm];
        const Json &mj = at("materials", gm);
        double base[3] = {1, 1, 1}, metallic = 1.0, rough = 1.0, em[3] = {0, 0, 0}, strength = 1.0;
        if (const Json *pbr = mj.get("pbrMetallicRoughness")) {
            if (const Json *c = pbr->get("baseColorFactor"))
                for (size_t k = 0; k < 3 && k < c->size(); ++k) base[k] = c->arr[k].num;
            metallic = pbr->number("metallicFactor", 1.0);
            rough = pbr->number("roughnessFactor", 1.0);
        }
        if (const Json *e = mj.get("emissiveFactor"))
            for (size_t k = 0; k < 3 && k < e->size(); ++k) em[k] = e->arr[k].num;
        if (const Json *ext = mj.get("extensions"))
            if (const Json *es = ext->get("KHR_materials_emissive_strength"))
                strength = es->number("emissiveStrength", 1.0);
        pt_material m{};
        for (int k = 0; k < 3; ++k) {
            m.kd[k] = (float)(base[k] * (1.0 - metallic));
            m.ks[k] = (float)(0.04 * (1.0 - metallic) + base[k] * metallic);
            m.ke[k] = (float)(em[k] * strength);
        }
        m.roughness = (float)rough;
        material_map[(size_t)gm] = (int)mats->size();
        mats->push_back(m);
        return material_map[(size_t)gm];
    }

    void primitive(const Json &prim, const Mat4 *world) {
        if (prim.number("mode", 4) != 4) return;                // triangle lists only (:139)
        const Json *attrs = prim.get("attributes");
        const long pos_acc = attrs ? attrs->index("POSITION") : -1;
        if (pos_acc < 0) throw std::runtime_error("primitive without POSITION");
        const View pv = view(pos_acc);
        if (pv.component != 5126 || pv.comps != 3) throw std::runtime_error("POSITION must be float VEC3");
        const size_t base_vertex = out->vtx.size() / 3;
        if (base_vertex + pv.count > 0x7FFFFFFFull) throw std::runtime_error("too many vertices");
        for (size_t i = 0; i < pv.count; ++i) {
            float p[3];
            std::memcpy(p, pv.data + i * pv.stride, 12);
            if (world) {
                const double x = p[0], y = p[1], z = p[2];
                const double *m = world->m;
                p[0] = (float)(m[0] * x + m[4] * y + m[8] * z + m[12]);
                p[1] = (float)(m[1] * x + m[5] * y + m[9] * z + m[13]);
                p[2] = (float)(m[2] * x + m[6] * y + m[10] * z + m[14]);
            }
            out->vtx.insert(out->vtx.end(), p, p + 3);
        }
        const long idx_acc = prim.index("indices");
        const size_t first_index = out->idx.size();
        if (idx_acc >= 0) {
            const View iv = view(idx_acc);
            if (iv.comps != 1 || (iv.component != 5121 && iv.component != 5123 && iv.component != 5125))
                throw std::runtime_error("indices must be unsigned scalars");
            for (size_t i = 0; i + 2 < iv.count; i += 3) {
                for (size_t k = 0; k < 3; ++k) {
                    const unsigned char *q = iv.data + (i + k) * iv.stride;
                    uint32_t v;
                    if (iv.component == 5121) v = *q;
                    else if (iv.component == 5123) { uint16_t h; std::memcpy(&h, q, 2); v = h; }
                    else v = rd32(q);
                    if (v >= pv.count) throw std::runtime_error("index " + std::to_string(v) + " outside the primitive's vertices");
                    out->idx.push_back((int32_t)(base_vertex + v));
                }
            }
        } else {
            for (size_t i = 0; i + 2 < pv.count; i += 3)
                for (size_t k = 0; k < 3; ++k) out->idx.push_back((int32_t)(base_vertex + i + k));
        }
        const size_t n_tri = (out->idx.size() - first_index) / 3;
        const int m = material(prim.index("material"));
        out->mat_ids.insert(out->mat_ids.end(), n_tri, m);
        if (first_only && n_tri) done = true;
    }

    void node(long ni, const Mat4 &parent, int depth) {
        if (done) return;
        if (depth > 256) throw std::runtime_error("node hierarchy too deep (cycle?)");
        const Json &n = at("nodes", ni);
        const Mat4 world = mul(parent, node_matrix(n));
        const long mi = n.index("mesh");
        if (mi >= 0 && model) {
            // every mesh once, in its own space; the node becomes an instance of it
            if ((size_t)mi >= mesh_map.size()) throw std::runtime_error("node.mesh out of range");
            if (mesh_map[(size_t)mi] == -1) {
                auto fresh = std::make_unique<pt_mesh>();
                out = fresh.get();
                if (const Json *prims = at("meshes", mi).get("primitives"))
                    for (const Json &p : prims->arr) primitive(p, nullptr);
                out = nullptr;
                if (fresh->idx.empty()) {
                    mesh_map[(size_t)mi] = -2;
                } else {
                    mesh_map[(size_t)mi] = (int)model->meshes.size();
                    model->meshes.push_back(fresh.release());
                }
            }
            if (mesh_map[(size_t)mi] >= 0) {
                pt_instance in{};
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 4; ++c) in.object_to_world[4 * r + c] = (float)world.m[4 * c + r];
                in.mesh = (uint32_t)mesh_map[(size_t)mi];
                in.material_offset = 0;
                model->instances.push_back(in);
            }
        } else if (mi >= 0) {
            const Json &mesh = at("meshes", mi);
            if (const Json *prims = mesh.get("primitives"))
                for (const Json &p : prims->arr) {
                    if (done) break;
                    primitive(p, first_only ? nullptr : &world);
                }
        }
        if (const Json *ch = n.get("children"))
            for (const Json &c : ch->arr) node((long)c.num, world, depth + 1);
    }
};

} // namespace

// Reads the file at path into g (document and buffers); returns PT_OK or fails.
static int open_gltf(const char *path, const char *who, Gltf &g) {
    std::vector<unsigned char> file;
    if (!read_file(path, file))
        return mipt::fail(PT_ERR_IO, std::string(who) + ": cannot open '" + path + "'");
    const unsigned char *json = file.data();
    size_t json_len = file.size();
    const unsigned char *bin = nullptr;
    size_t bin_len = 0;
    if (file.size() >= 12 && rd32(file.data()) == 0x46546C67u) {           // "glTF": binary container
        if (rd32(file.data() + 4) != 2) return mipt::fail(PT_ERR_UNSUPPORTED, std::string(who) + ": only glTF version 2");
        size_t off = 12;
        json = nullptr;
        while (off + 8 <= file.size()) {
            const size_t len = rd32(file.data() + off);
            const uint32_t type = rd32(file.data() + off + 4);
            if (off + 8 + len > file.size()) return mipt::fail(PT_ERR_IO, std::string(who) + ": truncated chunk");
            if (type == 0x4E4F534Au && !json) { json = file.data() + off + 8; json_len = len; }
            else if (type == 0x004E4942u && !bin) { bin = file.data() + off + 8; bin_len = len; }
            off += 8 + ((len + 3) & ~(size_t)3);
        }
        if (!json) return mipt::fail(PT_ERR_IO, std::string(who) + ": no JSON chunk");
    }
    JsonParser jp{reinterpret_cast<const char *>(json), reinterpret_cast<const char *>(json) + json_len};
    g.doc = jp.value();
    if (g.doc.type != Json::Obj) return mipt::fail(PT_ERR_INVALID, std::string(who) + ": the document is not an object");

    std::string dir(path);
    const size_t slash = dir.find_last_of("/\\");
    dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
    for (const Json &b : g.array("buffers").arr) {
        const Json *uri = b.get("uri");
        std::vector<unsigned char> data;
        if (!uri) {
            if (!bin) return mipt::fail(PT_ERR_INVALID, std::string(who) + ": buffer without uri outside a .glb");
            data.assign(bin, bin + bin_len);
        } else if (uri->str.compare(0, 5, "data:") == 0) {
            const size_t comma = uri->str.find(',');
            if (comma == std::string::npos) return mipt::fail(PT_ERR_INVALID, std::string(who) + ": malformed data URI");
            data = base64(uri->str, comma + 1);
        } else if (!read_file(dir + uri->str, data)) {
            return mipt::fail(PT_ERR_IO, std::string(who) + ": cannot open buffer '" + dir + uri->str + "'");
        }
        if (data.size() < (size_t)b.number("byteLength", 0))
            return mipt::fail(PT_ERR_IO, std::string(who) + ": buffer shorter than its byteLength");
        g.buffers.push_back(std::move(data));
    }
    g.material_map.assign(g.array("materials").size(), -1);
    g.mesh_map.assign(g.array("meshes").size(), -1);
    return PT_OK;
}

// Walks the default scene's roots, else every node that is nobody's child.
static void walk_scene(Gltf &g) {
    std::vector<long> roots;
    const Json &scenes = g.array("scenes");
    if (scenes.size()) {
        long si = g.doc.index("scene");
        if (si < 0) si = 0;
        if (const Json *rn = g.at("scenes", si).get("nodes"))
            for (const Json &r : rn->arr) roots.push_back((long)r.num);
    } else {
        const Json &nodes = g.array("nodes");
        std::vector<char> is_child(nodes.size(), 0);
        for (const Json &n : nodes.arr)
            if (const Json *ch = n.get("children"))
                for (const Json &c : ch->arr)
                    if (c.num >= 0 && (size_t)c.num < is_child.size()) is_child[(size_t)c.num] = 1;
        for (size_t i = 0; i < nodes.size(); ++i)
            if (!is_child[i]) roots.push_back((long)i);
    }
    for (long r : roots) g.node(r, identity(), 0);
}

extern "C" int pt_gltf_load(const char *path, uint32_t flags, pt_mesh **out) {
    if (!path || !out) return mipt::fail(PT_ERR_INVALID, "pt_gltf_load: null argument");
    *out = nullptr;
    try {
        Gltf g;
        const int rc = open_gltf(path, "pt_gltf_load", g);
        if (rc != PT_OK) return rc;
        auto mesh = std::make_unique<pt_mesh>();
        g.out = mesh.get();
        g.mats = &mesh->materials;
        g.first_only = (flags & PT_GLTF_FIRST_MESH_ONLY) != 0;
        walk_scene(g);
        if (mesh->idx.empty()) return mipt::fail(PT_ERR_INVALID, "pt_gltf_load: no triangles in the scene");
        *out = mesh.release();
        return PT_OK;
    } catch (const std::bad_alloc &) {
        return mipt::fail(PT_ERR_NOMEM, "out of host memory");
    } catch (const std::exception &e) {
        return mipt::fail(PT_ERR_INVALID, std::string("pt_gltf_load: ") + e.what());
    }
}

extern "C" {

int pt_gltf_load_model(const char *path, pt_model **out) {
    if (!path || !out) return mipt::fail(PT_ERR_INVALID, "pt_gltf_load_model: null argument");
    *out = nullptr;
    try {
        Gltf g;
        const int rc = open_gltf(path, "pt_gltf_load_model", g);
        if (rc != PT_OK) return rc;
        auto model = std::make_unique<pt_model>();
        g.model = model.get();
        g.mats = &model->materials;
        walk_scene(g);
        if (model->instances.empty())
            return mipt::fail(PT_ERR_INVALID, "pt_gltf_load_model: no triangles in the scene");
        *out = model.release();
        return PT_OK;
    } catch (const std::bad_alloc &) {
        return mipt::fail(PT_ERR_NOMEM, "out of host memory");
    } catch (const std::exception &e) {
        return mipt::fail(PT_ERR_INVALID, std::string("pt_gltf_load_model: ") + e.what());
    }
}

uint32_t pt_model_mesh_count(const pt_model *m) { return m ? (uint32_t)m->meshes.size() : 0; }
const pt_mesh *pt_model_mesh(const pt_model *m, uint32_t i) {
    return m && i < m->meshes.size() ? m->meshes[i] : nullptr;
}
uint32_t pt_model_instance_count(const pt_model *m) { return m ? (uint32_t)m->instances.size() : 0; }
const pt_instance *pt_model_instances(const pt_model *m) { return m ? m->instances.data() : nullptr; }
uint32_t pt_model_material_count(const pt_model *m) { return m ? (uint32_t)m->materials.size() : 0; }
const pt_material *pt_model_materials(const pt_model *m) { return m ? m->materials.data() : nullptr; }
int pt_model_destroy(pt_model *m) {
    delete m;
    return PT_OK;
}

} // extern "C"
