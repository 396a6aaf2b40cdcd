// Minimal Wavefront OBJ (+MTL) reader.
//
// Supplies what the reference takes from tinyobj::LoadObj (src/main.cpp:852-899):
// the packed vertex positions (attrib.vertices) and, per face corner, the
// position index (mesh.indices[].vertex_index), with polygons triangulated as
// fans.  All shapes of the file are merged into one mesh (the reference uses
// shapes[0] only, src/main.cpp:862), with each triangle's shape and its index within the
// shape kept beside it.  Additionally keeps per-triangle material
// ids and the Kd/Ks/Ke/Ns/Pr entries of the MTL file, which the shading
// kernels need and the reference never reads.
#include <cctype>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mipt.h"
#include "../csrc/internal.h"
#include "mesh.h"


namespace {

pt_material default_material() {
    pt_material m{};
    m.kd[0] = m.kd[1] = m.kd[2] = 0.5f;
    m.roughness = 1.f;
    return m;
}

std::string dirname_of(const std::string &path) {
    const size_t p = path.find_last_of("/\\");
    return p == std::string::npos ? std::string() : path.substr(0, p + 1);
}

bool read3(std::istringstream &ss, float *out) {
    return (bool)(ss >> out[0] >> out[1] >> out[2]);
}

void load_mtl(const std::string &path, std::map<std::string, int> &by_name,
              std::vector<pt_material> &mats) {
    std::ifstream f(path);
    if (!f) return;  // tinyobj also only warns about a missing material file
    std::string line;
    int cur = -1;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string tag;
        if (!(ss >> tag) || tag[0] == '#') continue;
        if (tag == "newmtl") {
            std::string name;
            ss >> name;
            mats.push_back(default_material());
            cur = (int)mats.size() - 1;
            by_name[name] = cur;
        } else if (cur >= 0) {
            pt_material &m = mats[(size_t)cur];
            float v[3];
            if (tag == "Kd" && read3(ss, v)) { m.kd[0] = v[0]; m.kd[1] = v[1]; m.kd[2] = v[2]; }
            else if (tag == "Ks" && read3(ss, v)) { m.ks[0] = v[0]; m.ks[1] = v[1]; m.ks[2] = v[2]; }
            else if (tag == "Ke" && read3(ss, v)) { m.ke[0] = v[0]; m.ke[1] = v[1]; m.ke[2] = v[2]; }
            else if (tag == "Ns") {
                float ns;
                if (ss >> ns) m.roughness = std::sqrt(std::sqrt(2.f / (std::fmax(ns, 0.f) + 2.f)));
            } else if (tag == "Pr") {
                float pr;
                if (ss >> pr) m.roughness = std::fmin(std::fmax(pr, 0.f), 1.f);
            }
        }
    }
}

} // namespace

extern "C" {

int pt_obj_load(const char *path, pt_mesh **out) {
    if (!path || !out) return mipt::fail(PT_ERR_INVALID, "pt_obj_load: null argument");
    *out = nullptr;
    try {
        std::ifstream f(path);
        if (!f) return mipt::fail(PT_ERR_IO, std::string("pt_obj_load: cannot open '") + path + "'");
        auto mesh = std::make_unique<pt_mesh>();
        std::map<std::string, int> mat_by_name;
        int cur_mat = -1;
        // shapes as tinyobj::LoadObj cuts them (src/main.cpp:860): an `o` or `g` statement ends
        // the running shape if it has faces
        int32_t cur_shape = 0, faces_in_shape = 0;
        std::string line;
        size_t lineno = 0;
        std::vector<int32_t> corners;
        while (std::getline(f, line)) {
            ++lineno;
            std::istringstream ss(line);
            std::string tag;
            if (!(ss >> tag) || tag[0] == '#') continue;
            if (tag == "v") {
                float v[3];
                if (!read3(ss, v))
                    return mipt::fail(PT_ERR_INVALID, std::string("pt_obj_load: ") + path + ":" +
                                                          std::to_string(lineno) + ": bad vertex");
                mesh->vtx.insert(mesh->vtx.end(), v, v + 3);
            } else if (tag == "f") {
                corners.clear();
                std::string tok;
                const int32_t nv = (int32_t)(mesh->vtx.size() / 3);
                while (ss >> tok) {
                    char *end = nullptr;
                    long vi = std::strtol(tok.c_str(), &end, 10);   // "v", "v/vt", "v//vn", "v/vt/vn"
                    if (end == tok.c_str() || vi == 0)
                        return mipt::fail(PT_ERR_INVALID, std::string("pt_obj_load: ") + path + ":" +
                                                              std::to_string(lineno) + ": bad face index");
                    const long z = vi > 0 ? vi - 1 : nv + vi;       // negative = relative to the end
                    if (z < 0 || z >= nv)
                        return mipt::fail(PT_ERR_INVALID, std::string("pt_obj_load: ") + path + ":" +
                                                              std::to_string(lineno) +
                                                              ": face index out of range");
                    corners.push_back((int32_t)z);
                }
                if (corners.size() < 3)
                    return mipt::fail(PT_ERR_INVALID, std::string("pt_obj_load: ") + path + ":" +
                                                          std::to_string(lineno) +
                                                          ": face with fewer than 3 vertices");
                for (size_t k = 1; k + 1 < corners.size(); ++k) {
                    mesh->idx.push_back(corners[0]);
                    mesh->idx.push_back(corners[k]);
                    mesh->idx.push_back(corners[k + 1]);
                    mesh->mat_ids.push_back(cur_mat);
                    mesh->shape_ids.push_back(cur_shape);
                    mesh->shape_prim_ids.push_back(faces_in_shape++);
                }
            } else if (tag == "o" || tag == "g") {
                if (faces_in_shape > 0) {
                    ++cur_shape;
                    faces_in_shape = 0;
                }
            } else if (tag == "mtllib") {
                std::string name;
                ss >> name;
                load_mtl(dirname_of(path) + name, mat_by_name, mesh->materials);
            } else if (tag == "usemtl") {
                std::string name;
                ss >> name;
                auto it = mat_by_name.find(name);
                cur_mat = it == mat_by_name.end() ? -1 : it->second;
            }
        }
        if (mesh->idx.empty())
            return mipt::fail(PT_ERR_INVALID, std::string("pt_obj_load: '") + path + "' has no faces");
        // Faces without a (known) material share one default entry appended at the end.
        int def = -1;
        for (auto &m : mesh->mat_ids)
            if (m < 0) {
                if (def < 0) {
                    mesh->materials.push_back(default_material());
                    def = (int)mesh->materials.size() - 1;
                }
                m = def;
            }
        mesh->n_shapes = (uint32_t)(faces_in_shape > 0 ? cur_shape + 1 : cur_shape);
        *out = mesh.release();
        return PT_OK;
    } catch (const std::bad_alloc &) {
        return mipt::fail(PT_ERR_NOMEM, "pt_obj_load: out of memory");
    } catch (const std::exception &e) {
        return mipt::fail(PT_ERR_INVALID, std::string("pt_obj_load: ") + e.what());
    }
}

uint32_t pt_mesh_vertex_count(const pt_mesh *m) { return m ? (uint32_t)(m->vtx.size() / 3) : 0; }
uint32_t pt_mesh_triangle_count(const pt_mesh *m) { return m ? (uint32_t)(m->idx.size() / 3) : 0; }
uint32_t pt_mesh_material_count(const pt_mesh *m) { return m ? (uint32_t)m->materials.size() : 0; }
const float *pt_mesh_vertices(const pt_mesh *m) { return m ? m->vtx.data() : nullptr; }
const int32_t *pt_mesh_indices(const pt_mesh *m) { return m ? m->idx.data() : nullptr; }
const int32_t *pt_mesh_material_ids(const pt_mesh *m) { return m ? m->mat_ids.data() : nullptr; }
const pt_material *pt_mesh_materials(const pt_mesh *m) { return m ? m->materials.data() : nullptr; }
uint32_t pt_mesh_shape_count(const pt_mesh *m) { return m ? m->n_shapes : 0; }
const int32_t *pt_mesh_shape_ids(const pt_mesh *m) {
    return m && !m->shape_ids.empty() ? m->shape_ids.data() : nullptr;
}
const int32_t *pt_mesh_shape_prim_ids(const pt_mesh *m) {
    return m && !m->shape_prim_ids.empty() ? m->shape_prim_ids.data() : nullptr;
}

// src/main.cpp:865-880: centre on the bounding box and divide by its largest extent.
int pt_mesh_normalize_unit_cube(pt_mesh *m) {
    if (!m) return mipt::fail(PT_ERR_INVALID, "pt_mesh_normalize_unit_cube: null mesh");
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = 0; i < m->vtx.size(); i += 3)
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::fmin(lo[k], m->vtx[i + k]);
            hi[k] = std::fmax(hi[k], m->vtx[i + k]);
        }
    float c[3], scale = 0.f;
    for (int k = 0; k < 3; ++k) {
        c[k] = 0.5f * (hi[k] + lo[k]);
        scale = std::fmax(scale, hi[k] - lo[k]);
    }
    if (!(scale > 0.f)) return mipt::fail(PT_ERR_INVALID, "pt_mesh_normalize_unit_cube: empty extent");
    for (size_t i = 0; i < m->vtx.size(); i += 3)
        for (int k = 0; k < 3; ++k) m->vtx[i + k] = (m->vtx[i + k] - c[k]) / scale;
    return PT_OK;
}

int pt_mesh_destroy(pt_mesh *m) {
    delete m;
    return PT_OK;
}

} // extern "C"
