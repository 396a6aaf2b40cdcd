// C ABI: files -- BVH cache, film checkpoints, image output.
//
// None of this exists in the reference: it rebuilds its BVH on every start
// (src/main_granite.cpp:289-293), has no accumulation buffer to keep, and presents to a
// swapchain.  The closest relative is the Vulkan pipeline cache it loads and saves around a run
// (src/main.cpp:290-315,1187-1196), whose pattern (read if present and valid, else rebuild and
// write) is what the BVH cache offers the caller.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mipt.h"
#include "internal.h"

using namespace mipt;

namespace {

// FNV-1a over 64-bit words (the tail bytes one at a time): cheap, order sensitive.
uint64_t fnv1a(const void *data, size_t bytes, uint64_t h = 0xcbf29ce484222325ull) {
    const unsigned char *p = static_cast<const unsigned char *>(data);
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        uint64_t w;
        std::memcpy(&w, p + i, 8);
        h = (h ^ w) * 0x100000001b3ull;
    }
    for (; i < bytes; ++i) h = (h ^ p[i]) * 0x100000001b3ull;
    return h;
}

struct FileCloser {
    void operator()(FILE *f) const { if (f) std::fclose(f); }
};
using File = std::unique_ptr<FILE, FileCloser>;

struct BvhFileHeader {
    char magic[8];            // "MIPTBVH1"
    uint32_t link_encoding;
    uint32_t n_nodes;
    uint64_t key;             // caller's key (pt_bvh_cache_key of the inputs it was built from)
    uint64_t payload_hash;    // fnv1a of the node array
};

struct FilmFileHeader {
    char magic[8];            // "MIPTFLM1"
    uint32_t width, height;
    uint64_t payload_hash;
};

} // namespace

extern "C" {

uint64_t pt_bvh_cache_key(const pt_bbox *leaf_bounds, uint32_t n_leaves, const pt_bvh_options *opt) {
    uint64_t h = fnv1a(leaf_bounds, leaf_bounds ? sizeof(pt_bbox) * (size_t)n_leaves : 0);
    if (opt) h = fnv1a(opt, sizeof(*opt), h);
    return fnv1a(&n_leaves, sizeof(n_leaves), h);
}

int pt_bvh_save(const pt_bvh *bvh, const char *path, uint64_t key) {
    if (!bvh || !path) return fail(PT_ERR_INVALID, "pt_bvh_save: null argument");
    File f(std::fopen(path, "wb"));
    if (!f) return fail(PT_ERR_IO, std::string("pt_bvh_save: cannot open ") + path);
    BvhFileHeader h{};
    std::memcpy(h.magic, "MIPTBVH1", 8);
    h.link_encoding = bvh->link_encoding;
    h.n_nodes = (uint32_t)bvh->nodes.size();
    h.key = key;
    h.payload_hash = fnv1a(bvh->nodes.data(), sizeof(pt_bbox) * bvh->nodes.size());
    if (std::fwrite(&h, sizeof(h), 1, f.get()) != 1 ||
        std::fwrite(bvh->nodes.data(), sizeof(pt_bbox), bvh->nodes.size(), f.get()) != bvh->nodes.size())
        return fail(PT_ERR_IO, std::string("pt_bvh_save: short write to ") + path);
    return PT_OK;
}

int pt_bvh_load(const char *path, uint64_t key, pt_bvh **out) {
    if (!path || !out) return fail(PT_ERR_INVALID, "pt_bvh_load: null argument");
    *out = nullptr;
    File f(std::fopen(path, "rb"));
    if (!f) return fail(PT_ERR_IO, std::string("pt_bvh_load: cannot open ") + path);
    BvhFileHeader h{};
    if (std::fread(&h, sizeof(h), 1, f.get()) != 1 || std::memcmp(h.magic, "MIPTBVH1", 8) != 0)
        return fail(PT_ERR_IO, std::string("pt_bvh_load: not a BVH cache file: ") + path);
    if (h.key != key) return fail(PT_ERR_STALE, "pt_bvh_load: cache was built from other inputs");
    if (h.n_nodes == 0 || (h.link_encoding != PT_LINKS_FLOAT_VALUE && h.link_encoding != PT_LINKS_INT_BITS))
        return fail(PT_ERR_IO, "pt_bvh_load: corrupt header");
    try {
        auto b = std::make_unique<pt_bvh>();
        b->link_encoding = h.link_encoding;
        b->nodes.resize(h.n_nodes);
        if (std::fread(b->nodes.data(), sizeof(pt_bbox), h.n_nodes, f.get()) != h.n_nodes)
            return fail(PT_ERR_IO, "pt_bvh_load: truncated file");
        if (fnv1a(b->nodes.data(), sizeof(pt_bbox) * (size_t)h.n_nodes) != h.payload_hash)
            return fail(PT_ERR_IO, "pt_bvh_load: checksum mismatch");
        *out = b.release();
    } catch (const std::bad_alloc &) {
        return fail(PT_ERR_NOMEM, "out of host memory");
    }
    return PT_OK;
}

int pt_film_save(pt_film *film, const char *path) {
    if (!film || !path) return fail(PT_ERR_INVALID, "pt_film_save: null argument");
    std::vector<float> px;
    try {
        px.resize(4ull * film->width * film->height);
    } catch (const std::bad_alloc &) {
        return fail(PT_ERR_NOMEM, "out of host memory");
    }
    int rc = pt_film_read(film, px.data());
    if (rc != PT_OK) return rc;
    File f(std::fopen(path, "wb"));
    if (!f) return fail(PT_ERR_IO, std::string("pt_film_save: cannot open ") + path);
    FilmFileHeader h{};
    std::memcpy(h.magic, "MIPTFLM1", 8);
    h.width = film->width;
    h.height = film->height;
    h.payload_hash = fnv1a(px.data(), px.size() * sizeof(float));
    if (std::fwrite(&h, sizeof(h), 1, f.get()) != 1 ||
        std::fwrite(px.data(), sizeof(float), px.size(), f.get()) != px.size())
        return fail(PT_ERR_IO, std::string("pt_film_save: short write to ") + path);
    return PT_OK;
}

int pt_film_load(pt_film *film, const char *path) {
    if (!film || !path) return fail(PT_ERR_INVALID, "pt_film_load: null argument");
    File f(std::fopen(path, "rb"));
    if (!f) return fail(PT_ERR_IO, std::string("pt_film_load: cannot open ") + path);
    FilmFileHeader h{};
    if (std::fread(&h, sizeof(h), 1, f.get()) != 1 || std::memcmp(h.magic, "MIPTFLM1", 8) != 0)
        return fail(PT_ERR_IO, std::string("pt_film_load: not a film checkpoint: ") + path);
    if (h.width != film->width || h.height != film->height)
        return fail(PT_ERR_INVALID, "pt_film_load: checkpoint has another image size");
    std::vector<float> px;
    try {
        px.resize(4ull * h.width * h.height);
    } catch (const std::bad_alloc &) {
        return fail(PT_ERR_NOMEM, "out of host memory");
    }
    if (std::fread(px.data(), sizeof(float), px.size(), f.get()) != px.size())
        return fail(PT_ERR_IO, "pt_film_load: truncated file");
    if (fnv1a(px.data(), px.size() * sizeof(float)) != h.payload_hash)
        return fail(PT_ERR_IO, "pt_film_load: checksum mismatch");
    hipError_t e = hipSetDevice(film->ctx->device);
    if (e == hipSuccess)
        e = hipMemcpyAsync(film->data, px.data(), px.size() * sizeof(float), hipMemcpyHostToDevice,
                           film->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(film->ctx->stream);
    if (e != hipSuccess) return hip_fail(e, "pt_film_load upload");
    return PT_OK;
}

int pt_film_write_image(pt_film *film, const char *path, uint32_t format, float exposure) {
    if (!film || !path) return fail(PT_ERR_INVALID, "pt_film_write_image: null argument");
    if (format != PT_IMAGE_PPM && format != PT_IMAGE_PFM)
        return fail(PT_ERR_INVALID, "pt_film_write_image: unknown format");
    if (!(exposure > 0.f)) return fail(PT_ERR_INVALID, "pt_film_write_image: exposure must be > 0");
    const uint32_t W = film->width, H = film->height;
    std::vector<float> px;
    try {
        px.resize(4ull * W * H);
    } catch (const std::bad_alloc &) {
        return fail(PT_ERR_NOMEM, "out of host memory");
    }
    int rc = pt_film_read(film, px.data());
    if (rc != PT_OK) return rc;
    File f(std::fopen(path, "wb"));
    if (!f) return fail(PT_ERR_IO, std::string("pt_film_write_image: cannot open ") + path);
    auto mean = [&](uint32_t x, uint32_t y, int c) {
        const float *p = px.data() + 4 * ((size_t)y * W + x);
        return p[3] > 0.f ? p[c] / p[3] * exposure : 0.f;
    };
    bool ok = true;
    if (format == PT_IMAGE_PFM) {                 // linear floats, bottom row first
        ok = std::fprintf(f.get(), "PF\n%u %u\n-1.0\n", W, H) > 0;
        std::vector<float> row(3 * (size_t)W);
        for (uint32_t y = H; ok && y-- > 0;) {
            for (uint32_t x = 0; x < W; ++x)
                for (int c = 0; c < 3; ++c) row[3 * (size_t)x + c] = mean(x, y, c);
            ok = std::fwrite(row.data(), sizeof(float), row.size(), f.get()) == row.size();
        }
    } else {                                      // 8-bit, sRGB transfer curve
        ok = std::fprintf(f.get(), "P6\n%u %u\n255\n", W, H) > 0;
        std::vector<unsigned char> row(3 * (size_t)W);
        for (uint32_t y = 0; ok && y < H; ++y) {
            for (uint32_t x = 0; x < W; ++x)
                for (int c = 0; c < 3; ++c) {
                    float v = mean(x, y, c);
                    v = v != v ? 0.f : std::fmin(std::fmax(v, 0.f), 1.f);
                    v = v <= 0.0031308f ? 12.92f * v : 1.055f * std::pow(v, 1.f / 2.4f) - 0.055f;
                    row[3 * (size_t)x + c] = (unsigned char)std::lround(v * 255.f);
                }
            ok = std::fwrite(row.data(), 1, row.size(), f.get()) == row.size();
        }
    }
    if (!ok) return fail(PT_ERR_IO, std::string("pt_film_write_image: short write to ") + path);
    return PT_OK;
}

} // extern "C"
