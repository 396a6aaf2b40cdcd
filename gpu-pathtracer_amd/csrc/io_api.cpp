// C ABI: files -- BVH cache, film checkpoints, image output.
//
// None of this exists in the reference: it rebuilds its BVH on every start
// (src/main_granite.cpp:289-293), has no accumulation buffer to keep, and presents to a
// swapchain.  The closest relative is the Vulkan pipeline cache it loads and saves around a run
// (src/main.cpp:290-315,1187-1196), whose pattern (read if present and valid, else rebuild and
// write) is what the BVH cache offers the caller.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
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

// Writes header + payload to a staging file of its own next to path (path + ".tmp.<pid>.<n>",
// created exclusively, so two savers of one path -- threads or processes -- never share it) and
// renames it over path once everything, the close included, has succeeded: an interrupted or
// failed save leaves the previous file as it was, and of several concurrent saves the last rename
// wins with a whole file.
int write_file_atomically(const char *who, const char *path, const void *header, size_t header_bytes,
                          const void *payload, size_t payload_bytes) {
    static std::atomic<unsigned> counter{0};
    std::string tmp;
    int fd = -1;
    for (int attempt = 0; attempt < 64 && fd < 0; ++attempt) {
        tmp = std::string(path) + ".tmp." + std::to_string((long long)getpid()) + "." +
              std::to_string(counter.fetch_add(1));
        fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_CLOEXEC, 0644);
        if (fd < 0 && errno != EEXIST) break;
    }
    FILE *f = fd >= 0 ? ::fdopen(fd, "wb") : nullptr;
    if (!f) {
        if (fd >= 0) {
            ::close(fd);
            std::remove(tmp.c_str());
        }
        return fail(PT_ERR_IO, std::string(who) + ": cannot create " + tmp);
    }
    bool ok = std::fwrite(header, header_bytes, 1, f) == 1 &&
              (payload_bytes == 0 || std::fwrite(payload, payload_bytes, 1, f) == 1) &&
              std::fflush(f) == 0;
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) {
        std::remove(tmp.c_str());
        return fail(PT_ERR_IO, std::string(who) + ": short write to " + tmp);
    }
    if (std::rename(tmp.c_str(), path) != 0) {
        std::remove(tmp.c_str());
        return fail(PT_ERR_IO, std::string(who) + ": cannot rename " + tmp + " to " + path);
    }
    return PT_OK;
}

// Bytes left in the file from the current position, or -1.
long long bytes_left(FILE *f) {
    const long here = std::ftell(f);
    if (here < 0 || std::fseek(f, 0, SEEK_END) != 0) return -1;
    const long end = std::ftell(f);
    if (end < 0 || std::fseek(f, here, SEEK_SET) != 0) return -1;
    return (long long)end - here;
}

// Bumped whenever the host builder, its optimiser or the leaf-box inflation change what they
// produce from the same inputs, so that trees cached by an older library are rebuilt.
constexpr uint64_t kBuilderRevision = 2;

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
    if (opt) {
        // only what determines the tree: num_threads does not
        pt_bvh_options o;
        std::memset(&o, 0, sizeof(o));
        o.use_sah = opt->use_sah;
        o.use_splits = opt->use_splits;
        o.max_split_depth = opt->max_split_depth;
        o.num_bins = opt->num_bins;
        o.min_overlap = opt->min_overlap;
        o.traversal_cost = opt->traversal_cost;
        o.extra_node_budget = opt->extra_node_budget;
        o.link_encoding = opt->link_encoding;
        o.optimize_passes = opt->optimize_passes;
        h = fnv1a(&o, sizeof(o), h);
    }
    h = fnv1a(&kBuilderRevision, sizeof(kBuilderRevision), h);
    return fnv1a(&n_leaves, sizeof(n_leaves), h);
}

int pt_bvh_save(const pt_bvh *bvh, const char *path, uint64_t key) {
    if (!bvh || !path) return fail(PT_ERR_INVALID, "pt_bvh_save: null argument");
    BvhFileHeader h{};
    std::memcpy(h.magic, "MIPTBVH1", 8);
    h.link_encoding = bvh->link_encoding;
    h.n_nodes = (uint32_t)bvh->nodes.size();
    h.key = key;
    h.payload_hash = fnv1a(bvh->nodes.data(), sizeof(pt_bbox) * bvh->nodes.size());
    return write_file_atomically("pt_bvh_save", path, &h, sizeof(h), bvh->nodes.data(),
                                 sizeof(pt_bbox) * bvh->nodes.size());
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
    // a skip-link array has 2n-1 nodes, and the file must hold exactly what the header announces
    // (checked before anything of that size is allocated)
    if (h.n_nodes % 2 == 0 || bytes_left(f.get()) != (long long)sizeof(pt_bbox) * h.n_nodes)
        return fail(PT_ERR_IO, "pt_bvh_load: header and file size disagree");
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
    FilmFileHeader h{};
    std::memcpy(h.magic, "MIPTFLM1", 8);
    h.width = film->width;
    h.height = film->height;
    h.payload_hash = fnv1a(px.data(), px.size() * sizeof(float));
    return write_file_atomically("pt_film_save", path, &h, sizeof(h), px.data(),
                                 px.size() * sizeof(float));
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
    const uint32_t container = format & 0xffu, curve = format & ~0xffu;
    if (container != PT_IMAGE_PPM && container != PT_IMAGE_PFM && container != PT_IMAGE_PNG)
        return fail(PT_ERR_INVALID, "pt_film_write_image: unknown format");
    if (curve != 0 && curve != PT_IMAGE_TONEMAP_REINHARD && curve != PT_IMAGE_TONEMAP_FILMIC)
        return fail(PT_ERR_INVALID, "pt_film_write_image: unknown tone curve");
    if (curve != 0 && container == PT_IMAGE_PFM)
        return fail(PT_ERR_INVALID, "pt_film_write_image: PFM holds linear radiance; a tone curve "
                                    "applies to the 8-bit formats only");
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
    // Display value of one pixel: tone curve on the luminance-preserving RGB triple, clamp,
    // sRGB transfer function, 8 bits.
    auto encode_pixel = [&](uint32_t x, uint32_t y, unsigned char *out) {
        float v[3];
        for (int c = 0; c < 3; ++c) {
            v[c] = mean(x, y, c);
            if (!(v[c] >= 0.f)) v[c] = 0.f;       // NaN and negative radiance show as black
        }
        if (curve == PT_IMAGE_TONEMAP_REINHARD) {
            // Reinhard on luminance: L -> L / (1 + L), colour ratios kept.
            const float L = 0.2126f * v[0] + 0.7152f * v[1] + 0.0722f * v[2];
            const float s = 1.f / (1.f + L);
            for (int c = 0; c < 3; ++c) v[c] *= s;
        } else if (curve == PT_IMAGE_TONEMAP_FILMIC) {
            // Per-channel rational fit of the ACES reference rendering curve (Narkowicz 2015).
            for (int c = 0; c < 3; ++c) {
                const float t = v[c];
                v[c] = (t * (2.51f * t + 0.03f)) / (t * (2.43f * t + 0.59f) + 0.14f);
            }
        }
        for (int c = 0; c < 3; ++c) {
            float t = std::fmin(v[c], 1.f);
            t = t <= 0.0031308f ? 12.92f * t : 1.055f * std::pow(t, 1.f / 2.4f) - 0.055f;
            out[c] = (unsigned char)std::lround(t * 255.f);
        }
    };
    bool ok = true;
    if (container == PT_IMAGE_PFM) {              // linear floats, bottom row first
        ok = std::fprintf(f.get(), "PF\n%u %u\n-1.0\n", W, H) > 0;
        std::vector<float> row(3 * (size_t)W);
        for (uint32_t y = H; ok && y-- > 0;) {
            for (uint32_t x = 0; x < W; ++x)
                for (int c = 0; c < 3; ++c) row[3 * (size_t)x + c] = mean(x, y, c);
            ok = std::fwrite(row.data(), sizeof(float), row.size(), f.get()) == row.size();
        }
    } else if (container == PT_IMAGE_PPM) {
        ok = std::fprintf(f.get(), "P6\n%u %u\n255\n", W, H) > 0;
        std::vector<unsigned char> row(3 * (size_t)W);
        for (uint32_t y = 0; ok && y < H; ++y) {
            for (uint32_t x = 0; x < W; ++x) encode_pixel(x, y, &row[3 * (size_t)x]);
            ok = std::fwrite(row.data(), 1, row.size(), f.get()) == row.size();
        }
    } else {
        // PNG, 8-bit RGB, no interlace.  The image data is a zlib stream of "stored" deflate
        // blocks (no compression: nothing to link against, and a decoder reads it like any
        // other PNG); every scanline starts with filter byte 0.
        std::vector<unsigned char> raw;
        try {
            raw.resize((size_t)H * (1 + 3 * (size_t)W));
        } catch (const std::bad_alloc &) {
            return fail(PT_ERR_NOMEM, "out of host memory");
        }
        for (uint32_t y = 0; y < H; ++y) {
            unsigned char *row = &raw[(size_t)y * (1 + 3 * (size_t)W)];
            row[0] = 0;
            for (uint32_t x = 0; x < W; ++x) encode_pixel(x, y, row + 1 + 3 * (size_t)x);
        }
        uint32_t crc_table[256];
        for (uint32_t n = 0; n < 256; ++n) {
            uint32_t c = n;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            crc_table[n] = c;
        }
        auto crc_update = [&](uint32_t c, const unsigned char *d, size_t n) {
            for (size_t i = 0; i < n; ++i) c = crc_table[(c ^ d[i]) & 0xffu] ^ (c >> 8);
            return c;
        };
        auto be32 = [](unsigned char *o, uint32_t v) {
            o[0] = (unsigned char)(v >> 24); o[1] = (unsigned char)(v >> 16);
            o[2] = (unsigned char)(v >> 8);  o[3] = (unsigned char)v;
        };
        auto chunk = [&](const char type[4], const unsigned char *d, size_t n) {
            unsigned char hdr[8], tail[4];
            be32(hdr, (uint32_t)n);
            std::memcpy(hdr + 4, type, 4);
            uint32_t c = crc_update(0xffffffffu, hdr + 4, 4);
            c = crc_update(c, d, n) ^ 0xffffffffu;
            be32(tail, c);
            return std::fwrite(hdr, 1, 8, f.get()) == 8 &&
                   (n == 0 || std::fwrite(d, 1, n, f.get()) == n) &&
                   std::fwrite(tail, 1, 4, f.get()) == 4;
        };
        static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
        unsigned char ihdr[13];
        be32(ihdr, W);
        be32(ihdr + 4, H);
        ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
        ok = std::fwrite(sig, 1, 8, f.get()) == 8 && chunk("IHDR", ihdr, 13);

        const size_t kBlock = 65535, n_blocks = (raw.size() + kBlock - 1) / kBlock;
        std::vector<unsigned char> z;
        if (2 + raw.size() + 5 * n_blocks + 4 > 0x7fffffffull)
            return fail(PT_ERR_INVALID, "pt_film_write_image: image too large for one PNG data chunk");
        try {
            z.reserve(2 + raw.size() + 5 * n_blocks + 4);
        } catch (const std::bad_alloc &) {
            return fail(PT_ERR_NOMEM, "out of host memory");
        }
        z.push_back(0x78); z.push_back(0x01);
        uint32_t s1 = 1, s2 = 0;                  // Adler-32 of the uncompressed bytes
        for (size_t off = 0, blk = 0; blk < n_blocks; ++blk, off += kBlock) {
            const size_t len = std::min(kBlock, raw.size() - off);
            z.push_back(blk + 1 == n_blocks ? 1 : 0);
            z.push_back((unsigned char)(len & 0xff)); z.push_back((unsigned char)(len >> 8));
            z.push_back((unsigned char)(~len & 0xff)); z.push_back((unsigned char)((~len >> 8) & 0xff));
            z.insert(z.end(), raw.begin() + (ptrdiff_t)off, raw.begin() + (ptrdiff_t)(off + len));
            for (size_t i = 0; i < len;) {        // 5552 bytes keep both sums below 2^32
                const size_t run = std::min<size_t>(5552, len - i);
                for (size_t k = 0; k < run; ++k) { s1 += raw[off + i + k]; s2 += s1; }
                s1 %= 65521u; s2 %= 65521u;
                i += run;
            }
        }
        unsigned char adler[4];
        be32(adler, (s2 << 16) | s1);
        z.insert(z.end(), adler, adler + 4);
        ok = ok && chunk("IDAT", z.data(), z.size()) && chunk("IEND", nullptr, 0);
    }
    if (!ok) return fail(PT_ERR_IO, std::string("pt_film_write_image: short write to ") + path);
    return PT_OK;
}

} // extern "C"
