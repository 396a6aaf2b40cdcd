// C ABI: frame assembly across GPUs over RCCL (xGMI).
//
// The reference is single-device; its only read-back is readback_ssbo
// (src/main_granite.cpp:499-522).  Work shards with no exchange on the data path (scene
// replicated, pixels and samples independent), so the one collective is the framebuffer at the
// end of a frame: a sum of the ranks' films (sample sharding) or each rank's tiles sent straight
// to the root over its own direct link (tile sharding).
//
// RCCL is looked up at run time (dlopen) the first time a communicator is asked for: a
// single-GPU user of the library needs no RCCL installed, and a process that already has one
// loaded (PyTorch ships its own librccl.so.1) shares that copy instead of getting a second.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstring>

#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mipt.h"
#include "internal.h"

using namespace mipt;

namespace {

struct Rccl {
    void *handle = nullptr;
    std::string error;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
};

const Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            const char *e = dlerror();
            r.error = std::string("RCCL not found (librccl.so.1): ") + (e ? e : "");
            return;
        }
        bool ok = true;
        auto sym = [&](const char *name) {
            void *p = dlsym(r.handle, name);
            if (!p) {
                ok = false;
                r.error = std::string("RCCL lacks ") + name;
            }
            return p;
        };
#define PT_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(sym(name))
        PT_SYM(GetUniqueId, "ncclGetUniqueId");
        PT_SYM(CommInitRank, "ncclCommInitRank");
        PT_SYM(CommInitAll, "ncclCommInitAll");
        PT_SYM(CommDestroy, "ncclCommDestroy");
        PT_SYM(GetErrorString, "ncclGetErrorString");
        PT_SYM(Reduce, "ncclReduce");
        PT_SYM(AllReduce, "ncclAllReduce");
        PT_SYM(AllGather, "ncclAllGather");
        PT_SYM(Send, "ncclSend");
        PT_SYM(Recv, "ncclRecv");
        PT_SYM(GroupStart, "ncclGroupStart");
        PT_SYM(GroupEnd, "ncclGroupEnd");
#undef PT_SYM
        if (!ok) {
            dlclose(r.handle);
            r.handle = nullptr;
        }
    });
    return r;
}

int nccl_fail(ncclResult_t e, const char *what) {
    const Rccl &r = rccl();
    return fail(PT_ERR_DEVICE, std::string(what) + ": " +
                                   (r.GetErrorString ? r.GetErrorString(e) : "RCCL error"));
}

#define NCCL_CHECK(expr)                                          \
    do {                                                          \
        ncclResult_t _e = (expr);                                 \
        if (_e != ncclSuccess) return nccl_fail(_e, #expr);       \
    } while (0)
#define HIP_CHECK(expr)                                           \
    do {                                                          \
        hipError_t _e = (expr);                                   \
        if (_e != hipSuccess) return hip_fail(_e, #expr);         \
    } while (0)

} // namespace

struct pt_comm {
    pt_context *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1;
    DeviceBuffer staging;       // packed tiles on their way out (peers) or in (root)
};

static_assert(PT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");

extern "C" {

int pt_comm_available(void) { return rccl().handle ? 1 : 0; }

int pt_comm_unique_id(void *id) {
    if (!id) return fail(PT_ERR_INVALID, "pt_comm_unique_id: null argument");
    const Rccl &r = rccl();
    if (!r.handle) return fail(PT_ERR_UNSUPPORTED, r.error);
    ncclUniqueId u;
    NCCL_CHECK(r.GetUniqueId(&u));
    std::memcpy(id, u.internal, PT_COMM_ID_BYTES);
    return PT_OK;
}

int pt_comm_create(pt_context *ctx, const void *id, int n_ranks, int rank, pt_comm **out) {
    if (!ctx || !id || !out) return fail(PT_ERR_INVALID, "pt_comm_create: null argument");
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return fail(PT_ERR_INVALID, "pt_comm_create: rank must be in [0, n_ranks)");
    const Rccl &r = rccl();
    if (!r.handle) return fail(PT_ERR_UNSUPPORTED, r.error);
    HIP_CHECK(hipSetDevice(ctx->device));
    ncclUniqueId u;
    std::memcpy(u.internal, id, PT_COMM_ID_BYTES);
    auto c = std::make_unique<pt_comm>();
    c->ctx = ctx;
    c->rank = rank;
    c->size = n_ranks;
    NCCL_CHECK(r.CommInitRank(&c->comm, n_ranks, u, rank));
    ctx->live_comms.fetch_add(1);
    *out = c.release();
    return PT_OK;
}

int pt_comm_create_all(pt_context *const *ctxs, int n, pt_comm **out) {
    if (!ctxs || !out || n < 1) return fail(PT_ERR_INVALID, "pt_comm_create_all: bad argument");
    for (int i = 0; i < n; ++i) out[i] = nullptr;
    const Rccl &r = rccl();
    if (!r.handle) return fail(PT_ERR_UNSUPPORTED, r.error);
    std::vector<int> devs((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return fail(PT_ERR_INVALID, "pt_comm_create_all: null context");
        devs[(size_t)i] = ctxs[i]->device;
        for (int j = 0; j < i; ++j)
            if (devs[(size_t)j] == devs[(size_t)i])
                return fail(PT_ERR_INVALID, "pt_comm_create_all: two contexts on one GPU");
    }
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    NCCL_CHECK(r.CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) {
        pt_comm *c = new pt_comm;
        c->ctx = ctxs[i];
        c->comm = comms[(size_t)i];
        c->rank = i;
        c->size = n;
        ctxs[i]->live_comms.fetch_add(1);
        out[i] = c;
    }
    return PT_OK;
}

int pt_comm_rank(const pt_comm *comm) { return comm ? comm->rank : -1; }
int pt_comm_size(const pt_comm *comm) { return comm ? comm->size : 0; }

int pt_comm_destroy(pt_comm *comm) {
    if (!comm) return PT_OK;
    (void)hipSetDevice(comm->ctx->device);
    (void)hipStreamSynchronize(comm->ctx->stream);
    if (comm->comm) (void)rccl().CommDestroy(comm->comm);
    comm->staging.release();
    comm->ctx->live_comms.fetch_sub(1);
    delete comm;
    return PT_OK;
}

int pt_film_reduce(pt_film *film, pt_comm *comm, int root) {
    if (!film || !comm) return fail(PT_ERR_INVALID, "pt_film_reduce: null argument");
    if (film->ctx != comm->ctx)
        return fail(PT_ERR_INVALID, "pt_film_reduce: film and communicator belong to different contexts");
    if (root >= comm->size) return fail(PT_ERR_INVALID, "pt_film_reduce: root out of range");
    const Rccl &r = rccl();
    HIP_CHECK(hipSetDevice(film->ctx->device));
    const size_t count = 4 * (size_t)film->width * film->height;
    hipStream_t st = film->ctx->stream;
    if (root < 0)
        NCCL_CHECK(r.AllReduce(film->data, film->data, count, ncclFloat, ncclSum, comm->comm, st));
    else
        NCCL_CHECK(r.Reduce(film->data, film->data, count, ncclFloat, ncclSum, root, comm->comm, st));
    return PT_OK;
}

int pt_film_gather_tiles(pt_film *film, pt_comm *comm, uint32_t tile_size, int root) {
    if (!film || !comm) return fail(PT_ERR_INVALID, "pt_film_gather_tiles: null argument");
    if (film->ctx != comm->ctx)
        return fail(PT_ERR_INVALID, "pt_film_gather_tiles: film and communicator belong to different contexts");
    if (root >= comm->size) return fail(PT_ERR_INVALID, "pt_film_gather_tiles: root out of range");
    if (comm->size == 1) return PT_OK;          // the only rank holds every tile already
    const Rccl &r = rccl();
    HIP_CHECK(hipSetDevice(film->ctx->device));
    hipStream_t st = film->ctx->stream;
    const uint32_t world = (uint32_t)comm->size, me = (uint32_t)comm->rank;
    // float4 slots of every rank's packed tiles (every rank can work them all out)
    std::vector<uint64_t> slots(world);
    uint64_t most = 0, total = 0;
    for (uint32_t k = 0; k < world; ++k) {
        slots[k] = pt_film_packed_size(film, tile_size, k, world);
        most = std::max(most, slots[k]);
        total += slots[k];
    }
    if (total == 0) return fail(PT_ERR_INVALID, "pt_film_gather_tiles: bad tile size");
    int rc;
    if (root < 0) {
        // every rank ends up with the whole frame: one all-gather of equal (padded) slices
        rc = comm->staging.reserve(16 * most * ((size_t)world + 1));
        if (rc != PT_OK) return rc;
        float *mine = (float *)comm->staging.ptr;
        float *all = mine + 4 * most;
        if (slots[me]) {
            rc = pt_film_pack_tiles(film, tile_size, me, world, mine);
            if (rc != PT_OK) return rc;
        }
        NCCL_CHECK(r.AllGather(mine, all, 4 * most, ncclFloat, comm->comm, st));
        for (uint32_t k = 0; k < world; ++k)
            if (k != me && slots[k]) {
                rc = pt_film_unpack_tiles(film, tile_size, k, world, all + 4 * most * k);
                if (rc != PT_OK) return rc;
            }
        return PT_OK;
    }
    if ((int)me != root) {
        // a peer: its tiles go straight to the root, over the link the two share
        if (slots[me] == 0) return PT_OK;
        rc = comm->staging.reserve(16 * slots[me]);
        if (rc != PT_OK) return rc;
        rc = pt_film_pack_tiles(film, tile_size, me, world, (float *)comm->staging.ptr);
        if (rc != PT_OK) return rc;
        NCCL_CHECK(r.Send(comm->staging.ptr, 4 * slots[me], ncclFloat, root, comm->comm, st));
        return PT_OK;
    }
    // the root: all peers' slices arrive side by side in one grouped operation
    rc = comm->staging.reserve(16 * (total - slots[me]));
    if (rc != PT_OK) return rc;
    float *base = (float *)comm->staging.ptr;
    NCCL_CHECK(r.GroupStart());
    {
        size_t off = 0;
        for (uint32_t k = 0; k < world; ++k) {
            if (k == me || slots[k] == 0) continue;
            const ncclResult_t e = r.Recv(base + off, 4 * slots[k], ncclFloat, (int)k, comm->comm, st);
            if (e != ncclSuccess) {
                (void)r.GroupEnd();
                return nccl_fail(e, "ncclRecv");
            }
            off += 4 * slots[k];
        }
    }
    NCCL_CHECK(r.GroupEnd());
    size_t off = 0;
    for (uint32_t k = 0; k < world; ++k) {
        if (k == me || slots[k] == 0) continue;
        rc = pt_film_unpack_tiles(film, tile_size, k, world, base + off);
        if (rc != PT_OK) return rc;
        off += 4 * slots[k];
    }
    return PT_OK;
}

} // extern "C"
