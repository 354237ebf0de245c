// vsgpu_comm.hip -- the shard exchange of a multi-GPU Flat index: RCCL over xGMI.
//
// One process per GPU.  Per query batch every rank contributes one fixed-size record of per-query candidate
// lists (tens of KB) and ONE ncclAllGather hands every rank all of them; deletes broadcast the moved row from
// its owner.  Nothing else crosses GPUs.  librccl is opened on first use (dlopen), so single-GPU users of
// libvsgpu.so carry no RCCL dependency and a process that already holds an RCCL (e.g. torch's) shares it.
#include <dlfcn.h>

#include "vsgpu_internal.hpp"

namespace {
// the slice of rccl.h this file uses (ABI of RCCL 2.x: rccl/rccl.h:40-43, 187, 220, 260, 339, 459-470, 591, 678)
struct RcclUniqueId {
    char internal[VSGPU_COMM_ID_BYTES];
};
typedef void *RcclComm;
enum { kRcclSuccess = 0, kRcclInt8 = 0 };
struct Rccl {
    void *so = nullptr;
    int (*GetUniqueId)(RcclUniqueId *) = nullptr;
    int (*CommInitRank)(RcclComm *, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, RcclComm, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl *rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
        }
        if (r.so) {
            r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.so, "ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.so, "ncclCommInitRank");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
            r.AllGather = (decltype(r.AllGather))dlsym(r.so, "ncclAllGather");
            r.Broadcast = (decltype(r.Broadcast))dlsym(r.so, "ncclBroadcast");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.Broadcast) r.so = nullptr;
        }
    }
    return r.so ? &r : nullptr;
}
const char *rccl_err(int rc) {
    Rccl *r = rccl();
    return (r && r->GetErrorString) ? r->GetErrorString(rc) : "?";
}
}  // namespace

struct vsgpu_comm {
    vsgpu_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    RcclComm comm = nullptr;
    hipStream_t stream = nullptr;
    void *d_send = nullptr, *d_recv = nullptr, *h_stage = nullptr;
    size_t send_cap = 0, recv_cap = 0, stage_cap = 0;
};

#define RCCLCHK(expr)                                                                                  \
    do {                                                                                               \
        int _rc = (expr);                                                                              \
        if (_rc != kRcclSuccess) return fail(VSGPU_ERR_HIP, "%s failed: %s", #expr, rccl_err(_rc));  \
    } while (0)

extern "C" int vsgpu_comm_unique_id(void *id128) {
    Rccl *r = rccl();
    if (!r) return fail(VSGPU_ERR_UNSUPPORTED, "librccl.so could not be opened: %s", dlerror());
    RcclUniqueId id;
    RCCLCHK(r->GetUniqueId(&id));
    memcpy(id128, id.internal, VSGPU_COMM_ID_BYTES);
    return VSGPU_OK;
}

extern "C" vsgpu_comm *vsgpu_comm_create(vsgpu_ctx *ctx, int rank, int world, const void *id128) {
    Rccl *r = rccl();
    if (!r) {
        fail(VSGPU_ERR_UNSUPPORTED, "librccl.so could not be opened: %s", dlerror());
        return nullptr;
    }
    if (!ctx || world < 1 || rank < 0 || rank >= world || !id128) {
        fail(VSGPU_ERR_ARG, "bad communicator arguments (rank %d of %d)", rank, world);
        return nullptr;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) {
        fail(VSGPU_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
        return nullptr;
    }
    vsgpu_comm *c = new vsgpu_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    RcclUniqueId id;
    memcpy(id.internal, id128, VSGPU_COMM_ID_BYTES);
    int rc = r->CommInitRank(&c->comm, world, id, rank);
    if (rc != kRcclSuccess) {
        fail(VSGPU_ERR_HIP, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, ctx->device, rccl_err(rc));
        delete c;
        return nullptr;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        fail(VSGPU_ERR_HIP, "stream creation failed");
        r->CommDestroy(c->comm);
        delete c;
        return nullptr;
    }
    return c;
}

extern "C" void vsgpu_comm_destroy(vsgpu_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) rccl()->CommDestroy(c->comm);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
extern "C" int vsgpu_comm_rank(const vsgpu_comm *c) { return c->rank; }
extern "C" int vsgpu_comm_world(const vsgpu_comm *c) { return c->world; }

static int comm_reserve(vsgpu_comm *c, size_t send_bytes, size_t recv_bytes) {
    auto grow = [](void *&p, size_t &cap, size_t need) -> hipError_t {
        if (need <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = (need + 0xFFFF) & ~(size_t)0xFFFF;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    };
    HIPCHK(grow(c->d_send, c->send_cap, send_bytes));
    HIPCHK(grow(c->d_recv, c->recv_cap, recv_bytes));
    const size_t stage = send_bytes + recv_bytes;
    if (stage > c->stage_cap) {
        if (c->h_stage) HIPCHK(hipHostFree(c->h_stage));
        c->h_stage = nullptr;
        c->stage_cap = 0;
        const size_t want = (stage + 0xFFFF) & ~(size_t)0xFFFF;
        HIPCHK(hipHostMalloc(&c->h_stage, want, hipHostMallocDefault));
        c->stage_cap = want;
    }
    return VSGPU_OK;
}

extern "C" int vsgpu_comm_allgather(vsgpu_comm *c, const void *send, size_t bytes, void *recv) {
    if (bytes == 0) return VSGPU_OK;
    HIPCHK(hipSetDevice(c->ctx->device));
    const size_t total = bytes * (size_t)c->world;
    int rc = comm_reserve(c, bytes, total);
    if (rc) return rc;
    char *hs = (char *)c->h_stage, *hr = hs + bytes;
    memcpy(hs, send, bytes);
    HIPCHK(hipMemcpyAsync(c->d_send, hs, bytes, hipMemcpyHostToDevice, c->stream));
    RCCLCHK(rccl()->AllGather(c->d_send, c->d_recv, bytes, kRcclInt8, c->comm, c->stream));
    HIPCHK(hipMemcpyAsync(hr, c->d_recv, total, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(recv, hr, total);
    return VSGPU_OK;
}

extern "C" int vsgpu_comm_broadcast(vsgpu_comm *c, void *buf, size_t bytes, int root) {
    if (bytes == 0) return VSGPU_OK;
    if (root < 0 || root >= c->world) return fail(VSGPU_ERR_ARG, "broadcast root %d of %d", root, c->world);
    HIPCHK(hipSetDevice(c->ctx->device));
    int rc = comm_reserve(c, bytes, bytes);
    if (rc) return rc;
    char *hs = (char *)c->h_stage;
    if (c->rank == root) {
        memcpy(hs, buf, bytes);
        HIPCHK(hipMemcpyAsync(c->d_send, hs, bytes, hipMemcpyHostToDevice, c->stream));
    }
    RCCLCHK(rccl()->Broadcast(c->d_send, c->d_send, bytes, kRcclInt8, root, c->comm, c->stream));
    if (c->rank != root) {
        HIPCHK(hipMemcpyAsync(hs, c->d_send, bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        memcpy(buf, hs, bytes);
    } else {
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return VSGPU_OK;
}
