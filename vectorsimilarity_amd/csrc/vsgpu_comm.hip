// vsgpu_comm.hip -- the shard exchange of a multi-GPU Flat index: RCCL over xGMI.
//
// One process per GPU.  Per query batch every rank contributes one fixed-size record of per-query candidate
// lists (tens of KB) and ONE ncclAllGather hands every rank all of them; deletes broadcast the moved row from
// its owner.  Nothing else crosses GPUs.  librccl is opened on first use (dlopen), so single-GPU users of
// libvsgpu.so carry no RCCL dependency and a process that already holds an RCCL (e.g. torch's) shares it.
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <mutex>

#include "vsgpu_internal.hpp"

namespace {
// the slice of rccl.h this file uses (ABI of RCCL 2.x: rccl/rccl.h:40-43, 187, 220, 260, 339, 459-470, 591, 678; ncclCommAbort,
// ncclCommGetAsyncError)
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
    int (*CommAbort)(RcclComm) = nullptr;
    int (*CommGetAsyncError)(RcclComm, int *) = nullptr;
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
            r.CommAbort = (decltype(r.CommAbort))dlsym(r.so, "ncclCommAbort");                        // (optional: old RCCLs lack them)
            r.CommGetAsyncError = (decltype(r.CommGetAsyncError))dlsym(r.so, "ncclCommGetAsyncError");
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
    // send / receive buffers are host memory mapped into the device (hipHostMalloc): the caller's record is written where the
    // collective reads it and read where the collective wrote it.  Round 3 staged through device buffers -- host -> pinned -> H2D
    // -> ncclAllGather -> D2H -> host -- and every one of those copies queued behind the scan another reader had running; the
    // records are tens of KB, so the PCIe hop inside the collective costs nothing next to that.
    void *h_send = nullptr, *h_recv = nullptr;
    size_t send_cap = 0, recv_cap = 0;
    // Communicators of more than one rank use device buffers between the mapped host blocks and the collective by default
    // (the canonical RCCL usage; mapped host memory as the collective's own buffers is measured on one rank only, where it
    // removes the copies from behind the scans).  VECSIM_GPU_EXCHANGE=mapped | staged overrides.
    bool staged = false;
    void *d_send = nullptr, *d_recv = nullptr;
    size_t dsend_cap = 0, drecv_cap = 0;
    // A collective that failed on this rank (RCCL error, HIP error, the peers not arriving within the time limit) ABORTS the
    // communicator: a rank that merely returned an error would leave its peers inside the collective for ever.  Their kernels see
    // the abort (or their own time limit) and every rank comes back with an error.  A dead communicator refuses further calls.
    // Threads: `mu` is held for the whole of a collective (one in flight per communicator).  vsgpu_comm_abort from ANOTHER thread
    // only raises `abort_req` and then waits for `mu`: the thread inside the collective sees the flag in comm_wait and performs the
    // abort itself (ncclCommAbort frees the communicator, so nobody may still be inside ncclCommGetAsyncError with it); a collective
    // that had already left comm_wait completes, and the abort runs behind it.
    std::atomic<bool> dead{false};
    std::atomic<bool> abort_req{false};
    std::mutex mu;
    long timeout_ms = 0;         // VECSIM_GPU_EXCHANGE_TIMEOUT_MS (default 120 s; 0 = wait for ever)
    uint64_t collectives = 0;    // issued so far
    long fail_at = -1;           // VECSIM_GPU_EXCHANGE_FAIL_AT = n: the n-th collective reports an RCCL failure (test hook)
};
static int comm_abort(vsgpu_comm *c, const char *why) {
    if (!c->dead.exchange(true)) {
        Rccl *r = rccl();
        if (c->comm && r && r->CommAbort) {
            (void)r->CommAbort(c->comm);   // frees the communicator and releases kernels waiting inside it
            c->comm = nullptr;
        }
    }
    return fail(VSGPU_ERR_HIP, "shard exchange failed on rank %d of %d (%s): communicator aborted", c->rank, c->world, why);
}
// waits for the collective's stream: like hipStreamSynchronize, but a peer that never arrives (it failed and aborted, or died)
// shows up as an asynchronous RCCL error or as the time limit instead of a hang
static int comm_wait(vsgpu_comm *c) {
    Rccl *r = rccl();
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spin = 1;; spin++) {
        const hipError_t e = hipStreamQuery(c->stream);
        // (a stream that drained because the communicator was aborted under it also reports success: the receive block then holds
        // whatever was there before -- never hand that back as a result)
        if (e == hipSuccess) return (c->dead || c->abort_req) ? comm_abort(c, "aborted while the exchange was in flight") : VSGPU_OK;
        if (e != hipErrorNotReady) return comm_abort(c, hipGetErrorString(e));
        if ((spin & 0x3FF) != 0) continue;
        if (c->abort_req) return comm_abort(c, "aborted by the caller");
        int async = kRcclSuccess;
        if (r->CommGetAsyncError && c->comm && r->CommGetAsyncError(c->comm, &async) == kRcclSuccess && async != kRcclSuccess)
            return comm_abort(c, rccl_err(async));
        if (c->timeout_ms > 0 &&
            std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > c->timeout_ms)
            return comm_abort(c, "peers did not arrive within VECSIM_GPU_EXCHANGE_TIMEOUT_MS");
    }
}

#define RCCLCHK(expr)                                                                                  \
    do {                                                                                               \
        int _rc = (expr);                                                                              \
        if (_rc != kRcclSuccess) return fail(VSGPU_ERR_HIP, "%s failed: %s", #expr, rccl_err(_rc));  \
    } while (0)
// inside a collective: a failure aborts the communicator (comm_abort), so that the peers come back too
#define COMMCHK_RCCL(c, expr)                                          \
    do {                                                               \
        int _rc = (expr);                                              \
        if (_rc != kRcclSuccess) return comm_abort(c, rccl_err(_rc)); \
    } while (0)
#define COMMCHK_HIP(c, expr)                                               \
    do {                                                                   \
        hipError_t _e = (expr);                                            \
        if (_e != hipSuccess) return comm_abort(c, hipGetErrorString(_e)); \
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
    c->staged = world > 1;
    if (const char *e = getenv("VECSIM_GPU_EXCHANGE")) c->staged = !strcmp(e, "staged") ? true : (!strcmp(e, "mapped") ? false : c->staged);
    c->timeout_ms = 120000;
    if (const char *e = getenv("VECSIM_GPU_EXCHANGE_TIMEOUT_MS")) c->timeout_ms = atol(e);
    if (const char *e = getenv("VECSIM_GPU_EXCHANGE_FAIL_AT")) c->fail_at = atol(e);
    // highest priority the device offers: the collective's kernel is dispatched ahead of whatever else is waiting for a CU
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi) != hipSuccess) {
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
    if (c->h_send) (void)hipHostFree(c->h_send);
    if (c->h_recv) (void)hipHostFree(c->h_recv);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
extern "C" int vsgpu_comm_abort(vsgpu_comm *c) {
    if (!c) return VSGPU_OK;
    c->abort_req = true;                       // an exchange in flight sees it in comm_wait and aborts on its own thread
    std::lock_guard<std::mutex> lk(c->mu);     // ... and has left the communicator when this lock is granted
    (void)hipSetDevice(c->ctx->device);
    (void)comm_abort(c, "aborted by the caller");
    return VSGPU_OK;
}
extern "C" int vsgpu_comm_staged(const vsgpu_comm *c) { return c->staged ? 1 : 0; }
extern "C" int vsgpu_comm_rank(const vsgpu_comm *c) { return c->rank; }
extern "C" int vsgpu_comm_world(const vsgpu_comm *c) { return c->world; }

static int comm_reserve(vsgpu_comm *c, size_t send_bytes, size_t recv_bytes) {
    auto grow = [](void *&p, size_t &cap, size_t need) -> hipError_t {
        if (need <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = (need + 0xFFFF) & ~(size_t)0xFFFF;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocMapped);
        if (e == hipSuccess) cap = want;
        return e;
    };
    HIPCHK(grow(c->h_send, c->send_cap, send_bytes));
    HIPCHK(grow(c->h_recv, c->recv_cap, recv_bytes));
    if (c->staged) {
        auto dgrow = [](void *&p, size_t &cap, size_t need) -> hipError_t {
            if (need <= cap) return hipSuccess;
            if (p) (void)hipFree(p);
            p = nullptr;
            cap = 0;
            const size_t want = (need + 0xFFFF) & ~(size_t)0xFFFF;
            hipError_t e = hipMalloc(&p, want);
            if (e == hipSuccess) cap = want;
            return e;
        };
        HIPCHK(dgrow(c->d_send, c->dsend_cap, send_bytes));
        HIPCHK(dgrow(c->d_recv, c->drecv_cap, recv_bytes));
    }
    return VSGPU_OK;
}

// entry of every collective: a dead communicator refuses; the test hook's chosen collective fails the way an RCCL call would
static int comm_enter(vsgpu_comm *c) {
    if (c->abort_req && !c->dead) return comm_abort(c, "aborted by the caller");
    if (c->dead) return fail(VSGPU_ERR_HIP, "shard exchange: the communicator of rank %d was aborted by an earlier failure", c->rank);
    if (c->fail_at >= 0 && (long)c->collectives == c->fail_at) {
        c->collectives++;
        return comm_abort(c, "VECSIM_GPU_EXCHANGE_FAIL_AT (test hook)");
    }
    c->collectives++;
    return VSGPU_OK;
}

extern "C" int vsgpu_comm_allgather(vsgpu_comm *c, const void *send, size_t bytes, void *recv) {
    if (bytes == 0) return VSGPU_OK;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->ctx->device));
    int rc = comm_enter(c);
    if (rc) return rc;
    const size_t total = bytes * (size_t)c->world;
    rc = comm_reserve(c, bytes, total);
    if (rc) return comm_abort(c, "buffer allocation");
    memcpy(c->h_send, send, bytes);
    if (c->staged) {
        COMMCHK_HIP(c, hipMemcpyAsync(c->d_send, c->h_send, bytes, hipMemcpyHostToDevice, c->stream));
        COMMCHK_RCCL(c, rccl()->AllGather(c->d_send, c->d_recv, bytes, kRcclInt8, c->comm, c->stream));
        COMMCHK_HIP(c, hipMemcpyAsync(c->h_recv, c->d_recv, total, hipMemcpyDeviceToHost, c->stream));
    } else {
        COMMCHK_RCCL(c, rccl()->AllGather(c->h_send, c->h_recv, bytes, kRcclInt8, c->comm, c->stream));
    }
    rc = comm_wait(c);
    if (rc) return rc;
    if (c->dead) return fail(VSGPU_ERR_HIP, "shard exchange: the communicator of rank %d was aborted", c->rank);
    memcpy(recv, c->h_recv, total);
    return VSGPU_OK;
}

extern "C" int vsgpu_comm_broadcast(vsgpu_comm *c, void *buf, size_t bytes, int root) {
    if (bytes == 0) return VSGPU_OK;
    if (root < 0 || root >= c->world) return fail(VSGPU_ERR_ARG, "broadcast root %d of %d", root, c->world);
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->ctx->device));
    int rc = comm_enter(c);
    if (rc) return rc;
    rc = comm_reserve(c, bytes, bytes);
    if (rc) return comm_abort(c, "buffer allocation");
    if (c->rank == root) memcpy(c->h_send, buf, bytes);
    if (c->staged) {
        if (c->rank == root) COMMCHK_HIP(c, hipMemcpyAsync(c->d_send, c->h_send, bytes, hipMemcpyHostToDevice, c->stream));
        COMMCHK_RCCL(c, rccl()->Broadcast(c->d_send, c->d_send, bytes, kRcclInt8, root, c->comm, c->stream));
        if (c->rank != root) COMMCHK_HIP(c, hipMemcpyAsync(c->h_send, c->d_send, bytes, hipMemcpyDeviceToHost, c->stream));
    } else {
        COMMCHK_RCCL(c, rccl()->Broadcast(c->h_send, c->h_send, bytes, kRcclInt8, root, c->comm, c->stream));
    }
    rc = comm_wait(c);
    if (rc) return rc;
    if (c->rank != root) memcpy(buf, c->h_send, bytes);
    return VSGPU_OK;
}
