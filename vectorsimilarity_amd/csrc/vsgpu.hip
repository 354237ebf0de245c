// vsgpu.hip -- implementation of include/vsgpu.h: device table + query orchestration (gfx950).
//
// No CPU fallback lives here: if there is no HIP device every entry point fails with
// VSGPU_ERR_NO_DEVICE and a message.  Host code in this file only moves bytes, builds lane tables,
// launches kernels and selects among *GPU-computed* scores.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <chrono>
#include <string>
#include <vector>

#include "vsgpu.h"
#include "lane_program.h"
#include "exact_kernels.hpp"
#include "mfma_kernels.hpp"
#include "mfma_lowp_kernels.hpp"
#include "mfma_free_kernels.hpp"
#include "hnsw_kernels.hpp"
#include "iter_kernels.hpp"

using namespace vsg;

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return fail(_e == hipErrorOutOfMemory ? VSGPU_ERR_OOM : VSGPU_ERR_HIP, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(_e), __FILE__, __LINE__);                        \
    } while (0)

extern "C" const char *vsgpu_last_error(void) { return g_err.c_str(); }

extern "C" int vsgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// ------------------------------------------------------------------ context
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct vsgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_d = nullptr;
    DevBuf qperm, qnorm, dense, tau, counts, cand, ids, qfrag, qn2, sel, selcnt;
    void *pinned = nullptr;
    size_t pinned_cap = 0;
    vsgpu_stats stats{};
    // options
    long opt_mfma = 1;
    long opt_mfma_variant = 0;
    long opt_lowp_variant = 0;
    long opt_hnsw_slots = 16;  // resident search waves (= visited-tag slots) per CU: 8 -> 264 K QPS, 12-32 -> 314-319 K (200 K x 768)
    long opt_lowp_qsplit = 0;  // int8: 1 = two 128-query workgroups per row tile instead of one 256-query one
    long opt_lowp_dbg = 0;   // diagnosis only: bit0 skip epilogue, bit1 skip LDS reads + MFMA, bit2 skip row DMA
    long opt_wg_per_cu = 2;
    long opt_mfma_min_q = 1;          // batches narrower than this stay on the exact kernel.  Measured (tools/bench_small_batches.py):
                                      // the MFMA filter wins from one query up (10M x 768: 4.7 ms vs 5.8-8.5 ms for 1-8 queries)
    long opt_dense_pairs = 1L << 16;  // nq*n at or below this: one dense score matrix + one select kernel
    long opt_probe_div = 0;           // probe ~ n / probe_div rows; 0 = chosen per call by probe_divisor()
    long opt_probe_cap = 32768;       // ... but at most this many probe tiles
    long opt_cand_cap = 8192;         // candidate slots per query
    int n_cu = 256;
};

// VSGPU_POISON=<byte>: fill every fresh device allocation with that byte (test aid: makes any read of memory the
// library never wrote deterministic instead of depending on what the allocator hands back)
static int poison_byte() {
    static const int v = [] {
        const char *e = getenv("VSGPU_POISON");
        return e ? (int)(strtol(e, nullptr, 0) & 0xFF) : -1;
    }();
    return v;
}
static void poison(void *p, size_t bytes) {
    if (poison_byte() >= 0 && p) {
        (void)hipMemset(p, poison_byte(), bytes);
        (void)hipDeviceSynchronize();
    }
}
// VSGPU_TIMING=1: host wall-clock marks of a top-k call on stderr (where the non-kernel time goes)
struct WallMarks {
    bool on = getenv("VSGPU_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    std::string out;
    void mark(const char *what) {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        char b[96];
        snprintf(b, sizeof b, " %s=%.3f", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        out += b;
        t0 = t1;
    }
    void flush(const char *tag) {
        if (on) fprintf(stderr, "VSGPU_TIMING %s:%s\n", tag, out.c_str());
    }
};
static int ensure(vsgpu_ctx *c, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return VSGPU_OK;
    if (b.p) HIPCHK(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = std::max(bytes, (size_t)4096);
    want = (want + 0xFFFF) & ~(size_t)0xFFFF;
    HIPCHK(hipMalloc(&b.p, want));
    poison(b.p, want);
    b.cap = want;
    return VSGPU_OK;
}
static int ensure_pinned(vsgpu_ctx *c, size_t bytes) {
    if (bytes <= c->pinned_cap) return VSGPU_OK;
    if (c->pinned) {
        HIPCHK(hipStreamSynchronize(c->stream));  // an async copy may still read the old staging buffer
        HIPCHK(hipHostFree(c->pinned));
    }
    c->pinned = nullptr;
    c->pinned_cap = 0;
    size_t want = (std::max(bytes, (size_t)1 << 20) + 0xFFFF) & ~(size_t)0xFFFF;
    HIPCHK(hipHostMalloc(&c->pinned, want, hipHostMallocDefault));
    if (poison_byte() >= 0) memset(c->pinned, poison_byte(), want);
    c->pinned_cap = want;
    return VSGPU_OK;
}

extern "C" vsgpu_ctx *vsgpu_ctx_create(int device) {
    int n = vsgpu_device_count();
    if (n <= 0) {
        fail(VSGPU_ERR_NO_DEVICE, "no HIP device visible: the gfx950 kernels are the only compute path");
        return nullptr;
    }
    if (device < 0 || device >= n) {
        fail(VSGPU_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        fail(VSGPU_ERR_HIP, "hipSetDevice(%d) failed", device);
        return nullptr;
    }
    vsgpu_ctx *c = new vsgpu_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev_a) != hipSuccess || hipEventCreate(&c->ev_b) != hipSuccess ||
        hipEventCreate(&c->ev_c) != hipSuccess || hipEventCreate(&c->ev_d) != hipSuccess) {
        fail(VSGPU_ERR_HIP, "stream/event creation failed on device %d", device);
        delete c;
        return nullptr;
    }
    return c;
}

extern "C" void vsgpu_ctx_destroy(vsgpu_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (DevBuf *b : {&c->qperm, &c->qnorm, &c->dense, &c->tau, &c->counts, &c->cand, &c->ids, &c->qfrag, &c->qn2, &c->sel, &c->selcnt})
        if (b->p) (void)hipFree(b->p);
    if (c->pinned) (void)hipHostFree(c->pinned);
    (void)hipEventDestroy(c->ev_a);
    (void)hipEventDestroy(c->ev_b);
    (void)hipEventDestroy(c->ev_c);
    (void)hipEventDestroy(c->ev_d);
    (void)hipStreamDestroy(c->stream);
    delete c;
}
extern "C" int vsgpu_ctx_device(const vsgpu_ctx *c) { return c ? c->device : -1; }
extern "C" int vsgpu_ctx_sync(vsgpu_ctx *c) {
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return VSGPU_OK;
}
extern "C" void vsgpu_stats_reset(vsgpu_ctx *c) {
    char name[64];
    memcpy(name, c->stats.scan_kernel, sizeof name);
    c->stats = vsgpu_stats{};
    memcpy(c->stats.scan_kernel, name, sizeof name);
}
extern "C" void vsgpu_stats_get(vsgpu_ctx *c, vsgpu_stats *out) { *out = c->stats; }
extern "C" int vsgpu_set_option(vsgpu_ctx *c, const char *name, long value) {
    std::string n(name);
    if (n == "mfma") c->opt_mfma = value;
    else if (n == "mfma_variant") c->opt_mfma_variant = value;
    else if (n == "lowp_variant") c->opt_lowp_variant = value;
    else if (n == "lowp_dbg") c->opt_lowp_dbg = value;
    else if (n == "hnsw_slots") c->opt_hnsw_slots = std::max(1L, std::min(64L, value));
    else if (n == "lowp_qsplit") c->opt_lowp_qsplit = value;
    else if (n == "wg_per_cu") c->opt_wg_per_cu = std::max(1L, value);
    else if (n == "mfma_min_q") c->opt_mfma_min_q = std::max(1L, value);
    else if (n == "dense_pairs") c->opt_dense_pairs = value;
    else if (n == "probe_div") c->opt_probe_div = std::max(0L, value);
    else if (n == "probe_cap") c->opt_probe_cap = std::min(1L << 20, std::max(64L, value));
    else if (n == "cand_cap") c->opt_cand_cap = std::max(16L, value);
    else return fail(VSGPU_ERR_ARG, "unknown option %s", name);
    return VSGPU_OK;
}

// ------------------------------------------------------------------ table
struct vsgpu_table {
    vsgpu_ctx *ctx = nullptr;
    int type = 0, metric = 0, tier = 0;
    size_t dim = 0, row_bytes = 0;
    LaneProgram prog;
    int32_t *d_offs = nullptr;
    std::vector<char *> slabs;
    char **d_slabs = nullptr;
    size_t d_slabs_cap = 0;
    uint32_t slab_shift = 0;
    size_t n = 0;
    int ek = 0, opk = 0, epi = 0;
    int bt_max = 1;  // largest query tile whose LDS image fits
    // MFMA filter path (fp32, AVX-512-order tier, dim a multiple of 64): |x|^2 per row, slab-parallel
    bool mfma_ok = false;
    int ksteps = 0;
    // low-precision MFMA filter (bf16/fp16/int8 rows): kernel shape picked at create time
    bool lowp_ok = false;
    int lp_kind = 0, lp_ksteps = 0, lp_rt = 0, lp_qtile = 0;
    std::vector<float *> norm_slabs;
    float **d_norm_slabs = nullptr;
};

static size_t acc_bytes(int type) { return type == VSGPU_F64 ? 8 : 4; }

extern "C" vsgpu_table *vsgpu_table_create(vsgpu_ctx *c, int type, int metric, int tier, size_t dim,
                                           size_t row_bytes) {
    if (!c) {
        fail(VSGPU_ERR_ARG, "null context");
        return nullptr;
    }
    if (type < VSGPU_F32 || type > VSGPU_U8 || metric < VSGPU_L2 || metric > VSGPU_COSINE || dim == 0 ||
        row_bytes < dim * (size_t)elem_bytes_of(type)) {
        fail(VSGPU_ERR_ARG, "bad table parameters (type %d metric %d dim %zu row_bytes %zu)", type, metric,
             dim, row_bytes);
        return nullptr;
    }
    bool is_int = (type == VSGPU_I8 || type == VSGPU_U8);
    if (metric == VSGPU_COSINE && !is_int) metric = VSGPU_IP;  // fp Cosine == IP kernel on normalised blobs
    vsgpu_table *t = new vsgpu_table();
    t->ctx = c;
    t->type = type;
    t->metric = metric;
    t->tier = tier;
    t->dim = dim;
    t->row_bytes = row_bytes;
    t->prog = build_lane_program(type, metric == VSGPU_L2 ? VSGPU_L2 : VSGPU_IP, tier, dim);
    t->ek = type;  // ElemKind values equal the type codes
    bool l2 = (metric == VSGPU_L2);
    t->opk = t->prog.fused ? (l2 ? OP_L2_FMA : OP_IP_FMA) : (l2 ? OP_L2_MULADD : OP_IP_MULADD);
    if (is_int) t->epi = l2 ? EPI_INT_L2 : (metric == VSGPU_IP ? EPI_INT_IP : EPI_INT_COS);
    else t->epi = l2 ? EPI_L2 : EPI_ONE_MINUS;
    // LDS budget 64 KiB: offs + BT query images
    size_t offs_b = ((size_t)t->prog.steps * t->prog.vl * 4 + 15) & ~(size_t)15;
    size_t q_b = (size_t)t->prog.steps * t->prog.vl * acc_bytes(type);
    size_t budget = 64 * 1024;
    if (offs_b + q_b > budget) {
        fail(VSGPU_ERR_UNSUPPORTED, "dim %zu too large for the table-driven kernel's LDS image", dim);
        delete t;
        return nullptr;
    }
    size_t fit = (budget - offs_b) / q_b;
    t->bt_max = fit >= 8 ? 8 : (fit >= 4 ? 4 : 1);
    if (!t->prog.fused) t->bt_max = 1;  // scalar-tier variants are only instantiated for BT=1
    {
        // fp32 MFMA filter: any dim up to 3072.  The kernel instance is the next compiled width (k-steps of 32
        // elements); the columns past `dim` hold the start of the next row in LDS and zeros in the query fragments
        // (finite x 0 = 0; a NaN there only makes the filter pass the row on to the exact re-rank).
        static const int kInst[] = {4, 6, 8, 10, 12, 16, 20, 24, 28, 30, 32, 40, 48, 64, 80, 96};
        size_t ks = 0;
        for (int v : kInst)
            if ((size_t)v * 32 >= dim) {
                ks = (size_t)v;
                break;
            }
        t->mfma_ok = (type == VSGPU_F32 && !t->prog.scalar_tier && row_bytes == dim * 4 && ks != 0);
        t->ksteps = (int)ks;
        const size_t data_bytes = dim * (size_t)elem_bytes_of(type);
        // low-precision rows: like fp32, any dim runs at the next compiled width (bf16/fp16: 256, 512, 768, 1024,
        // 1536 elements; int8/uint8: 512, 768, 1024) with zero query columns past `dim`
        if (!t->prog.scalar_tier && (type == VSGPU_BF16 || type == VSGPU_F16) && tier != VSGPU_TIER_AVX512_BF16 &&
            dim <= 1536 && row_bytes == data_bytes) {
            static const int w16[] = {8, 16, 24, 32, 48};
            static const int rt16[] = {64, 32, 32, 16, 16};
            for (int i = 0; i < 5; i++)
                if ((size_t)w16[i] * 32 >= dim) {
                    t->lowp_ok = true;
                    t->lp_kind = type == VSGPU_BF16 ? LP_BF16 : LP_F16;
                    t->lp_ksteps = w16[i];
                    t->lp_rt = rt16[i];
                    t->lp_qtile = 128;
                    break;
                }
        }
        if ((type == VSGPU_I8 || (type == VSGPU_U8 && metric != VSGPU_COSINE)) && dim <= 1024) {
            t->lowp_ok = true;
            t->lp_kind = type == VSGPU_I8 ? LP_I8 : LP_U8;  // uint8 Cosine would need two aux values per row: exact path
            t->lp_ksteps = dim <= 512 ? 8 : (dim <= 768 ? 12 : 16);
            t->lp_rt = t->lp_ksteps == 16 ? 32 : 64;
            t->lp_qtile = 256;
        }
    }
    // slabs of ~64 MiB, power-of-two row count
    size_t rows = ((size_t)64 << 20) / row_bytes;
    uint32_t shift = 0;
    while (((size_t)2 << shift) <= rows) shift++;
    if (shift < 6) shift = 6;
    t->slab_shift = shift;
    if (hipSetDevice(c->device) != hipSuccess ||
        hipMalloc((void **)&t->d_offs, std::max<size_t>(16, t->prog.offs.size() * 4)) != hipSuccess ||
        hipMemcpy(t->d_offs, t->prog.offs.data(), t->prog.offs.size() * 4, hipMemcpyHostToDevice) !=
            hipSuccess) {
        fail(VSGPU_ERR_HIP, "lane table upload failed");
        delete t;
        return nullptr;
    }
    return t;
}

extern "C" void vsgpu_table_destroy(vsgpu_table *t) {
    if (!t) return;
    (void)hipSetDevice(t->ctx->device);
    (void)hipStreamSynchronize(t->ctx->stream);
    for (char *s : t->slabs) (void)hipFree(s);
    for (float *s : t->norm_slabs) (void)hipFree(s);
    if (t->d_slabs) (void)hipFree(t->d_slabs);
    if (t->d_norm_slabs) (void)hipFree(t->d_norm_slabs);
    if (t->d_offs) (void)hipFree(t->d_offs);
    delete t;
}
extern "C" size_t vsgpu_table_size(const vsgpu_table *t) { return t->n; }
extern "C" size_t vsgpu_table_bytes(const vsgpu_table *t) {
    return t->slabs.size() * (((size_t)1 << t->slab_shift) * t->row_bytes);
}

static int grow_to(vsgpu_table *t, size_t rows) {
    const size_t slab_rows = (size_t)1 << t->slab_shift;
    size_t need = (rows + slab_rows - 1) / slab_rows;
    bool changed = false;
    while (t->slabs.size() < need) {
        char *p = nullptr;
        // + slack: the MFMA filter reads a row out to its kernel width (< 128 extra floats past the last row)
        HIPCHK(hipMalloc((void **)&p, slab_rows * t->row_bytes + 1024));
        poison(p, slab_rows * t->row_bytes);
        t->slabs.push_back(p);
        if (t->mfma_ok || t->lowp_ok) {
            float *np = nullptr;
            HIPCHK(hipMalloc((void **)&np, slab_rows * sizeof(float)));
            poison(np, slab_rows * sizeof(float));
            t->norm_slabs.push_back(np);
        }
        changed = true;
    }
    if (changed) {
        if (t->slabs.size() > t->d_slabs_cap) {
            // the old pointer table may still be read by nothing: all launches are synchronised per call
            if (t->d_slabs) HIPCHK(hipFree(t->d_slabs));
            if (t->d_norm_slabs) HIPCHK(hipFree(t->d_norm_slabs));
            t->d_slabs = nullptr;
            t->d_norm_slabs = nullptr;
            size_t cap = std::max<size_t>(64, t->slabs.size() * 2);
            HIPCHK(hipMalloc((void **)&t->d_slabs, cap * sizeof(char *)));
            if (t->mfma_ok || t->lowp_ok) HIPCHK(hipMalloc((void **)&t->d_norm_slabs, cap * sizeof(float *)));
            t->d_slabs_cap = cap;
        }
        HIPCHK(hipMemcpyAsync(t->d_slabs, t->slabs.data(), t->slabs.size() * sizeof(char *), hipMemcpyHostToDevice,
                              t->ctx->stream));
        if (t->mfma_ok || t->lowp_ok)
            HIPCHK(hipMemcpyAsync(t->d_norm_slabs, t->norm_slabs.data(), t->norm_slabs.size() * sizeof(float *),
                                  hipMemcpyHostToDevice, t->ctx->stream));
        HIPCHK(hipStreamSynchronize(t->ctx->stream));
    }
    return VSGPU_OK;
}
static inline char *row_ptr(const vsgpu_table *t, size_t id) {
    const size_t mask = ((size_t)1 << t->slab_shift) - 1;
    return t->slabs[id >> t->slab_shift] + (id & mask) * t->row_bytes;
}

// recompute |x|^2 of rows [first, first+n) (fp32 tables on the MFMA path only)
static int update_norms(vsgpu_table *t, size_t first, size_t n) {
    if (!(t->mfma_ok || t->lowp_ok) || n == 0) return VSGPU_OK;
    const size_t slab_rows = (size_t)1 << t->slab_shift;
    size_t id = first, left = n;
    while (left) {
        size_t in_slab = std::min(left, slab_rows - (id & (slab_rows - 1)));
        float *np = t->norm_slabs[id >> t->slab_shift] + (id & (slab_rows - 1));
        const dim3 g((unsigned)((in_slab + 3) / 4));
        if (t->mfma_ok)
            hipLaunchKernelGGL(k_row_norms_f32, g, dim3(256), 0, t->ctx->stream, (const char *)row_ptr(t, id),
                               (uint32_t)t->row_bytes, (uint32_t)t->dim, (uint32_t)in_slab, np);
        else if (t->lp_kind == LP_I8 || t->lp_kind == LP_U8)
            hipLaunchKernelGGL(k_row_aux_i8, g, dim3(256), 0, t->ctx->stream, (const char *)row_ptr(t, id),
                               (uint32_t)t->row_bytes, (uint32_t)t->dim, (uint32_t)in_slab,
                               t->lp_kind == LP_U8 ? (t->metric == VSGPU_L2 ? 2 : 3) : (t->metric == VSGPU_COSINE ? 1 : 0),
                               (uint32_t *)np);
        else
            hipLaunchKernelGGL(k_row_norms_h16, g, dim3(256), 0, t->ctx->stream, (const char *)row_ptr(t, id),
                               (uint32_t)t->row_bytes, (uint32_t)t->dim, (uint32_t)in_slab, t->type, (uint32_t *)np);
        id += in_slab;
        left -= in_slab;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(t->ctx->stream));
    return VSGPU_OK;
}

extern "C" int vsgpu_table_append(vsgpu_table *t, const void *host_rows, size_t n) {
    if (n == 0) return VSGPU_OK;
    HIPCHK(hipSetDevice(t->ctx->device));
    if (t->n + n > 0xFFFFFFF0ull) return fail(VSGPU_ERR_UNSUPPORTED, "more than 2^32 rows per device table");
    int rc = grow_to(t, t->n + n);
    if (rc) return rc;
    const size_t slab_rows = (size_t)1 << t->slab_shift;
    const char *src = (const char *)host_rows;
    size_t id = t->n, left = n;
    while (left) {
        size_t in_slab = std::min(left, slab_rows - (id & (slab_rows - 1)));
        HIPCHK(hipMemcpyAsync(row_ptr(t, id), src, in_slab * t->row_bytes, hipMemcpyHostToDevice, t->ctx->stream));
        src += in_slab * t->row_bytes;
        id += in_slab;
        left -= in_slab;
    }
    HIPCHK(hipStreamSynchronize(t->ctx->stream));  // the caller's buffer is borrowed for this call only
    rc = update_norms(t, t->n, n);
    if (rc) return rc;
    t->n += n;
    return VSGPU_OK;
}
extern "C" int vsgpu_table_write(vsgpu_table *t, size_t id, const void *host_row) {
    if (id >= t->n) return fail(VSGPU_ERR_ARG, "row %zu out of range", id);
    HIPCHK(hipSetDevice(t->ctx->device));
    // same stream as the norm kernel and the queries (the ctx stream does not synchronise with the legacy stream)
    HIPCHK(hipMemcpyAsync(row_ptr(t, id), host_row, t->row_bytes, hipMemcpyHostToDevice, t->ctx->stream));
    HIPCHK(hipStreamSynchronize(t->ctx->stream));
    return update_norms(t, id, 1);
}
extern "C" int vsgpu_table_move(vsgpu_table *t, size_t dst, size_t src) {
    if (dst >= t->n || src >= t->n) return fail(VSGPU_ERR_ARG, "move %zu <- %zu out of range", dst, src);
    if (dst == src) return VSGPU_OK;
    HIPCHK(hipSetDevice(t->ctx->device));
    HIPCHK(hipMemcpyAsync(row_ptr(t, dst), row_ptr(t, src), t->row_bytes, hipMemcpyDeviceToDevice, t->ctx->stream));
    return update_norms(t, dst, 1);
}
extern "C" int vsgpu_table_truncate(vsgpu_table *t, size_t new_size) {
    if (new_size > t->n) return fail(VSGPU_ERR_ARG, "truncate beyond size");
    t->n = new_size;  // slabs are kept (the reference frees whole blocks; capacity is not on the query path)
    return VSGPU_OK;
}
extern "C" int vsgpu_table_read(vsgpu_table *t, size_t id, void *host_row) {
    if (id >= t->n) return fail(VSGPU_ERR_ARG, "row %zu out of range", id);
    HIPCHK(hipSetDevice(t->ctx->device));
    HIPCHK(hipMemcpyAsync(host_row, row_ptr(t, id), t->row_bytes, hipMemcpyDeviceToHost, t->ctx->stream));
    HIPCHK(hipStreamSynchronize(t->ctx->stream));
    return VSGPU_OK;
}
extern "C" int vsgpu_table_append_synthetic(vsgpu_table *t, size_t n, uint64_t seed) {
    if (t->type == VSGPU_F64 || t->type == VSGPU_U8) return fail(VSGPU_ERR_UNSUPPORTED, "synthetic fill: fp32/bf16/fp16/int8 only");
    if (n == 0) return VSGPU_OK;
    HIPCHK(hipSetDevice(t->ctx->device));
    if (t->n + n > 0xFFFFFFF0ull) return fail(VSGPU_ERR_UNSUPPORTED, "more than 2^32 rows per device table");
    int rc = grow_to(t, t->n + n);
    if (rc) return rc;
    const size_t slab_rows = (size_t)1 << t->slab_shift;
    size_t id = t->n, left = n;
    while (left) {
        size_t in_slab = std::min(left, slab_rows - (id & (slab_rows - 1)));
        uint64_t count = (uint64_t)in_slab * t->dim;
        int grid = (int)std::min<uint64_t>((count + 255) / 256, 8192);
        if (t->type == VSGPU_F32)
            hipLaunchKernelGGL(k_fill_uniform_f32, dim3(grid), dim3(256), 0, t->ctx->stream, (float *)row_ptr(t, id),
                               (uint64_t)id * t->dim, count, seed);
        else if (t->type == VSGPU_BF16 || t->type == VSGPU_F16)
            hipLaunchKernelGGL(k_fill_uniform_h16, dim3(grid), dim3(256), 0, t->ctx->stream, (uint16_t *)row_ptr(t, id),
                               (uint64_t)id * t->dim, count, seed, t->type == VSGPU_BF16 ? 1 : 0);
        else
            hipLaunchKernelGGL(k_fill_rows_i8, dim3((unsigned)((in_slab + 3) / 4)), dim3(256), 0, t->ctx->stream,
                               row_ptr(t, id), (uint32_t)t->row_bytes, (uint32_t)t->dim, (uint64_t)id, (uint32_t)in_slab, seed,
                               t->row_bytes > t->dim ? 1 : 0);
        id += in_slab;
        left -= in_slab;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(t->ctx->stream));
    rc = update_norms(t, t->n, n);
    if (rc) return rc;
    t->n += n;
    return VSGPU_OK;
}

// ------------------------------------------------------------------ query staging
// widen one stored element to the accumulator type (host side of the LDS query image only)
static inline float widen_f16(uint16_t h) {
    _Float16 v;
    memcpy(&v, &h, 2);
    return (float)v;
}
static inline float widen_bf16(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// Build [nq][steps][vl] permuted/widened query images in pinned memory and upload them.
static int stage_queries(vsgpu_table *t, const void *queries, size_t nq, size_t qstride) {
    vsgpu_ctx *c = t->ctx;
    const LaneProgram &pg = t->prog;
    const size_t per_q = (size_t)pg.steps * pg.vl;
    const size_t ab = acc_bytes(t->type);
    const size_t bytes = nq * per_q * ab;
    int rc = ensure_pinned(c, bytes + nq * 4);
    if (rc) return rc;
    rc = ensure(c, c->qperm, bytes);
    if (rc) return rc;
    char *dst = (char *)c->pinned;
    const int32_t *offs = pg.offs.data();
    bool identity = (t->type == VSGPU_F32 || t->type == VSGPU_F64);
    for (size_t i = 0; identity && i < per_q; i++) identity = offs[i] == (int32_t)(i * ab);
    for (size_t q = 0; q < nq; q++) {
        const char *src = (const char *)queries + q * qstride;
        char *o = dst + q * per_q * ab;
        // dims that fill every lane of every step (fp32/fp64, no residual head) make the table the identity
        if (identity) {
            memcpy(o, src, per_q * ab);
            continue;
        }
        switch (t->type) {
        case VSGPU_F32: {
            float *of = (float *)o;
            for (size_t i = 0; i < per_q; i++) {
                float v = 0;
                if (offs[i] >= 0) memcpy(&v, src + offs[i], 4);
                of[i] = v;
            }
            break;
        }
        case VSGPU_F64: {
            double *od = (double *)o;
            for (size_t i = 0; i < per_q; i++) {
                double v = 0;
                if (offs[i] >= 0) memcpy(&v, src + offs[i], 8);
                od[i] = v;
            }
            break;
        }
        case VSGPU_F16:
        case VSGPU_BF16: {
            float *of = (float *)o;
            const bool f16 = (t->type == VSGPU_F16);
            for (size_t i = 0; i < per_q; i++) {
                float v = 0;
                if (offs[i] >= 0) {
                    uint16_t h;
                    memcpy(&h, src + offs[i], 2);
                    v = f16 ? widen_f16(h) : widen_bf16(h);
                }
                of[i] = v;
            }
            break;
        }
        case VSGPU_I8: {
            int *oi = (int *)o;
            for (size_t i = 0; i < per_q; i++) oi[i] = offs[i] >= 0 ? (int)*(const int8_t *)(src + offs[i]) : 0;
            break;
        }
        default: {
            int *oi = (int *)o;
            for (size_t i = 0; i < per_q; i++) oi[i] = offs[i] >= 0 ? (int)*(const uint8_t *)(src + offs[i]) : 0;
            break;
        }
        }
    }
    HIPCHK(hipMemcpyAsync(c->qperm.p, c->pinned, bytes, hipMemcpyHostToDevice, c->stream));
    if (t->epi == EPI_INT_COS) {
        float *qn = (float *)((char *)c->pinned + bytes);
        for (size_t q = 0; q < nq; q++) memcpy(&qn[q], (const char *)queries + q * qstride + t->dim, 4);
        rc = ensure(c, c->qnorm, nq * 4);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(c->qnorm.p, qn, nq * 4, hipMemcpyHostToDevice, c->stream));
    }
    return VSGPU_OK;
}

// ------------------------------------------------------------------ kernel dispatch
template <int EK, int OPK, int BT> static void launch_scan_t(const ScanParams &P, dim3 grid, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL((k_exact_scan<EK, OPK, BT>), grid, dim3(256), lds, s, P);
}
template <int EK, int OPK> static void launch_scan_bt(int bt, const ScanParams &P, dim3 grid, size_t lds, hipStream_t s) {
    if (bt == 8) launch_scan_t<EK, OPK, 8>(P, grid, lds, s);
    else if (bt == 4) launch_scan_t<EK, OPK, 4>(P, grid, lds, s);
    else launch_scan_t<EK, OPK, 1>(P, grid, lds, s);
}
template <int EK> static void launch_scan_op(int opk, int bt, const ScanParams &P, dim3 grid, size_t lds, hipStream_t s) {
    switch (opk) {
    case OP_L2_FMA: launch_scan_bt<EK, OP_L2_FMA>(bt, P, grid, lds, s); break;
    case OP_IP_FMA: launch_scan_bt<EK, OP_IP_FMA>(bt, P, grid, lds, s); break;
    case OP_L2_MULADD: launch_scan_t<EK, OP_L2_MULADD, 1>(P, grid, lds, s); break;
    default: launch_scan_t<EK, OP_IP_MULADD, 1>(P, grid, lds, s); break;
    }
}
static void launch_scan(int ek, int opk, int bt, const ScanParams &P, dim3 grid, size_t lds, hipStream_t s) {
    switch (ek) {
    case EK_F32: launch_scan_op<EK_F32>(opk, bt, P, grid, lds, s); break;
    case EK_F64: launch_scan_op<EK_F64>(opk, bt, P, grid, lds, s); break;
    case EK_BF16: launch_scan_op<EK_BF16>(opk, bt, P, grid, lds, s); break;
    case EK_F16: launch_scan_op<EK_F16>(opk, bt, P, grid, lds, s); break;
    case EK_I8: launch_scan_op<EK_I8>(opk, bt, P, grid, lds, s); break;
    default: launch_scan_op<EK_U8>(opk, bt, P, grid, lds, s); break;
    }
}
static int tile_rows_of(int ek) { return (ek == EK_F64 || ek == EK_BF16) ? (256 / 16) * 4 : (256 / 32) * 4; }

static int pick_bt(const vsgpu_table *t, size_t nq) {
    int bt = t->bt_max;
    while (bt > 1 && (size_t)bt / 2 >= nq) bt /= 2;  // 8 -> 4 -> ... while the smaller tile still covers nq
    if (bt == 2) bt = 4;
    if (bt > 1 && nq == 1) bt = 1;
    return bt;
}

// Fill the table/program part of ScanParams and launch over compact rows.
static int run_scan(vsgpu_table *t, ScanParams &P, size_t nq, bool timed) {
    vsgpu_ctx *c = t->ctx;
    P.slabs = t->d_slabs;
    P.slab_shift = t->slab_shift;
    P.slab_mask = (uint32_t)(((size_t)1 << t->slab_shift) - 1);
    P.row_stride = (uint32_t)t->row_bytes;
    P.offs = t->d_offs;
    P.steps = t->prog.steps;
    P.full_from = t->prog.full_from();
    P.full_to = t->prog.full_to();
    P.qperm = c->qperm.p;
    P.nq = (int)nq;
    P.epilogue = t->epi;
    P.norm_off = (uint32_t)t->dim;
    P.qnorm = (const float *)c->qnorm.p;
    const int bt = pick_bt(t, nq);
    const int tile_rows = tile_rows_of(t->ek);
    const uint32_t n_tiles = (P.n_compact + tile_rows - 1) / tile_rows;
    if (n_tiles == 0) return VSGPU_OK;
    const size_t offs_b = ((size_t)t->prog.steps * t->prog.vl * 4 + 15) & ~(size_t)15;
    const size_t lds = offs_b + (size_t)bt * t->prog.steps * t->prog.vl * acc_bytes(t->type);
    const uint32_t q_tiles = (uint32_t)((nq + bt - 1) / bt);
    uint32_t gx = std::min<uint32_t>(n_tiles, (uint32_t)c->n_cu * 8);
    if (timed) HIPCHK(hipEventRecord(c->ev_a, c->stream));
    launch_scan(t->ek, t->opk, bt, P, dim3(gx, q_tiles), lds, c->stream);
    HIPCHK(hipGetLastError());
    if (timed) HIPCHK(hipEventRecord(c->ev_b, c->stream));
    return VSGPU_OK;
}

static void account_scan(vsgpu_ctx *c, vsgpu_table *t, uint64_t rows, uint64_t passes, const char *name) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->ev_a, c->ev_b) == hipSuccess) {
        c->stats.scan_ms += ms;
        c->stats.scan_launches += 1;
        c->stats.scan_rows += rows;
        c->stats.scan_bytes += rows * t->row_bytes;
        (void)passes;
        snprintf(c->stats.scan_kernel, sizeof c->stats.scan_kernel, "%s", name);
    }
}

// ------------------------------------------------------------------ dense scores
// scores of compact rows (contiguous range or id list) for nq staged queries -> host doubles [nq][n]
// big host-side loops (widening a few million scores) run on a handful of threads
template <typename F> static void host_parallel(size_t n, size_t grain, F f) {
    size_t workers = std::min<size_t>(8, std::max<size_t>(1, std::thread::hardware_concurrency()));
    workers = std::min(workers, std::max<size_t>(1, n / grain));
    if (workers <= 1) {
        f(0, n);
        return;
    }
    std::vector<std::thread> pool;
    const size_t per = (n + workers - 1) / workers;
    for (size_t w = 0; w < workers; w++)
        if (w * per < n) pool.emplace_back(f, w * per, std::min(n, (w + 1) * per));
    for (auto &th : pool) th.join();
}
static int dense_to_host(vsgpu_table *t, size_t nq, const uint32_t *d_ids, size_t first, size_t n,
                         double *out /*[nq][n]*/) {
    vsgpu_ctx *c = t->ctx;
    const bool f64 = (t->type == VSGPU_F64);
    const size_t sb = f64 ? 8 : 4;
    int rc = ensure(c, c->dense, nq * n * sb);
    if (rc) return rc;
    ScanParams P{};
    P.row_ids = d_ids;
    P.row_begin = (uint32_t)first;
    P.row_end = (uint32_t)t->n;
    P.n_compact = (uint32_t)n;
    P.tile_step = (uint32_t)tile_rows_of(t->ek);
    P.mode = MODE_DENSE;
    P.out = c->dense.p;
    P.out_stride = n;
    rc = run_scan(t, P, nq, false);
    if (rc) return rc;
    rc = ensure_pinned(c, nq * n * sb);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->pinned, c->dense.p, nq * n * sb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (f64) memcpy(out, c->pinned, nq * n * 8);
    else {
        const float *src = (const float *)c->pinned;
        host_parallel(nq * n, (size_t)1 << 19, [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) out[i] = (double)src[i];
        });
    }
    return VSGPU_OK;
}

extern "C" int vsgpu_scores(vsgpu_table *t, const void *query, size_t first, size_t n, double *scores) {
    if (first + n > t->n) return fail(VSGPU_ERR_ARG, "range [%zu,%zu) beyond table size %zu", first, first + n, t->n);
    if (n == 0) return VSGPU_OK;
    HIPCHK(hipSetDevice(t->ctx->device));
    int rc = stage_queries(t, query, 1, 0);
    if (rc) return rc;
    // chunk so the dense buffer stays modest
    const size_t chunk = (size_t)1 << 24;
    for (size_t off = 0; off < n; off += chunk) {
        size_t m = std::min(chunk, n - off);
        rc = dense_to_host(t, 1, nullptr, first + off, m, scores + off);
        if (rc) return rc;
    }
    return VSGPU_OK;
}

// ------------------------------------------------------------------ device-resident score vector (batch iterator)
struct vsgpu_scorebuf {
    vsgpu_table *t = nullptr;
    size_t n = 0;
    uint32_t *scores = nullptr;  // float bits, one per row of the table at creation time
    uint32_t *work = nullptr;    // [0..2047] histogram, [2048] compact count, then row ids for retire
    uint2 *out = nullptr;
    size_t out_cap = 0, work_ids_cap = 0;
};
extern "C" void vsgpu_scorebuf_destroy(vsgpu_scorebuf *b) {
    if (!b) return;
    (void)hipSetDevice(b->t->ctx->device);
    if (b->scores) (void)hipFree(b->scores);
    if (b->work) (void)hipFree(b->work);
    if (b->out) (void)hipFree(b->out);
    delete b;
}
extern "C" vsgpu_scorebuf *vsgpu_scorebuf_create(vsgpu_table *t, const void *query) {
    vsgpu_ctx *c = t->ctx;
    if (t->type == VSGPU_F64 || t->n == 0 || t->n > 0xFFFFFFF0ull) return nullptr;  // fp64 scores are doubles: host path
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    if (stage_queries(t, query, 1, 0)) return nullptr;
    WallMarks wm;
    auto *b = new vsgpu_scorebuf();
    b->t = t;
    b->n = t->n;
    b->work_ids_cap = 4096;
    if (hipMalloc((void **)&b->scores, b->n * 4) != hipSuccess || hipMalloc((void **)&b->work, (2064 + b->work_ids_cap) * 4) != hipSuccess) {
        vsgpu_scorebuf_destroy(b);
        return nullptr;
    }
    const size_t chunk = (size_t)1 << 26;
    for (size_t off = 0; off < b->n; off += chunk) {
        const size_t m = std::min(chunk, b->n - off);
        ScanParams P{};
        P.row_ids = nullptr;
        P.row_begin = (uint32_t)off;
        P.row_end = (uint32_t)t->n;
        P.n_compact = (uint32_t)m;
        P.tile_step = (uint32_t)tile_rows_of(t->ek);
        P.mode = MODE_DENSE;
        P.out = b->scores + off;
        P.out_stride = m;
        if (run_scan(t, P, 1, false)) {
            vsgpu_scorebuf_destroy(b);
            return nullptr;
        }
    }
    wm.mark("alloc+launch");
    if (hipStreamSynchronize(c->stream) != hipSuccess) {
        vsgpu_scorebuf_destroy(b);
        return nullptr;
    }
    wm.mark("dense_scan");
    wm.flush("scorebuf_create");
    return b;
}
extern "C" size_t vsgpu_scorebuf_rows(const vsgpu_scorebuf *b) { return b->n; }
extern "C" int vsgpu_scorebuf_next(vsgpu_scorebuf *b, size_t k, size_t cap, uint32_t *ids, double *scores, uint32_t *count) {
    vsgpu_ctx *c = b->t->ctx;
    HIPCHK(hipSetDevice(c->device));
    *count = 0;
    if (k == 0) return VSGPU_OK;
    if (cap > b->out_cap) {
        if (b->out) HIPCHK(hipFree(b->out));
        b->out = nullptr;
        HIPCHK(hipMalloc((void **)&b->out, cap * sizeof(uint2)));
        b->out_cap = cap;
    }
    const uint32_t n = (uint32_t)b->n;
    const dim3 grid((unsigned)std::min<size_t>((b->n + 255) / 256, (size_t)c->n_cu * 8));
    // three histogram passes: bits 31..21, 20..10, 9..0 of the order-preserving key
    const int shifts[3] = {21, 10, 0};
    const uint32_t bins[3] = {2048, 2048, 1024};
    uint32_t mask = 0, prefix = 0;
    uint64_t need = k;
    std::vector<uint32_t> h(2048);
    bool all = false;
    for (int p = 0; p < 3 && !all; p++) {
        HIPCHK(hipMemsetAsync(b->work, 0, 2049 * 4, c->stream));
        hipLaunchKernelGGL(k_iter_hist, grid, dim3(256), 0, c->stream, (const uint32_t *)b->scores, n, mask, prefix, shifts[p], bins[p],
                           b->work);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h.data(), b->work, bins[p] * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        uint64_t cum = 0;
        uint32_t bsel = bins[p];
        for (uint32_t i = 0; i < bins[p]; i++) {
            if (cum + h[i] >= need) {
                bsel = i;
                break;
            }
            cum += h[i];
        }
        if (bsel == bins[p]) {  // fewer than k live scores under this prefix: everything qualifies
            all = true;
            break;
        }
        need -= cum;
        prefix |= bsel << shifts[p];
        mask |= (bins[p] - 1) << shifts[p];
    }
    const uint32_t tkey = all ? 0xFFFFFFFFu : prefix;
    HIPCHK(hipMemsetAsync(b->work + 2048, 0, 4, c->stream));
    hipLaunchKernelGGL(k_iter_compact, grid, dim3(256), 0, c->stream, (const uint32_t *)b->scores, n, tkey, b->out, b->work + 2048,
                       (uint32_t)cap);
    HIPCHK(hipGetLastError());
    uint32_t got = 0;
    HIPCHK(hipMemcpyAsync(&got, b->work + 2048, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (got > cap) {
        *count = VSGPU_COUNT_OVERFLOW;
        return VSGPU_OK;
    }
    std::vector<uint2> rec(got);
    if (got) {
        HIPCHK(hipMemcpyAsync(rec.data(), b->out, got * sizeof(uint2), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    for (uint32_t i = 0; i < got; i++) {
        float f;
        memcpy(&f, &rec[i].y, 4);
        ids[i] = rec[i].x;
        scores[i] = (double)f;
    }
    *count = got;
    return VSGPU_OK;
}
extern "C" int vsgpu_scorebuf_retire(vsgpu_scorebuf *b, const uint32_t *rows, size_t m) {
    if (m == 0) return VSGPU_OK;
    vsgpu_ctx *c = b->t->ctx;
    HIPCHK(hipSetDevice(c->device));
    if (m > b->work_ids_cap) {
        uint32_t *w = nullptr;
        HIPCHK(hipMalloc((void **)&w, (2064 + m) * 4));
        HIPCHK(hipFree(b->work));
        b->work = w;
        b->work_ids_cap = m;
    }
    HIPCHK(hipMemcpyAsync(b->work + 2064, rows, m * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_iter_retire, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, b->scores, (const uint32_t *)(b->work + 2064),
                       (uint32_t)m);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return VSGPU_OK;
}
extern "C" int vsgpu_scorebuf_read(vsgpu_scorebuf *b, double *all) {
    vsgpu_ctx *c = b->t->ctx;
    HIPCHK(hipSetDevice(c->device));
    std::vector<float> tmp(b->n);
    HIPCHK(hipMemcpyAsync(tmp.data(), b->scores, b->n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    host_parallel(b->n, (size_t)1 << 19, [&](size_t a, size_t e) {
        for (size_t i = a; i < e; i++) all[i] = (double)tmp[i];
    });
    return VSGPU_OK;
}

extern "C" int vsgpu_scores_of(vsgpu_table *t, const void *query, const uint32_t *ids, size_t n, double *scores) {
    if (n == 0) return VSGPU_OK;
    for (size_t i = 0; i < n; i++)
        if (ids[i] >= t->n) return fail(VSGPU_ERR_ARG, "row id %u beyond table size %zu", ids[i], t->n);
    vsgpu_ctx *c = t->ctx;
    HIPCHK(hipSetDevice(c->device));
    int rc = stage_queries(t, query, 1, 0);
    if (rc) return rc;
    rc = ensure(c, c->ids, n * 4);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->ids.p, ids, n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));  // ids is caller memory: finish the copy before returning paths diverge
    return dense_to_host(t, 1, (const uint32_t *)c->ids.p, 0, n, scores);
}

// ------------------------------------------------------------------ selection helpers (host, on GPU scores)
struct Hit {
    uint32_t id;
    double score;
};
// keep rows with score <= T (T = k-th smallest), ascending id
static void select_upto_kth(std::vector<Hit> &hits, size_t k) {
    if (hits.size() > k) {
        std::vector<double> s(hits.size());
        for (size_t i = 0; i < hits.size(); i++) s[i] = hits[i].score;
        std::nth_element(s.begin(), s.begin() + (k - 1), s.end());
        const double T = s[k - 1];
        size_t w = 0;
        for (size_t i = 0; i < hits.size(); i++)
            if (hits[i].score <= T) hits[w++] = hits[i];
        hits.resize(w);
    }
    std::sort(hits.begin(), hits.end(), [](const Hit &a, const Hit &b) { return a.id < b.id; });
}
static void emit(const std::vector<Hit> &hits, size_t q, size_t cap, uint32_t *ids, double *scores, uint32_t *counts) {
    if (hits.size() > cap) {
        counts[q] = VSGPU_COUNT_OVERFLOW;
        return;
    }
    counts[q] = (uint32_t)hits.size();
    for (size_t i = 0; i < hits.size(); i++) {
        ids[q * cap + i] = hits[i].id;
        scores[q * cap + i] = hits[i].score;
    }
}

// D2H of the per-query candidate lists (exact scores already in place), host-side selection of
// {score <= T_k} in id order; queries whose list overflowed fall back to a dense exact pass.
static int topk_dense_path(vsgpu_table *t, size_t nq, size_t k, size_t cap, uint32_t *ids, double *scores,
                           uint32_t *counts, size_t q_first, size_t q_count, const void *queries, size_t qstride);
static int collect_candidates(vsgpu_table *t, const void *queries, size_t nq, size_t qstride, size_t k, size_t cap,
                              size_t ccap, uint32_t *ids, double *scores, uint32_t *counts, const char *scan_name) {
    vsgpu_ctx *c = t->ctx;
    const size_t n = t->n;
    // GPU: keep, per query, the candidates with exact score <= T_k; only those travel to the host
    const size_t ocap = cap;
    int rc = ensure(c, c->sel, nq * ocap * sizeof(uint2));
    if (rc) return rc;
    rc = ensure(c, c->selcnt, nq * 8);
    if (rc) return rc;
    hipLaunchKernelGGL(k_select_upto_kth, dim3((unsigned)nq), dim3(256), 0, c->stream, (const uint2 *)c->cand.p,
                       (const uint32_t *)c->counts.p, (uint32_t)ccap, (uint32_t)std::min(k, n), (uint2 *)c->sel.p,
                       (uint32_t *)c->selcnt.p, (uint32_t)ocap);
    HIPCHK(hipGetLastError());
    // raw candidate counts ride along (statistics + "fewer than k" sanity check)
    HIPCHK(hipMemcpyAsync((uint32_t *)c->selcnt.p + nq, c->counts.p, nq * 4, hipMemcpyDeviceToDevice, c->stream));
    rc = ensure_pinned(c, nq * 8 + nq * ocap * sizeof(uint2));
    if (rc) return rc;
    uint32_t *hsel = (uint32_t *)c->pinned;
    uint2 *hrec = (uint2 *)((char *)c->pinned + nq * 8);
    HIPCHK(hipMemcpyAsync(hsel, c->selcnt.p, nq * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(hrec, c->sel.p, nq * ocap * sizeof(uint2), hipMemcpyDeviceToHost, c->stream));
    WallMarks wm;
    HIPCHK(hipStreamSynchronize(c->stream));
    wm.mark("wait_gpu");
    {
        account_scan(c, t, n, 1, scan_name);
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev_c, c->ev_d) == hipSuccess) c->stats.other_ms += ms;
    }
    // copy out of the pinned staging area: the dense fallback below reuses (and may reallocate) it
    std::vector<uint32_t> hsel_v(hsel, hsel + 2 * nq);
    std::vector<uint2> hrec_v(hrec, hrec + nq * ocap);
    hsel = hsel_v.data();
    hrec = hrec_v.data();
    const uint32_t *hraw = hsel + nq;
    if (getenv("VSGPU_VERIFY")) {
        // Test aid: recompute every query densely and report any row at or below the k-th exact score that the
        // filter -> re-rank -> select pipeline did not deliver, together with where it was lost.
        std::vector<uint2> hc(nq * ccap);
        std::vector<uint32_t> hcnt(nq);
        std::vector<float> htau(nq);
        HIPCHK(hipMemcpy(hc.data(), c->cand.p, nq * ccap * sizeof(uint2), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hcnt.data(), c->counts.p, nq * 4, hipMemcpyDeviceToHost));
        if (c->tau.p) HIPCHK(hipMemcpy(htau.data(), c->tau.p, nq * 4, hipMemcpyDeviceToHost));
        std::vector<double> row(n), tmp;
        for (size_t q = 0; q < nq; q++) {
            if (hraw[q] > ccap || hsel[q] == VSGPU_COUNT_OVERFLOW) continue;
            if (vsgpu_scores(t, (const char *)queries + q * qstride, 0, n, row.data())) break;
            tmp = row;
            const size_t kk = std::min(k, n);
            std::nth_element(tmp.begin(), tmp.begin() + (kk - 1), tmp.end());
            const double T = tmp[kk - 1];
            for (size_t i = 0; i < n; i++) {
                if (!(row[i] <= T)) continue;
                bool in_sel = false;
                for (size_t j = 0; j < hsel[q]; j++) in_sel |= (hrec_v[q * ocap + j].x == (uint32_t)i);
                if (in_sel) continue;
                long at = -1;
                const uint32_t cn = std::min<uint32_t>(hcnt[q], (uint32_t)ccap);
                for (uint32_t j = 0; j < cn; j++)
                    if (hc[q * ccap + j].x == (uint32_t)i) at = j;
                float stored = 0;
                if (at >= 0) memcpy(&stored, &hc[q * ccap + at].y, 4);
                size_t dup = 0, oob = 0;
                for (uint32_t j = 0; j < cn; j++) {
                    oob += hc[q * ccap + j].x >= n;
                    for (uint32_t j2 = j + 1; j2 < cn && j2 < j + 2; j2++) dup += 0;
                }
                fprintf(stderr,
                        "VSGPU_VERIFY MISS %s q=%zu row=%zu exact=%.9g T_k=%.9g tau=%.9g cand_slot=%ld stored=%.9g count=%u "
                        "raw=%u sel=%u oob_rows=%zu n=%zu\n",
                        scan_name, q, i, row[i], T, (double)htau[q], at, (double)stored, hcnt[q], hraw[q], hsel[q], oob, n);
            }
        }
    }
    std::vector<Hit> hits;
    for (size_t q = 0; q < nq; q++) {
        if (hraw[q] > ccap || hraw[q] < std::min(k, n)) {
            // more candidates than slots (heavy ties / adversarial data): exact dense fallback
            c->stats.fallbacks++;
            rc = topk_dense_path(t, nq, k, cap, ids, scores, counts, q, 1, queries, qstride);
            if (rc) return rc;
            continue;
        }
        c->stats.candidates += hraw[q];
        if (hsel[q] == VSGPU_COUNT_OVERFLOW) {  // more than `cap` rows tie at or below T_k
            counts[q] = VSGPU_COUNT_OVERFLOW;
            continue;
        }
        hits.resize(hsel[q]);
        for (size_t i = 0; i < hsel[q]; i++) {
            uint2 r = hrec[q * ocap + i];
            float f;
            memcpy(&f, &r.y, 4);
            hits[i] = Hit{r.x, (double)f};
        }
        std::sort(hits.begin(), hits.end(), [](const Hit &a, const Hit &b) { return a.id < b.id; });
        emit(hits, q, cap, ids, scores, counts);
    }
    wm.mark("host_post");
    wm.flush("collect");
    return VSGPU_OK;
}

// Candidate slots per query: the threshold comes from a sample of `probe_rows` rows, so about
// k * n / probe_rows rows pass the filter (more with a loose bound); leave 3x headroom.
// Probe size.  Probing n/div rows costs (n/div) row reads at streaming speed and leaves ~k*div candidates per query,
// each re-scored from a randomly placed row at a fraction r ~ 0.15 of that speed: the sum is smallest at
// div = sqrt(r * n / (nq * k)).  Measured optima (tools/sweep_probe.py, tools/bench_dims.py): 48 at 10 M x 768, batch 64,
// k 10 (formula: 48); larger probes win on small tables.  Integer kinds have no re-rank and a flat optimum: fixed 48.
static uint32_t probe_divisor(const vsgpu_ctx *c, size_t n, size_t nq, size_t k, bool rerank) {
    if (c->opt_probe_div > 0) return (uint32_t)c->opt_probe_div;
    if (!rerank) return 48;
    const double d = std::sqrt(0.15 * (double)n / (double)(std::max<size_t>(nq, 1) * std::max<size_t>(k, 1)));
    return (uint32_t)std::min(64.0, std::max(8.0, d));
}

static size_t candidate_capacity(const vsgpu_ctx *c, size_t k, size_t n, size_t probe_rows) {
    double expect = (double)k * (double)n / (double)std::max<size_t>(probe_rows, 1);
    size_t want = (size_t)std::min(expect * 3.0 + 64.0, 1048576.0);
    return std::max<size_t>((size_t)c->opt_cand_cap, want);
}

// ------------------------------------------------------------------ MFMA filter path (fp32, wide batches)
static inline uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// Default shapes: the probe always uses 64-row tiles;
// the filter uses 16-row x 1-KiB stages with non-temporal DMA when dim % 256 == 0 and the tile is at
// least as long as the ring (dim >= 512), else 64-row x 256-B stages.  Measured on 10M x 768 (profiles/):
// 64-row default policy 5.5 ms, 64-row nt 5.16 ms, 16-row nt 4.99 ms.
template <int KS> static uint32_t launch_filter_ks(MfmaParams Q, size_t n, uint32_t wgs, unsigned q_tiles, hipStream_t s) {
    if constexpr (KS % 8 == 0 && KS >= 16) {
        Q.n_tiles = (uint32_t)((n + 15) / 16);
        hipLaunchKernelGGL((k_mfma_filter<KS, MF_FILTER, 3, 2, 1, 16>), dim3(std::min(Q.n_tiles, wgs), q_tiles), dim3(256),
                           mf_lds_bytes(3), s, Q);
    } else {
        Q.n_tiles = (uint32_t)((n + 63) / 64);
        hipLaunchKernelGGL((k_mfma_filter<KS, MF_FILTER, 3, 2, 1, 64>), dim3(std::min(Q.n_tiles, wgs), q_tiles), dim3(256),
                           mf_lds_bytes(3), s, Q);
    }
    return Q.n_tiles;
}
template <int KS> static void launch_probe_ks(const MfmaParams &P, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((k_mfma_filter<KS, MF_PROBE, 3, 0, 1, 64>), grid, dim3(256), mf_lds_bytes(3), s, P);
}
// tuning variants of the d=768 filter kernel (option "mfma_variant"): ring depth / cache policy /
// occupancy / tile shape.  Returns the tile height (rows) of the launched variant, 0 if none matched.
#define MF_VARIANT(NS_, AUX_, MINW_, RT_)                                                                     \
    {                                                                                                         \
        Q.n_tiles = (uint32_t)((n + RT_ - 1) / RT_);                                                          \
        dim3 grid(std::min(Q.n_tiles, wgs), q_tiles);                                                         \
        hipLaunchKernelGGL((k_mfma_filter<24, MF_FILTER, NS_, AUX_, MINW_, RT_>), grid, dim3(256), mf_lds_bytes(NS_), \
                           s, Q);                                                                             \
        return RT_;                                                                                           \
    }
static int launch_mfma_variant(int variant, MfmaParams Q, size_t n, uint32_t wgs, unsigned q_tiles, hipStream_t s) {
    switch (variant) {
    case 1: MF_VARIANT(3, 2, 1, 64)
    case 2: MF_VARIANT(4, 0, 1, 64)
    case 3: MF_VARIANT(4, 2, 1, 64)
    case 4: MF_VARIANT(3, 0, 3, 64)
    case 5: MF_VARIANT(3, 2, 3, 64)
    case 6: MF_VARIANT(3, 2, 1, 16)
    case 7: MF_VARIANT(4, 2, 1, 16)
    case 8: MF_VARIANT(3, 2, 3, 16)
    case 9: MF_VARIANT(3, 0, 1, 16)
#define MF_VARIANT_X(NS_, AUX_, MINW_, RT_, XOPT_)                                                                         \
    {                                                                                                                 \
        Q.n_tiles = (uint32_t)((n + RT_ - 1) / RT_);                                                                  \
        dim3 grid(std::min(Q.n_tiles, wgs), q_tiles);                                                                 \
        auto kern = k_mfma_filter<24, MF_FILTER, NS_, AUX_, MINW_, RT_, XOPT_>;                                     \
        if (mf_lds_bytes(NS_) > 64 * 1024)                                                                            \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      mf_lds_bytes(NS_));                                                              \
        hipLaunchKernelGGL(kern, grid, dim3(256), mf_lds_bytes(NS_), s, Q);                                           \
        return RT_;                                                                                                   \
    }
    case 10: MF_VARIANT_X(3, 2, 1, 16, 1)   // scalar slab loads: no per-tile ring drain
    case 11: MF_VARIANT_X(4, 2, 1, 16, 1)
    case 12: MF_VARIANT_X(3, 2, 3, 16, 1)
    case 13: MF_VARIANT_X(3, 0, 1, 16, 1)
    case 14: MF_VARIANT_X(3, 2, 1, 16, 2)   // one norm copy per workgroup
    case 15: MF_VARIANT_X(3, 2, 1, 16, 4)   // survivor pre-screen
    case 16: MF_VARIANT_X(3, 2, 1, 16, 6)
    default: return 0;
    }
}
static void launch_filter(int ksteps, const MfmaParams &P, size_t n, uint32_t wgs, unsigned q_tiles, hipStream_t s) {
    switch (ksteps) {
    case 4: launch_filter_ks<4>(P, n, wgs, q_tiles, s); break;
    case 6: launch_filter_ks<6>(P, n, wgs, q_tiles, s); break;
    case 8: launch_filter_ks<8>(P, n, wgs, q_tiles, s); break;
    case 10: launch_filter_ks<10>(P, n, wgs, q_tiles, s); break;
    case 12: launch_filter_ks<12>(P, n, wgs, q_tiles, s); break;
    case 16: launch_filter_ks<16>(P, n, wgs, q_tiles, s); break;
    case 20: launch_filter_ks<20>(P, n, wgs, q_tiles, s); break;
    case 24: launch_filter_ks<24>(P, n, wgs, q_tiles, s); break;
    case 28: launch_filter_ks<28>(P, n, wgs, q_tiles, s); break;
    case 30: launch_filter_ks<30>(P, n, wgs, q_tiles, s); break;
    case 40: launch_filter_ks<40>(P, n, wgs, q_tiles, s); break;
    case 48: launch_filter_ks<48>(P, n, wgs, q_tiles, s); break;
    case 80: launch_filter_ks<80>(P, n, wgs, q_tiles, s); break;
    case 64: launch_filter_ks<64>(P, n, wgs, q_tiles, s); break;
    case 96: launch_filter_ks<96>(P, n, wgs, q_tiles, s); break;
    default: launch_filter_ks<32>(P, n, wgs, q_tiles, s); break;
    }
}
static void launch_probe(int ksteps, const MfmaParams &P, dim3 grid, hipStream_t s) {
    switch (ksteps) {
    case 4: launch_probe_ks<4>(P, grid, s); break;
    case 6: launch_probe_ks<6>(P, grid, s); break;
    case 8: launch_probe_ks<8>(P, grid, s); break;
    case 10: launch_probe_ks<10>(P, grid, s); break;
    case 12: launch_probe_ks<12>(P, grid, s); break;
    case 16: launch_probe_ks<16>(P, grid, s); break;
    case 20: launch_probe_ks<20>(P, grid, s); break;
    case 24: launch_probe_ks<24>(P, grid, s); break;
    case 28: launch_probe_ks<28>(P, grid, s); break;
    case 30: launch_probe_ks<30>(P, grid, s); break;
    case 40: launch_probe_ks<40>(P, grid, s); break;
    case 48: launch_probe_ks<48>(P, grid, s); break;
    case 80: launch_probe_ks<80>(P, grid, s); break;
    case 64: launch_probe_ks<64>(P, grid, s); break;
    case 96: launch_probe_ks<96>(P, grid, s); break;
    default: launch_probe_ks<32>(P, grid, s); break;
    }
}

static int topk_mfma(vsgpu_table *t, const void *queries, size_t nq, size_t qstride, size_t k, size_t cap,
                     uint32_t *ids, double *scores, uint32_t *counts) {
    vsgpu_ctx *c = t->ctx;
    const size_t n = t->n, dim = t->dim;
    const int KS = t->ksteps;
    const size_t q_tiles = (nq + MF_QTILE - 1) / MF_QTILE, nqp = q_tiles * MF_QTILE;
    const bool l2 = (t->metric == VSGPU_L2);

    // (1) exact-order query images for the re-rank, (2) bf16 B-operand fragments + |q|^2 for the filter
    int rc = stage_queries(t, queries, nq, qstride);
    if (rc) return rc;
    const size_t kdim = (size_t)KS * 32;       // kernel width >= dim
    std::vector<uint16_t> frag(nqp * kdim, 0);  // [q_tile][wave][kstep][lane][8]
    std::vector<float> qn2(nqp, 0.f), tau0(nqp, -INFINITY);
    for (size_t q = 0; q < nq; q++) {
        const float *src = (const float *)((const char *)queries + q * qstride);
        double ss = 0;
        for (size_t i = 0; i < dim; i++) ss += (double)src[i] * (double)src[i];
        qn2[q] = (float)ss;
        const size_t qt = q / MF_QTILE, w = (q % MF_QTILE) / 16, nn = q % 16;
        for (int s = 0; s < KS; s++)
            for (int kq = 0; kq < 4; kq++) {
                const size_t lane = (size_t)kq * 16 + nn;
                uint16_t *dst = &frag[((((qt * 4 + w) * KS + s) * 64) + lane) * 8];
                for (int j = 0; j < 8; j++) {
                    const size_t e = (size_t)32 * s + 8 * kq + j;
                    dst[j] = e < dim ? bf16_rne(src[e]) : (uint16_t)0;
                }
            }
    }
    rc = ensure(c, c->qfrag, frag.size() * 2);
    if (rc) return rc;
    rc = ensure(c, c->qn2, nqp * 4);
    if (rc) return rc;
    rc = ensure(c, c->tau, nqp * 4);
    if (rc) return rc;
    rc = ensure(c, c->counts, nqp * 4);
    if (rc) return rc;
    const uint32_t total_tiles = (uint32_t)((n + MF_TILE_ROWS - 1) / MF_TILE_ROWS);
    uint32_t probe_tiles = std::max<uint32_t>(total_tiles / probe_divisor(c, n, nq, k, true), (uint32_t)(4 * k));
    probe_tiles = std::min<uint32_t>(std::min<uint32_t>(probe_tiles, total_tiles), (uint32_t)c->opt_probe_cap);
    const size_t ccap = candidate_capacity(c, k, n, (size_t)probe_tiles * MF_TILE_ROWS);
    rc = ensure(c, c->cand, nqp * ccap * sizeof(uint2));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->qfrag.p, frag.data(), frag.size() * 2, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->qn2.p, qn2.data(), nqp * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->tau.p, tau0.data(), nqp * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->counts.p, 0, nqp * 4, c->stream));

    // rigorous |a - s_ref| <= cE*(|x|^2+|q|^2) + absE   (derivation: DESIGN.md §5.2)
    const double u = std::ldexp(1.0, -24);
    const double cq = std::ldexp(1.0, -8) * (1.0 + std::ldexp(1.0, -10)) + (double)kdim * std::ldexp(1.0, -22) * 1.01;
    const double gref = ((double)kdim / 32.0 + 12.0) * u;
    const float cE = (float)(((cq + 2.0 * gref) * 1.001 + 16.0 * u) * (1.0 + 1e-6));
    const float absE = l2 ? 1e-30f : 1e-6f;

    const uint32_t tile_step = total_tiles / probe_tiles;
    uint32_t M = 64;  // group minima sorted per query (more probe tiles than that are grouped, see topk_lowp)
    while (M < probe_tiles && M < 8192) M <<= 1;

    rc = ensure(c, c->dense, nqp * (size_t)probe_tiles * 4);
    if (rc) return rc;

    MfmaParams P{};
    P.slabs = t->d_slabs;
    P.norm_slabs = t->d_norm_slabs;
    P.slab_shift = t->slab_shift;
    P.slab_mask = (uint32_t)(((size_t)1 << t->slab_shift) - 1);
    P.row_stride = (uint32_t)t->row_bytes;
    P.n_rows = (uint32_t)n;
    P.qfrag = (const uint4 *)c->qfrag.p;
    P.qn2 = (const float *)c->qn2.p;
    P.cE = cE;
    P.absE = absE;
    P.is_l2 = l2 ? 1 : 0;
    P.tau = (const float *)c->tau.p;
    P.counts = (uint32_t *)c->counts.p;
    P.cand = (uint2 *)c->cand.p;
    P.cap = (uint32_t)ccap;
    const uint32_t wg_cap = (uint32_t)c->n_cu * 2;

    HIPCHK(hipEventRecord(c->ev_c, c->stream));
    {   // probe: strided tiles -> per (tile, query) upper bounds
        MfmaParams Q = P;
        Q.tile_first = 0;
        Q.tile_step = tile_step;
        Q.n_tiles = probe_tiles;
        Q.tilemin = (float *)c->dense.p;
        Q.tilemin_stride = probe_tiles;
        launch_probe(KS, Q, dim3(std::min(probe_tiles, wg_cap), (unsigned)q_tiles), c->stream);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_probe_threshold, dim3((unsigned)nq), dim3(1024), M * sizeof(float), c->stream,
                           (const float *)c->dense.p, (size_t)probe_tiles, probe_tiles, (uint32_t)k, M, (float *)c->tau.p);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(c->ev_d, c->stream));
    HIPCHK(hipEventRecord(c->ev_a, c->stream));
    {   // filter: every tile once
        MfmaParams Q = P;
        Q.tile_first = 0;
        Q.tile_step = 1;
        Q.n_tiles = total_tiles;
        const uint32_t wgs = (uint32_t)c->n_cu * (uint32_t)c->opt_wg_per_cu;
        if (!(KS == 24 && launch_mfma_variant((int)c->opt_mfma_variant, Q, n, wgs, (unsigned)q_tiles, c->stream)))
            launch_filter(KS, Q, n, wgs, (unsigned)q_tiles, c->stream);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(c->ev_b, c->stream));
    {   // exact re-rank of the survivors, in place
        ScanParams S{};
        S.slabs = t->d_slabs;
        S.slab_shift = t->slab_shift;
        S.slab_mask = P.slab_mask;
        S.row_stride = P.row_stride;
        S.offs = t->d_offs;
        S.steps = t->prog.steps;
        S.qperm = c->qperm.p;
        S.nq = (int)nq;
        S.epilogue = t->epi;
        S.counts = (uint32_t *)c->counts.p;
        S.cand = (uint2 *)c->cand.p;
        S.cap = (uint32_t)ccap;
        dim3 grid(64, (unsigned)nq);
        if (t->opk == OP_L2_FMA) hipLaunchKernelGGL((k_exact_pairs<EK_F32, OP_L2_FMA>), grid, dim3(256), 0, c->stream, S);
        else hipLaunchKernelGGL((k_exact_pairs<EK_F32, OP_IP_FMA>), grid, dim3(256), 0, c->stream, S);
        HIPCHK(hipGetLastError());
    }
    return collect_candidates(t, queries, nq, qstride, k, cap, ccap, ids, scores, counts, "k_mfma_filter");
}

// ------------------------------------------------------------------ low-precision MFMA filter path
// One launcher for every instantiation: ring depths above 3 slots need more than the default 64 KiB of
// dynamic LDS, which HIP only grants after the attribute is raised.
template <int LK, int KS, int MODE, int RT, int NW, int NQW, int MINW, int NS, int STAGE = MF_STAGE_BYTES, int DIST = 0, int DLATE = 0, int ISS = 0>
static void launch_lowp_k(const LowpParams &P, dim3 grid, hipStream_t s) {
    constexpr int lds_bytes = lowp_lds_bytes(NW, KS, RT, NS, STAGE);
    static_assert(lds_bytes <= 160 * 1024, "LDS ring does not fit");
    auto go = [&](auto kern) {
        if (lds_bytes > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds_bytes, s, P);
    };
    // the diagnosis build (run-time dbg switches, paired query tiles) exists for the plain 3-slot filter kernels only
    constexpr bool has_diag = MODE == MF_FILTER && NS == 3 && DIST == 0 && DLATE == 0 && KS >= 16 && ISS == 0;
    if constexpr (has_diag) {
        if (P.dbg || P.pair_map) return go(k_mfma_filter_lowp<LK, KS, MODE, RT, NW, NQW, MINW, NS, STAGE, false, DIST, DLATE, true>);
    }
    // (no diagnosis build of this variant: the switches are compiled out, the production kernel runs)
    go(k_mfma_filter_lowp<LK, KS, MODE, RT, NW, NQW, MINW, NS, STAGE, false, DIST, DLATE, false, ISS>);
}
// barrier-free variant (mfma_free_kernels.hpp): NS slots, D units requested ahead, landed signalled L units early
template <int LK, int KS, int RT, int NW, int NQW, int NS, int STAGE, int D, int L>
static void launch_lowp_free(const LowpParams &P, dim3 grid, hipStream_t s) {
    constexpr int lds_bytes = free_lds_bytes(NW, KS, RT, NS, STAGE, D);
    static_assert(lds_bytes <= 160 * 1024, "LDS ring does not fit");
    auto kern = k_mfma_filter_free<LK, KS, RT, NW, NQW, NS, STAGE, D, L>;
    if (lds_bytes > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds_bytes, s, P);
}
template <int LK, int KS, int RT, int NW, int NQW, int NS, int STAGE = MF_STAGE_BYTES>
static void launch_lowp_skew(const LowpParams &P, dim3 grid, hipStream_t s) {
    constexpr int lds_bytes = lowp_lds_bytes(NW, KS, RT, NS, STAGE, true);
    static_assert(lds_bytes <= 160 * 1024, "LDS ring does not fit");
    auto kern = k_mfma_filter_lowp<LK, KS, MF_FILTER, RT, NW, NQW, 1, NS, STAGE, true>;
    if (lds_bytes > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds_bytes, s, P);
}
template <int LK, int KS, int RT, int NQW>
static void launch_lowp_t(int mode, const LowpParams &P, dim3 grid, hipStream_t s) {
    if (mode == MF_PROBE) launch_lowp_k<LK, KS, MF_PROBE, RT, 8, NQW, 1, 3>(P, grid, s);
    else launch_lowp_k<LK, KS, MF_FILTER, RT, 8, NQW, 1, 3>(P, grid, s);
}
template <int KS, int RT, int LK = LP_I8> static void launch_lowp_i8(int mode, const LowpParams &P, dim3 grid, hipStream_t s) {
    if (mode == MF_PROBE) launch_lowp_k<LK, KS, MF_PROBE, RT, 16, 1, 1, 3>(P, grid, s);
    else launch_lowp_k<LK, KS, MF_FILTER, RT, 16, 1, 1, 3>(P, grid, s);
}
template <int LK> static void launch_lowp_h16(int ks, int mode, const LowpParams &P, dim3 grid, hipStream_t s) {
    switch (ks) {
    case 8: launch_lowp_t<LK, 8, 64, 1>(mode, P, grid, s); break;
    case 16: launch_lowp_t<LK, 16, 32, 1>(mode, P, grid, s); break;
    case 24: launch_lowp_t<LK, 24, 32, 1>(mode, P, grid, s); break;
    case 48: launch_lowp_t<LK, 48, 16, 1>(mode, P, grid, s); break;  // d = 1536: 192 VGPRs of query fragments per wave
    default: launch_lowp_t<LK, 32, 16, 1>(mode, P, grid, s); break;
    }
}
// tuning variants (option "lowp_variant") for the two BASELINE shapes: bf16 d=768 and int8 d=1024.  A variant
// picks its own tile height, so it sizes the tile count and the grid itself.
static bool launch_lowp_variant(const vsgpu_table *t, int variant, LowpParams P, uint32_t max_wgs, unsigned q_tiles,
                                hipStream_t s) {
    if (variant == 0) return false;
    auto go = [&](int rt, auto launcher) {
        P.tile_first = 0;
        P.tile_step = 1;
        P.n_tiles = (uint32_t)((t->n + rt - 1) / rt);
        launcher(P, dim3(std::min(P.n_tiles, max_wgs), q_tiles), s);
        return true;
    };
    if (t->lp_kind == LP_BF16 && t->lp_ksteps == 24) {
        switch (variant) {
        case 1: return go(32, launch_lowp_k<LP_BF16, 24, MF_FILTER, 32, 8, 1, 1, 4>);
        case 2: return go(32, launch_lowp_k<LP_BF16, 24, MF_FILTER, 32, 8, 1, 1, 6>);
        case 3: return go(32, launch_lowp_k<LP_BF16, 24, MF_FILTER, 32, 8, 1, 1, 8>);
        case 4: return go(32, launch_lowp_k<LP_BF16, 24, MF_FILTER, 32, 8, 1, 2, 4>);
        case 5: return go(32, launch_lowp_k<LP_BF16, 24, MF_FILTER, 32, 8, 1, 1, 4, 24576>);   // 32 rows x 768 B
        case 6: return go(32, launch_lowp_k<LP_BF16, 24, MF_FILTER, 32, 8, 1, 1, 3, 49152>);   // 32 whole rows
        case 7: return go(16, launch_lowp_k<LP_BF16, 24, MF_FILTER, 16, 8, 1, 1, 5, 24576>);   // 16 whole rows
        case 8: return go(32, launch_lowp_k<LP_BF16, 24, MF_FILTER, 32, 8, 1, 1, 4, 16384, 2>);  // 4 slots, 2 ahead: plain barrier
        case 9: return go(32, launch_lowp_k<LP_BF16, 24, MF_FILTER, 32, 8, 1, 1, 5, 16384, 3>);
        case 10: return go(32, launch_lowp_k<LP_BF16, 24, MF_FILTER, 32, 8, 1, 1, 3, 16384, 0, 2>);  // refill requested after 2 / 4 / 8 fragments
        case 11: return go(32, launch_lowp_k<LP_BF16, 24, MF_FILTER, 32, 8, 1, 1, 3, 16384, 0, 4>);
        case 12: return go(32, launch_lowp_k<LP_BF16, 24, MF_FILTER, 32, 8, 1, 1, 3, 16384, 0, 8>);
        case 60: return go(32, launch_lowp_k<LP_BF16, 24, MF_FILTER, 32, 8, 1, 1, 3, 16384, 0, 0, 4>);   // 4 of the 8 waves request rows
        case 50: return go(32, launch_lowp_k<LP_BF16, 24, MF_FILTER, 32, 8, 1, 1, 3, 16384, 0, -1>);   // staggered refill
        case 51: return go(32, launch_lowp_k<LP_BF16, 24, MF_FILTER, 32, 8, 1, 1, 4, 16384, 0, -1>);
        case 40: return go(32, launch_lowp_free<LP_BF16, 24, 32, 8, 1, 8, 16384, 5, 2>);   // barrier-free ring
        case 41: return go(32, launch_lowp_free<LP_BF16, 24, 32, 8, 1, 8, 16384, 4, 2>);
        case 42: return go(32, launch_lowp_free<LP_BF16, 24, 32, 8, 1, 8, 16384, 6, 3>);
        case 43: return go(32, launch_lowp_free<LP_BF16, 24, 32, 8, 1, 6, 16384, 4, 2>);
        case 20: return go(32, launch_lowp_skew<LP_BF16, 24, 32, 8, 1, 3>);                      // phase-skewed halves
        case 21: return go(32, launch_lowp_skew<LP_BF16, 24, 32, 8, 1, 4>);
        case 22: return go(32, launch_lowp_skew<LP_BF16, 24, 32, 8, 1, 6>);
        }
    }
    if (t->lp_kind == LP_I8 && t->lp_ksteps == 16) {
        switch (variant) {
        case 1: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 4>);
        case 2: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 6>);
        case 3: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 8>);
        case 4: return go(16, launch_lowp_k<LP_I8, 16, MF_FILTER, 16, 16, 1, 1, 4>);           // 16 whole rows
        case 5: return go(16, launch_lowp_k<LP_I8, 16, MF_FILTER, 16, 16, 1, 1, 6>);
        case 6: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 4, 32768>);    // 32 whole rows
        case 7: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 3, 32768>);
        case 8: return go(64, launch_lowp_k<LP_I8, 16, MF_FILTER, 64, 16, 1, 1, 4, 32768>);    // 64 rows x 512 B
        case 9: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 8, 2, 1, 3>);            // 8 waves x 32 queries
        case 10: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 8, 2, 1, 4>);
        case 11: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 4, 4, 1, 3>);           // 4 waves x 64 queries
        case 12: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 4, 4, 1, 4>);
        case 17: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 4, 4, 1, 3, 32768>);    // 4 waves x 64 queries, whole rows
        case 18: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 4, 4, 1, 4, 32768>);
        case 19: return go(16, launch_lowp_k<LP_I8, 16, MF_FILTER, 16, 4, 4, 1, 4>);           // 16 whole rows per unit
        case 60: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 3, 32768, 0, 0, 8>);   // 8 of the 16 waves request rows
        case 61: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 3, 32768, 0, 0, 4>);   // one per SIMD
        case 62: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 4, 32768, 0, 0, 8>);
        case 50: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 3, 32768, 0, -1>);   // staggered refill
        case 51: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 4, 32768, 0, -1>);
        case 52: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 4, 16384, 0, -1>);
        case 40: return go(32, launch_lowp_free<LP_I8, 16, 32, 16, 1, 8, 16384, 5, 2>);   // barrier-free ring
        case 41: return go(32, launch_lowp_free<LP_I8, 16, 32, 16, 1, 8, 16384, 4, 2>);
        case 42: return go(32, launch_lowp_free<LP_I8, 16, 32, 16, 1, 8, 16384, 6, 3>);
        case 43: return go(32, launch_lowp_free<LP_I8, 16, 32, 16, 1, 4, 32768, 2, 1>);
        case 44: return go(32, launch_lowp_free<LP_I8, 16, 32, 16, 1, 6, 16384, 4, 2>);
        case 30: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 3, 32768, 0, 4>);   // refill requested after 4 / 8 / 16 / all fragments
        case 35: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 3, 32768, 0, 2>);
        case 36: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 4, 32768, 0, 4>);
        case 31: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 3, 32768, 0, 8>);
        case 32: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 3, 32768, 0, 16>);
        case 33: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 4, 32768, 0, 8>);
        case 34: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 4, 32768, 0, 32>);
        case 15: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 4, 32768, 2>);  // 4 slots, 2 ahead: plain barrier
        case 16: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 4, 16384, 2>);
        case 13: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 8, 2, 1, 3, 32768>);    // 8 waves x 32 queries, whole rows
        case 14: return go(32, launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 8, 2, 1, 4, 32768>);
        case 20: return go(32, launch_lowp_skew<LP_I8, 16, 32, 16, 1, 3, 32768>);                // phase-skewed halves
        case 21: return go(32, launch_lowp_skew<LP_I8, 16, 32, 16, 1, 4>);
        case 22: return go(32, launch_lowp_skew<LP_I8, 16, 32, 16, 1, 3>);
        case 24: return go(32, launch_lowp_skew<LP_I8, 16, 32, 16, 1, 6>);
        }
    }
    return false;
}
static void launch_lowp(const vsgpu_table *t, int mode, const LowpParams &P, dim3 grid, hipStream_t s) {
    if (t->lp_kind == LP_BF16) launch_lowp_h16<LP_BF16>(t->lp_ksteps, mode, P, grid, s);
    else if (t->lp_kind == LP_F16) launch_lowp_h16<LP_F16>(t->lp_ksteps, mode, P, grid, s);
    else if (t->lp_kind == LP_U8) {
        switch (t->lp_ksteps) {
        case 8: launch_lowp_i8<8, 64, LP_U8>(mode, P, grid, s); break;
        case 12: launch_lowp_i8<12, 64, LP_U8>(mode, P, grid, s); break;
        default:
            if (mode == MF_FILTER) launch_lowp_k<LP_U8, 16, MF_FILTER, 32, 16, 1, 1, 3, 32768>(P, grid, s);
            else launch_lowp_i8<16, 32, LP_U8>(mode, P, grid, s);
            break;
        }
    } else {
        // 16 waves x 16 queries.  d=1024 filter: a ring slot holds 32 whole rows (1 KiB per DMA instruction, one
        // barrier per 32 KiB): 3.54 TB/s against 3.24 for 16 KiB half-row slots, 3.1 for 8 waves x 32 queries and
        // 2.5 for 4 waves x 64 queries (profiles/r01_tuning_lowp.txt)
        switch (t->lp_ksteps) {
        case 8: launch_lowp_i8<8, 64>(mode, P, grid, s); break;
        case 12: launch_lowp_i8<12, 64>(mode, P, grid, s); break;
        default:
            if (mode == MF_FILTER) launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 3, 32768>(P, grid, s);
            else launch_lowp_i8<16, 32>(mode, P, grid, s);
            break;
        }
    }
}

// int8 with the query batch split over two 8-wave workgroups (blockIdx.y): both stream the same row tiles, the
// second reader is expected to hit L2 (same XCD when gridDim.x % 8 == 0)
static void launch_lowp_i8_split(const vsgpu_table *t, int mode, const LowpParams &P, dim3 grid, hipStream_t s) {
    if (t->lp_ksteps == 16 && t->lp_rt == 32) {
        if (mode == MF_PROBE) launch_lowp_k<LP_I8, 16, MF_PROBE, 32, 8, 1, 2, 3>(P, grid, s);
        else if (grid.y == 2 && grid.x % 8 == 0) {
            // both query tiles resident together: 2 x 8 waves per CU need <= 128 VGPRs (4 waves per SIMD)
            LowpParams Q = P;
            Q.pair_map = 1;
            launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 8, 1, 4, 3>(Q, dim3(grid.x * 2), s);
        } else launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 8, 1, 2, 3>(P, grid, s);
    } else if (t->lp_ksteps == 12) {
        if (mode == MF_PROBE) launch_lowp_k<LP_I8, 12, MF_PROBE, 64, 8, 1, 2, 3>(P, grid, s);
        else launch_lowp_k<LP_I8, 12, MF_FILTER, 64, 8, 1, 2, 3>(P, grid, s);
    } else {
        if (mode == MF_PROBE) launch_lowp_k<LP_I8, 8, MF_PROBE, 64, 8, 1, 2, 3>(P, grid, s);
        else launch_lowp_k<LP_I8, 8, MF_FILTER, 64, 8, 1, 2, 3>(P, grid, s);
    }
}

static int topk_lowp(vsgpu_table *t, const void *queries, size_t nq, size_t qstride, size_t k, size_t cap,
                     uint32_t *ids, double *scores, uint32_t *counts) {
    vsgpu_ctx *c = t->ctx;
    const size_t n = t->n, dim = t->dim;
    const int KS = t->lp_ksteps, RT = t->lp_rt;
    const bool qsplit = t->lp_kind == LP_I8 && c->opt_lowp_qsplit;
    const size_t QT = qsplit ? 128 : (size_t)t->lp_qtile, NQW = QT / 128;
    const size_t q_tiles = (nq + QT - 1) / QT, nqp = q_tiles * QT;
    const bool is_int = (t->lp_kind == LP_I8 || t->lp_kind == LP_U8);
    const bool is_u8 = (t->lp_kind == LP_U8);
    const size_t eb = is_int ? 1 : 2;
    const size_t kelem = is_int ? 64 : 32;        // elements per MFMA k-step
    const size_t per_lane = kelem / 4;            // elements per lane per k-step (16 bytes)

    int rc = VSGPU_OK;
    if (!is_int) {
        rc = stage_queries(t, queries, nq, qstride);   // exact-order images for the re-rank
        if (rc) return rc;
    }
    // fragments: [q_tile][wave 8][NQW][KSTEPS][lane 64][16 B]
    const size_t kdim = (size_t)KS * kelem;  // kernel width >= dim
    std::vector<unsigned char> frag(nqp * kdim * eb, 0);
    std::vector<uint32_t> qaux(nqp, 0);
    std::vector<float> tau0(nqp, -INFINITY);
    for (size_t q = 0; q < nq; q++) {
        const unsigned char *src = (const unsigned char *)queries + q * qstride;
        const size_t qt = q / QT, w = (q % QT) / (16 * NQW), nt = ((q % QT) % (16 * NQW)) / 16, nn = q % 16;
        for (int s = 0; s < KS; s++)
            for (int kq = 0; kq < 4; kq++) {
                const size_t lane = (size_t)kq * 16 + nn;
                unsigned char *dst = &frag[(((((qt * 8 + w) * NQW + nt) * KS + s) * 64) + lane) * 16];
                const size_t e0 = (kelem * s + per_lane * kq) * eb, have = e0 < dim * eb ? std::min<size_t>(16, dim * eb - e0) : 0;
                if (have) memcpy(dst, src + e0, have);
                if (is_u8)
                    for (size_t b = 0; b < have; b++) dst[b] ^= 0x80;  // q - 128 as int8 (columns past dim stay 0)
            }
        if (is_u8) {
            int s1 = 0, s2 = 0;
            for (size_t i = 0; i < dim; i++) {
                const int v = (int)src[i] - 128;
                s1 += v;
                s2 += v * v;
            }
            const int aux = t->epi == EPI_INT_L2 ? s2 : 128 * s1 + 16384 * (int)dim;
            memcpy(&qaux[q], &aux, 4);
        } else if (is_int) {
            if (t->epi == EPI_INT_COS) memcpy(&qaux[q], src + dim, 4);
            else if (t->epi == EPI_INT_L2) {
                int ss = 0;
                for (size_t i = 0; i < dim; i++) ss += (int)(int8_t)src[i] * (int)(int8_t)src[i];
                memcpy(&qaux[q], &ss, 4);
            }
        } else {
            double ss = 0;
            for (size_t i = 0; i < dim; i++) {
                uint16_t h;
                memcpy(&h, src + 2 * i, 2);
                double v = t->type == VSGPU_BF16 ? (double)widen_bf16(h) : (double)widen_f16(h);
                ss += v * v;
            }
            float f = (float)ss;
            memcpy(&qaux[q], &f, 4);
        }
    }
    rc = ensure(c, c->qfrag, frag.size());
    if (rc) return rc;
    rc = ensure(c, c->qn2, nqp * 4);
    if (rc) return rc;
    rc = ensure(c, c->tau, nqp * 4);
    if (rc) return rc;
    rc = ensure(c, c->counts, nqp * 4);
    if (rc) return rc;
    const uint32_t total_tiles = (uint32_t)((n + RT - 1) / RT);
    uint32_t probe_tiles = std::max<uint32_t>(total_tiles / probe_divisor(c, n, nq, k, !is_int), (uint32_t)(4 * k));
    probe_tiles = std::min<uint32_t>(std::min<uint32_t>(probe_tiles, total_tiles), (uint32_t)c->opt_probe_cap);
    const size_t ccap = candidate_capacity(c, k, n, (size_t)probe_tiles * RT);
    rc = ensure(c, c->cand, nqp * ccap * sizeof(uint2));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->qfrag.p, frag.data(), frag.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->qn2.p, qaux.data(), nqp * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->tau.p, tau0.data(), nqp * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->counts.p, 0, nqp * 4, c->stream));

    LowpParams P{};
    P.slabs = t->d_slabs;
    P.aux_slabs = (const uint32_t *const *)t->d_norm_slabs;
    P.slab_shift = t->slab_shift;
    P.slab_mask = (uint32_t)(((size_t)1 << t->slab_shift) - 1);
    P.row_stride = (uint32_t)t->row_bytes;
    P.n_rows = (uint32_t)n;
    P.qfrag = (const uint4 *)c->qfrag.p;
    P.qaux = (const uint32_t *)c->qn2.p;
    if (is_int) {
        P.epi = t->epi == EPI_INT_L2 ? LE_I8_L2 : (t->epi == EPI_INT_IP ? (is_u8 ? LE_U8_IP : LE_I8_IP) : LE_I8_COS);
    } else {
        P.epi = t->metric == VSGPU_L2 ? LE_FP_L2 : LE_FP_IP;
        // bf16*bf16 / fp16*fp16 products are exact in fp32: only accumulation order/rounding differs
        const double u = std::ldexp(1.0, -24);
        const double cq = (double)kdim * std::ldexp(1.0, -22) * 1.01;
        const double gref = ((double)kdim / 16.0 + 12.0) * u;
        P.cE = (float)(((cq + 2.0 * gref + 4.0 * u) * 1.001) * (1.0 + 1e-6));
        P.absE = t->metric == VSGPU_L2 ? 1e-30f : 1e-6f;
    }
    P.tau = (const float *)c->tau.p;
    P.counts = (uint32_t *)c->counts.p;
    P.cand = (uint2 *)c->cand.p;
    P.cap = (uint32_t)ccap;

    const uint32_t tile_step = total_tiles / probe_tiles;
    // k_probe_threshold sorts M group minima per query in LDS; more probe tiles than that are grouped (the k-th
    // smallest group minimum still has k distinct rows at or below it, and with k << M grouping costs nothing)
    uint32_t M = 64;
    while (M < probe_tiles && M < 8192) M <<= 1;
    rc = ensure(c, c->dense, nqp * (size_t)probe_tiles * 4);
    if (rc) return rc;
    const uint32_t wgs = (uint32_t)c->n_cu * 2;

    HIPCHK(hipEventRecord(c->ev_c, c->stream));
    {
        LowpParams Q = P;
        Q.tile_first = 0;
        Q.tile_step = tile_step;
        Q.n_tiles = probe_tiles;
        Q.tilemin = (float *)c->dense.p;
        Q.tilemin_stride = probe_tiles;
        if (qsplit) launch_lowp_i8_split(t, MF_PROBE, Q, dim3(std::min(probe_tiles, wgs), (unsigned)q_tiles), c->stream);
        else launch_lowp(t, MF_PROBE, Q, dim3(std::min(probe_tiles, wgs), (unsigned)q_tiles), c->stream);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_probe_threshold, dim3((unsigned)nq), dim3(1024), M * sizeof(float), c->stream,
                           (const float *)c->dense.p, (size_t)probe_tiles, probe_tiles, (uint32_t)k, M, (float *)c->tau.p);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(c->ev_d, c->stream));
    HIPCHK(hipEventRecord(c->ev_a, c->stream));
    {
        LowpParams Q = P;
        Q.tile_first = 0;
        Q.tile_step = 1;
        Q.n_tiles = total_tiles;
        const uint32_t fw = (uint32_t)c->n_cu * (uint32_t)c->opt_wg_per_cu;
        Q.dbg = (int)c->opt_lowp_dbg;
        uint32_t *d_ph = nullptr;
        const size_t ph_words = (size_t)fw * q_tiles * 16 * 8;
        if (Q.dbg & 8) {
            HIPCHK(hipMalloc(&d_ph, ph_words * 4));
            HIPCHK(hipMemsetAsync(d_ph, 0, ph_words * 4, c->stream));
            Q.tilemin = reinterpret_cast<float *>(d_ph);
        }
        if (qsplit) launch_lowp_i8_split(t, MF_FILTER, Q, dim3(std::min(total_tiles, fw), (unsigned)q_tiles), c->stream);
        else if (!launch_lowp_variant(t, (int)c->opt_lowp_variant, Q, fw, (unsigned)q_tiles, c->stream))
            launch_lowp(t, MF_FILTER, Q, dim3(std::min(total_tiles, fw), (unsigned)q_tiles), c->stream);
        HIPCHK(hipGetLastError());
        if (d_ph) {  // phase sums of every wave: mean cycles per tile, printed once per launch
            std::vector<uint32_t> h(ph_words);
            HIPCHK(hipMemcpyAsync(h.data(), d_ph, ph_words * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            HIPCHK(hipFree(d_ph));
            double sum[5] = {0, 0, 0, 0, 0}, tiles = 0;
            for (size_t w = 0; w < ph_words / 8; w++) {
                if (!h[w * 8 + 5]) continue;
                for (int i = 0; i < 5; i++) sum[i] += h[w * 8 + i];
                tiles += h[w * 8 + 5];
            }
            if (tiles > 0)
                fprintf(stderr, "lowp phases, mean s_memtime ticks per wave and tile: vmcnt-wait %.0f  barrier %.0f  refill-request %.0f  "
                                "reads+mfma-issue %.0f  epilogue %.0f\n",
                        sum[0] / tiles, sum[1] / tiles, sum[2] / tiles, sum[3] / tiles, sum[4] / tiles);
        }
    }
    HIPCHK(hipEventRecord(c->ev_b, c->stream));
    if (c->opt_lowp_dbg) {  // diagnosis run: the kernel's output is meaningless, report its time only
        HIPCHK(hipStreamSynchronize(c->stream));
        account_scan(c, t, n, 1, "k_mfma_filter_lowp(dbg)");
        for (size_t q = 0; q < nq; q++) counts[q] = 0;
        return VSGPU_OK;
    }
    if (!is_int) {
        ScanParams S{};
        S.slabs = t->d_slabs;
        S.slab_shift = t->slab_shift;
        S.slab_mask = P.slab_mask;
        S.row_stride = P.row_stride;
        S.offs = t->d_offs;
        S.steps = t->prog.steps;
        S.qperm = c->qperm.p;
        S.nq = (int)nq;
        S.epilogue = t->epi;
        S.counts = (uint32_t *)c->counts.p;
        S.cand = (uint2 *)c->cand.p;
        S.cap = (uint32_t)ccap;
        dim3 grid(64, (unsigned)nq);
        const bool l2 = (t->opk == OP_L2_FMA);
        if (t->type == VSGPU_BF16) {
            if (l2) hipLaunchKernelGGL((k_exact_pairs<EK_BF16, OP_L2_FMA>), grid, dim3(256), 0, c->stream, S);
            else hipLaunchKernelGGL((k_exact_pairs<EK_BF16, OP_IP_FMA>), grid, dim3(256), 0, c->stream, S);
        } else {
            if (l2) hipLaunchKernelGGL((k_exact_pairs<EK_F16, OP_L2_FMA>), grid, dim3(256), 0, c->stream, S);
            else hipLaunchKernelGGL((k_exact_pairs<EK_F16, OP_IP_FMA>), grid, dim3(256), 0, c->stream, S);
        }
        HIPCHK(hipGetLastError());
    }
    return collect_candidates(t, queries, nq, qstride, k, cap, ccap, ids, scores, counts,
                              is_int ? "k_mfma_filter_lowp(i8)" : "k_mfma_filter_lowp(h16)");
}

// ------------------------------------------------------------------ top-K
static int topk_dense_path(vsgpu_table *t, size_t nq, size_t k, size_t cap, uint32_t *ids, double *scores,
                           uint32_t *counts, size_t q_first, size_t q_count, const void *queries, size_t qstride) {
    // queries [q_first, q_first+q_count) answered from full score vectors
    const size_t n = t->n;
    std::vector<double> row(n);
    std::vector<Hit> hits;
    for (size_t q = q_first; q < q_first + q_count; q++) {
        int rc = vsgpu_scores(t, (const char *)queries + q * qstride, 0, n, row.data());
        if (rc) return rc;
        hits.resize(n);
        for (size_t i = 0; i < n; i++) hits[i] = Hit{(uint32_t)i, row[i]};
        select_upto_kth(hits, k);
        emit(hits, q, cap, ids, scores, counts);
    }
    (void)nq;
    return VSGPU_OK;
}

extern "C" int vsgpu_topk(vsgpu_table *t, const void *queries, size_t nq, size_t qstride, size_t k, size_t cap,
                          uint32_t *ids, double *scores, uint32_t *counts) {
    vsgpu_ctx *c = t->ctx;
    if (nq == 0) return VSGPU_OK;
    if (k == 0 || t->n == 0) {
        for (size_t q = 0; q < nq; q++) counts[q] = 0;
        return VSGPU_OK;
    }
    HIPCHK(hipSetDevice(c->device));
    const size_t n = t->n;
    const bool f64 = (t->type == VSGPU_F64);

    // small problems (and fp64): one dense score matrix, selection on the host
    if (f64 || (double)n * (double)nq <= (double)c->opt_dense_pairs || n <= 4 * k) {
        if (!f64 && n * nq * 4 <= ((size_t)1 << 28)) {
            int rc = stage_queries(t, queries, nq, qstride);
            if (rc) return rc;
            HIPCHK(hipEventRecord(c->ev_c, c->stream));
            ScanParams P{};
            // dense_to_host re-records nothing: time it here as the scan
            rc = ensure(c, c->dense, nq * n * 4);
            if (rc) return rc;
            P.row_ids = nullptr;
            P.row_begin = 0;
            P.row_end = (uint32_t)n;
            P.n_compact = (uint32_t)n;
            P.tile_step = (uint32_t)tile_rows_of(t->ek);
            P.mode = MODE_DENSE;
            P.out = c->dense.p;
            P.out_stride = n;
            rc = run_scan(t, P, nq, true);
            if (rc) return rc;
            // selection on the GPU: only the rows with score <= T_k travel to the host
            const size_t ocap = cap;
            rc = ensure(c, c->sel, nq * ocap * sizeof(uint2));
            if (rc) return rc;
            rc = ensure(c, c->selcnt, nq * 4);
            if (rc) return rc;
            hipLaunchKernelGGL(k_select_dense_upto_kth, dim3((unsigned)nq), dim3(1024), 0, c->stream, (const float *)c->dense.p,
                               n, (uint32_t)n, (uint32_t)std::min(k, n), (uint2 *)c->sel.p, (uint32_t *)c->selcnt.p,
                               (uint32_t)ocap);
            HIPCHK(hipGetLastError());
            rc = ensure_pinned(c, nq * 4 + nq * ocap * sizeof(uint2));
            if (rc) return rc;
            uint32_t *hsel = (uint32_t *)c->pinned;
            uint2 *hrec = (uint2 *)((char *)c->pinned + nq * 4);
            HIPCHK(hipMemcpyAsync(hsel, c->selcnt.p, nq * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(hrec, c->sel.p, nq * ocap * sizeof(uint2), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            account_scan(c, t, n, 1, "k_exact_scan(dense)");
            std::vector<Hit> hits;
            for (size_t q = 0; q < nq; q++) {
                if (hsel[q] == VSGPU_COUNT_OVERFLOW) {
                    counts[q] = VSGPU_COUNT_OVERFLOW;
                    continue;
                }
                hits.resize(hsel[q]);
                for (size_t i = 0; i < hsel[q]; i++) {
                    float f;
                    memcpy(&f, &hrec[q * ocap + i].y, 4);
                    hits[i] = Hit{hrec[q * ocap + i].x, (double)f};
                }
                std::sort(hits.begin(), hits.end(), [](const Hit &a, const Hit &b) { return a.id < b.id; });
                emit(hits, q, cap, ids, scores, counts);
            }
            return VSGPU_OK;
        }
        if (f64) {
            // fp64: dense double scores per group of queries, 64-bit selection on the GPU, survivors to the host
            const size_t ocap = cap;
            const size_t group = std::max<size_t>(1, std::min<size_t>(nq, ((size_t)1 << 30) / (n * 8)));
            int rc = ensure(c, c->dense, group * n * 8);
            if (rc) return rc;
            rc = ensure(c, c->sel, group * ocap * sizeof(SelRec64));
            if (rc) return rc;
            rc = ensure(c, c->selcnt, group * 4);
            if (rc) return rc;
            std::vector<uint32_t> hcnt(group);
            std::vector<SelRec64> hrec(group * ocap);
            std::vector<Hit> hits;
            for (size_t q0 = 0; q0 < nq; q0 += group) {
                const size_t g = std::min(group, nq - q0);
                rc = stage_queries(t, (const char *)queries + q0 * qstride, g, qstride);
                if (rc) return rc;
                ScanParams P{};
                P.row_ids = nullptr;
                P.row_begin = 0;
                P.row_end = (uint32_t)n;
                P.n_compact = (uint32_t)n;
                P.tile_step = (uint32_t)tile_rows_of(t->ek);
                P.mode = MODE_DENSE;
                P.out = c->dense.p;
                P.out_stride = n;
                rc = run_scan(t, P, g, true);
                if (rc) return rc;
                hipLaunchKernelGGL(k_select_dense_upto_kth_f64, dim3((unsigned)g), dim3(1024), 0, c->stream, (const double *)c->dense.p, n,
                                   (uint32_t)n, (uint32_t)std::min(k, n), (SelRec64 *)c->sel.p, (uint32_t *)c->selcnt.p, (uint32_t)ocap);
                HIPCHK(hipGetLastError());
                HIPCHK(hipMemcpyAsync(hcnt.data(), c->selcnt.p, g * 4, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipMemcpyAsync(hrec.data(), c->sel.p, g * ocap * sizeof(SelRec64), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                account_scan(c, t, n, 1, "k_exact_scan(dense f64)");
                for (size_t j = 0; j < g; j++) {
                    if (hcnt[j] == VSGPU_COUNT_OVERFLOW) {
                        counts[q0 + j] = VSGPU_COUNT_OVERFLOW;
                        continue;
                    }
                    hits.resize(hcnt[j]);
                    for (size_t i = 0; i < hcnt[j]; i++) {
                        double d;
                        memcpy(&d, &hrec[j * ocap + i].bits, 8);
                        hits[i] = Hit{(uint32_t)hrec[j * ocap + i].row, d};
                    }
                    std::sort(hits.begin(), hits.end(), [](const Hit &a, const Hit &b) { return a.id < b.id; });
                    emit(hits, q0 + j, cap, ids, scores, counts);
                }
            }
            return VSGPU_OK;
        }
        return topk_dense_path(t, nq, k, cap, ids, scores, counts, 0, nq, queries, qstride);
    }

    if (t->lowp_ok && c->opt_mfma && nq >= (size_t)c->opt_mfma_min_q) {
        WallMarks w;
        int rc = topk_lowp(t, queries, nq, qstride, k, cap, ids, scores, counts);
        w.mark("vsgpu_topk_total");
        w.flush("lowp");
        return rc;
    }
    if (t->mfma_ok && c->opt_mfma && nq >= (size_t)c->opt_mfma_min_q) {
        WallMarks w;
        int rc = topk_mfma(t, queries, nq, qstride, k, cap, ids, scores, counts);
        w.mark("vsgpu_topk_total");
        w.flush("mfma");
        return rc;
    }

    // ---- probe -> threshold -> filtered scan ----
    int rc = stage_queries(t, queries, nq, qstride);
    if (rc) return rc;
    const int tile_rows = tile_rows_of(t->ek);
    const size_t total_tiles = (n + tile_rows - 1) / tile_rows;
    size_t probe_tiles = std::max<size_t>(total_tiles / (size_t)probe_divisor(c, n, nq, k, false), (64 * k + tile_rows - 1) / tile_rows);
    probe_tiles = std::min(probe_tiles, total_tiles);
    const size_t tile_stride = total_tiles / probe_tiles;  // >= 1
    const size_t n0 = std::min(n, probe_tiles * (size_t)tile_rows);
    uint32_t M = 1024;
    while (M < 64 * k && M < 8192) M <<= 1;
    while (M > 64 && M > n0) M >>= 1;

    rc = ensure(c, c->dense, nq * n0 * 4);
    if (rc) return rc;
    rc = ensure(c, c->tau, nq * 4);
    if (rc) return rc;
    rc = ensure(c, c->counts, nq * 4);
    if (rc) return rc;
    const size_t ccap = candidate_capacity(c, k, n, n0);
    rc = ensure(c, c->cand, nq * ccap * sizeof(uint2));
    if (rc) return rc;

    HIPCHK(hipEventRecord(c->ev_c, c->stream));
    {
        ScanParams P{};
        P.row_ids = nullptr;
        P.row_begin = 0;
        P.row_end = (uint32_t)n;
        P.n_compact = (uint32_t)n0;
        P.tile_step = (uint32_t)(tile_stride * tile_rows);
        P.mode = MODE_DENSE;
        P.out = c->dense.p;
        P.out_stride = n0;
        rc = run_scan(t, P, nq, false);
        if (rc) return rc;
    }
    // rows of the last probe tile may lie beyond the table (skipped, left uninitialised): bound n0
    // to the rows actually written.  With tile_stride*tile_rows spacing only the final tile can be short.
    size_t last_tile_first = (probe_tiles - 1) * tile_stride * (size_t)tile_rows;
    size_t n0_valid = (probe_tiles - 1) * (size_t)tile_rows + std::min<size_t>(tile_rows, n - last_tile_first);
    hipLaunchKernelGGL(k_probe_threshold, dim3((unsigned)nq), dim3(1024), M * sizeof(float), c->stream,
                       (const float *)c->dense.p, n0, (uint32_t)n0_valid, (uint32_t)k, M, (float *)c->tau.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(c->counts.p, 0, nq * 4, c->stream));
    HIPCHK(hipEventRecord(c->ev_d, c->stream));
    {
        ScanParams P{};
        P.row_ids = nullptr;
        P.row_begin = 0;
        P.row_end = (uint32_t)n;
        P.n_compact = (uint32_t)n;
        P.tile_step = (uint32_t)tile_rows;
        P.mode = MODE_FILTER;
        P.tau = c->tau.p;
        P.counts = (uint32_t *)c->counts.p;
        P.cand = (uint2 *)c->cand.p;
        P.cap = (uint32_t)ccap;
        rc = run_scan(t, P, nq, true);
        if (rc) return rc;
    }
    return collect_candidates(t, queries, nq, qstride, k, cap, ccap, ids, scores, counts, "k_exact_scan(filter)");
}

// ------------------------------------------------------------------ range
extern "C" int vsgpu_range(vsgpu_table *t, const void *query, double radius, size_t cap, uint32_t *ids,
                           double *scores, uint32_t *count) {
    vsgpu_ctx *c = t->ctx;
    *count = 0;
    if (t->n == 0) return VSGPU_OK;
    HIPCHK(hipSetDevice(c->device));
    const size_t n = t->n;
    // dense scores, threshold on the host: the comparison is `score <= radius` on the DistType value
    // (brute_force.h:305-314).  (A filtered scan is used once the table is large.)
    std::vector<Hit> hits;
    if (t->type == VSGPU_F64 || n <= ((size_t)1 << 20)) {
        std::vector<double> row(n);
        int rc = vsgpu_scores(t, query, 0, n, row.data());
        if (rc) return rc;
        const double r = (t->type == VSGPU_F64) ? radius : (double)(float)radius;
        for (size_t i = 0; i < n; i++)
            if (row[i] <= r) hits.push_back(Hit{(uint32_t)i, row[i]});
    } else {
        int rc = stage_queries(t, query, 1, 0);
        if (rc) return rc;
        const size_t ccap = std::max<size_t>((size_t)c->opt_cand_cap, cap);
        rc = ensure(c, c->tau, 4);
        if (rc) return rc;
        rc = ensure(c, c->counts, 4);
        if (rc) return rc;
        rc = ensure(c, c->cand, ccap * sizeof(uint2));
        if (rc) return rc;
        float rf = (float)radius;
        HIPCHK(hipMemcpyAsync(c->tau.p, &rf, 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemsetAsync(c->counts.p, 0, 4, c->stream));
        ScanParams P{};
        P.row_begin = 0;
        P.row_end = (uint32_t)n;
        P.n_compact = (uint32_t)n;
        P.tile_step = (uint32_t)tile_rows_of(t->ek);
        P.mode = MODE_FILTER;
        P.tau = c->tau.p;
        P.counts = (uint32_t *)c->counts.p;
        P.cand = (uint2 *)c->cand.p;
        P.cap = (uint32_t)ccap;
        rc = run_scan(t, P, 1, true);
        if (rc) return rc;
        uint32_t hcount = 0;
        HIPCHK(hipMemcpyAsync(&hcount, c->counts.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        account_scan(c, t, n, 1, "k_exact_scan(filter)");
        if (hcount > ccap) {
            *count = VSGPU_COUNT_OVERFLOW;
            return VSGPU_OK;
        }
        std::vector<uint2> hc(std::max<uint32_t>(hcount, 1));
        if (hcount) {
            HIPCHK(hipMemcpyAsync(hc.data(), c->cand.p, hcount * sizeof(uint2), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
        hits.resize(hcount);
        for (uint32_t i = 0; i < hcount; i++) {
            float f;
            memcpy(&f, &hc[i].y, 4);
            hits[i] = Hit{hc[i].x, (double)f};
        }
        std::sort(hits.begin(), hits.end(), [](const Hit &a, const Hit &b) { return a.id < b.id; });
    }
    if (hits.size() > cap) {
        *count = VSGPU_COUNT_OVERFLOW;
        return VSGPU_OK;
    }
    *count = (uint32_t)hits.size();
    for (size_t i = 0; i < hits.size(); i++) {
        ids[i] = hits[i].id;
        scores[i] = hits[i].score;
    }
    return VSGPU_OK;
}

// ------------------------------------------------------------------ HNSW graph snapshot + search
struct vsgpu_graph {
    vsgpu_table *t = nullptr;
    uint32_t M = 16, M0 = 32;
    size_t n = 0;
    DevBuf links0, cnt0, upper_off, upper, deleted, labels;
    uint32_t entry = 0xFFFFFFFFu;
    int max_level = -1;
    // visited tags: one u16 per node per resident search wave
    DevBuf tags, slot_epoch;
    size_t tag_slots = 0, tag_n = 0;
    DevBuf out_labels, out_scores, out_counts, stat;
};

extern "C" vsgpu_graph *vsgpu_graph_create(vsgpu_table *t, size_t M) {
    if (!t || M < 2 || M > 32) {
        fail(VSGPU_ERR_ARG, "graph: M must be in [2, 32] (2M neighbours are scored by one wavefront)");
        return nullptr;
    }
    if (t->type == VSGPU_F64) {
        fail(VSGPU_ERR_UNSUPPORTED, "graph search: fp64 tables are not supported yet");
        return nullptr;
    }
    vsgpu_graph *g = new vsgpu_graph();
    g->t = t;
    g->M = (uint32_t)M;
    g->M0 = (uint32_t)(2 * M);
    return g;
}
extern "C" void vsgpu_graph_destroy(vsgpu_graph *g) {
    if (!g) return;
    (void)hipSetDevice(g->t->ctx->device);
    (void)hipStreamSynchronize(g->t->ctx->stream);
    for (DevBuf *b : {&g->links0, &g->cnt0, &g->upper_off, &g->upper, &g->deleted, &g->labels, &g->tags, &g->slot_epoch,
                      &g->out_labels, &g->out_scores, &g->out_counts, &g->stat})
        if (b->p) (void)hipFree(b->p);
    delete g;
}
extern "C" int vsgpu_graph_upload(vsgpu_graph *g, size_t n, const uint32_t *links0, const uint16_t *cnt0,
                                  const uint32_t *upper_off, const uint32_t *upper, size_t upper_words,
                                  const uint8_t *deleted, const uint64_t *labels, uint32_t entry, int max_level) {
    vsgpu_ctx *c = g->t->ctx;
    HIPCHK(hipSetDevice(c->device));
    if (n > g->t->n) return fail(VSGPU_ERR_ARG, "graph has %zu nodes but the table holds %zu rows", n, g->t->n);
    int rc;
    if ((rc = ensure(c, g->links0, n * g->M0 * 4))) return rc;
    if ((rc = ensure(c, g->cnt0, n * 2))) return rc;
    if ((rc = ensure(c, g->upper_off, n * 4))) return rc;
    if ((rc = ensure(c, g->upper, std::max<size_t>(upper_words, 1) * 4))) return rc;
    if ((rc = ensure(c, g->deleted, n))) return rc;
    if ((rc = ensure(c, g->labels, n * 8))) return rc;
    if (n) {
        HIPCHK(hipMemcpyAsync(g->links0.p, links0, n * g->M0 * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(g->cnt0.p, cnt0, n * 2, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(g->upper_off.p, upper_off, n * 4, hipMemcpyHostToDevice, c->stream));
        if (upper_words) HIPCHK(hipMemcpyAsync(g->upper.p, upper, upper_words * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(g->deleted.p, deleted, n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(g->labels.p, labels, n * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));  // the caller's arrays are borrowed for this call only
    }
    g->n = n;
    g->entry = entry;
    g->max_level = max_level;
    return VSGPU_OK;
}

template <int EK> static void launch_hnsw_ek(int opk, const HnswParams &P, dim3 grid, size_t lds, hipStream_t s) {
    if (opk == OP_L2_FMA) hipLaunchKernelGGL((k_hnsw_search<EK, OP_L2_FMA>), grid, dim3(64), lds, s, P);
    else if (opk == OP_IP_FMA) hipLaunchKernelGGL((k_hnsw_search<EK, OP_IP_FMA>), grid, dim3(64), lds, s, P);
    else if (opk == OP_L2_MULADD) hipLaunchKernelGGL((k_hnsw_search<EK, OP_L2_MULADD>), grid, dim3(64), lds, s, P);
    else hipLaunchKernelGGL((k_hnsw_search<EK, OP_IP_MULADD>), grid, dim3(64), lds, s, P);
}

// top-k search (range == nullptr) or range search (range = {radius, epsilon}; k is then the result capacity per query)
static int graph_run(vsgpu_graph *g, const void *queries, size_t nq, size_t qstride, size_t k, size_t ef, const double *range,
                     uint64_t *labels, double *scores, uint32_t *counts, uint64_t *dist_evals) {
    vsgpu_table *t = g->t;
    vsgpu_ctx *c = t->ctx;
    if (dist_evals) *dist_evals = 0;
    if (nq == 0) return VSGPU_OK;
    if (k == 0 || g->n == 0 || g->entry == 0xFFFFFFFFu) {
        for (size_t q = 0; q < nq; q++) counts[q] = 0;
        return VSGPU_OK;
    }
    HIPCHK(hipSetDevice(c->device));
    ef = range ? 1 : std::max(ef, k);
    if (ef > 4096) return fail(VSGPU_ERR_UNSUPPORTED, "ef %zu too large for the LDS heaps", ef);
    int rc = stage_queries(t, queries, nq, qstride);
    if (rc) return rc;
    const size_t ab = acc_bytes(t->type);
    size_t ccap = 2 * ef;
    if (range) {
        // the reference's candidate set is unbounded: give the window what LDS allows (overflow is reported)
        const size_t fixed = 2 * (((size_t)t->prog.steps * t->prog.vl * std::max<size_t>(ab, 4) + 15) & ~(size_t)15) + 1024;
        ccap = 64;
        while (ccap < 3072 && fixed + (2 * (2 * ccap) + 2) * 8 + 64 <= 60 * 1024) ccap *= 2;
    }
    size_t lds = (((size_t)t->prog.steps * t->prog.vl * 4 + 15) & ~(size_t)15);
    lds += (((size_t)t->prog.steps * t->prog.vl * ab + 15) & ~(size_t)15);
    lds += (ef + 2) * 8;
    lds += (((ef + 2) * 4 + 15) & ~(size_t)15);
    lds += (2 * ccap + 2) * 4;
    lds += (((2 * ccap + 2) * 4 + 15) & ~(size_t)15);
    lds += 64 * 4 + 64 * 4;
    if (lds > 64 * 1024) return fail(VSGPU_ERR_UNSUPPORTED, "ef %zu / dim %zu need %zu B of LDS per query", ef, t->dim, lds);
    // resident search waves = tag slots
    const size_t slots = std::min<size_t>(nq, (size_t)c->n_cu * (size_t)c->opt_hnsw_slots);
    if (slots > g->tag_slots || g->n > g->tag_n) {
        const size_t ns = std::max(slots, g->tag_slots), nn = std::max(g->n, g->tag_n);
        // grow with headroom on the node axis: the graph usually keeps growing between searches
        const size_t nn2 = std::max(nn, g->tag_n + g->tag_n / 2);
        if ((rc = ensure(c, g->tags, ns * nn2 * 2))) return rc;
        if ((rc = ensure(c, g->slot_epoch, ns * 4))) return rc;
        HIPCHK(hipMemsetAsync(g->tags.p, 0, ns * nn2 * 2, c->stream));
        HIPCHK(hipMemsetAsync(g->slot_epoch.p, 0, ns * 4, c->stream));
        g->tag_slots = ns;
        g->tag_n = nn2;
    }
    if ((rc = ensure(c, g->out_labels, nq * k * 8))) return rc;
    if ((rc = ensure(c, g->out_scores, nq * k * 4))) return rc;
    if ((rc = ensure(c, g->out_counts, nq * 4))) return rc;
    if ((rc = ensure(c, g->stat, 16))) return rc;
    HIPCHK(hipMemsetAsync(g->stat.p, 0, 16, c->stream));

    HnswParams P{};
    P.slabs = t->d_slabs;
    P.slab_shift = t->slab_shift;
    P.slab_mask = (uint32_t)(((size_t)1 << t->slab_shift) - 1);
    P.row_stride = (uint32_t)t->row_bytes;
    P.offs = t->d_offs;
    P.steps = t->prog.steps;
    P.qperm = c->qperm.p;
    P.nq = (int)nq;
    P.epilogue = t->epi;
    P.norm_off = (uint32_t)t->dim;
    P.qnorm = (const float *)c->qnorm.p;
    P.links0 = (const uint32_t *)g->links0.p;
    P.cnt0 = (const uint16_t *)g->cnt0.p;
    P.upper_off = (const uint32_t *)g->upper_off.p;
    P.upper = (const uint32_t *)g->upper.p;
    P.deleted = (const uint8_t *)g->deleted.p;
    P.labels = (const uint64_t *)g->labels.p;
    P.M0 = g->M0;
    P.M = g->M;
    P.entry = g->entry;
    P.max_level = g->max_level;
    P.n = (uint32_t)g->tag_n;  // tag row pitch
    P.tags = (uint16_t *)g->tags.p;
    P.slot_epoch = (uint32_t *)g->slot_epoch.p;
    P.ef = (uint32_t)ef;
    P.k = (uint32_t)k;
    P.ccap = (uint32_t)ccap;
    P.out_labels = (uint64_t *)g->out_labels.p;
    P.out_scores = (float *)g->out_scores.p;
    P.out_counts = (uint32_t *)g->out_counts.p;
    P.stat_dists = (uint64_t *)g->stat.p;
    P.next_query = (uint32_t *)((char *)g->stat.p + 8);
    if (range) {
        P.range = 1;
        P.radius = (float)range[0];
        P.epsilon = range[1];
        P.rcap = (uint32_t)k;
    }
    HIPCHK(hipEventRecord(c->ev_a, c->stream));
    const dim3 grid((unsigned)slots);
    switch (t->ek) {
    case EK_F32: launch_hnsw_ek<EK_F32>(t->opk, P, grid, lds, c->stream); break;
    case EK_BF16: launch_hnsw_ek<EK_BF16>(t->opk, P, grid, lds, c->stream); break;
    case EK_F16: launch_hnsw_ek<EK_F16>(t->opk, P, grid, lds, c->stream); break;
    case EK_I8: launch_hnsw_ek<EK_I8>(t->opk, P, grid, lds, c->stream); break;
    default: launch_hnsw_ek<EK_U8>(t->opk, P, grid, lds, c->stream); break;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev_b, c->stream));
    std::vector<float> hs(nq * k);
    uint64_t hstat = 0;
    HIPCHK(hipMemcpyAsync(labels, g->out_labels.p, nq * k * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(hs.data(), g->out_scores.p, nq * k * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(counts, g->out_counts.p, nq * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&hstat, g->stat.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < nq * k; i++) scores[i] = (double)hs[i];
    if (dist_evals) *dist_evals = hstat;
    {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev_a, c->ev_b) == hipSuccess) {
            c->stats.scan_ms += ms;
            c->stats.scan_launches += 1;
            c->stats.scan_rows += hstat;                    // rows gathered = distance evaluations
            c->stats.scan_bytes += hstat * t->row_bytes;
            snprintf(c->stats.scan_kernel, sizeof c->stats.scan_kernel, range ? "k_hnsw_search(range)" : "k_hnsw_search");
        }
    }
    return VSGPU_OK;
}
extern "C" int vsgpu_graph_search(vsgpu_graph *g, const void *queries, size_t nq, size_t qstride, size_t k, size_t ef,
                                  uint64_t *labels, double *scores, uint32_t *counts, uint64_t *dist_evals) {
    return graph_run(g, queries, nq, qstride, k, ef, nullptr, labels, scores, counts, dist_evals);
}
extern "C" int vsgpu_graph_range(vsgpu_graph *g, const void *queries, size_t nq, size_t qstride, double radius,
                                 double epsilon, size_t cap, uint64_t *labels, double *scores, uint32_t *counts,
                                 uint64_t *dist_evals) {
    if (cap == 0) return fail(VSGPU_ERR_ARG, "range search needs room for results");
    const double range[2] = {radius, epsilon};
    return graph_run(g, queries, nq, qstride, cap, 1, range, labels, scores, counts, dist_evals);
}
