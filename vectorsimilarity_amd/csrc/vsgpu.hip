// vsgpu.hip -- implementation of include/vsgpu.h: device table + query orchestration (gfx950).
//
// No CPU fallback lives here: if there is no HIP device every entry point fails with
// VSGPU_ERR_NO_DEVICE and a message.  Host code in this file only moves bytes, builds lane tables,
// launches kernels and selects among *GPU-computed* scores.  The MFMA filter paths, the HNSW search and
// the RCCL exchange live in their own translation units (vsgpu_mfma.hip, vsgpu_lowp.hip, vsgpu_hnsw.hip,
// vsgpu_comm.hip); vsgpu_internal.hpp is what they share.
#include "vsgpu_internal.hpp"
#include "mfma_kernels.hpp"
#include "mfma_lowp_kernels.hpp"
#include "iter_kernels.hpp"

using namespace vsg;

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
int vsg_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

extern "C" const char *vsgpu_last_error(void) { return g_err.c_str(); }

extern "C" int vsgpu_device_synchronize(int device) {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    return VSGPU_OK;
}
extern "C" int vsgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}


// VSGPU_POISON=<byte>: fill every fresh device allocation with that byte (test aid: makes any read of memory the
// library never wrote deterministic instead of depending on what the allocator hands back)
int poison_byte() {
    static const int v = [] {
        const char *e = getenv("VSGPU_POISON");
        return e ? (int)(strtol(e, nullptr, 0) & 0xFF) : -1;
    }();
    return v;
}
void poison(void *p, size_t bytes) {
    if (poison_byte() >= 0 && p) {
        (void)hipMemset(p, poison_byte(), bytes);
        (void)hipDeviceSynchronize();
    }
}
int ensure(vsgpu_ctx *c, DevBuf &b, size_t bytes) {
    if (b.alias) {   // an alias is only good for the call that made it: back to the buffer's own memory
        b.p = b.own_p;
        b.cap = b.own_cap;
        b.alias = false;
        b.own_p = nullptr;
        b.own_cap = 0;
    }
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
// b becomes a region of another buffer (its own memory, if any, is released: every call ends with a stream sync, nothing reads it)
void alias_into(DevBuf &b, void *p, size_t bytes) {
    if (b.p && !b.alias) {   // park the buffer's own allocation (ensure() takes it back)
        if (b.own_p && b.own_p != b.p) (void)hipFree(b.own_p);
        b.own_p = b.p;
        b.own_cap = b.cap;
    }
    b.p = p;
    b.cap = bytes;
    b.alias = true;
}
int ensure_pin_up(vsgpu_ctx *c, size_t bytes) {
    if (bytes <= c->pin_up_cap) return VSGPU_OK;
    if (c->pin_up) {
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipHostFree(c->pin_up));
    }
    c->pin_up = nullptr;
    c->pin_up_cap = 0;
    const size_t want = (std::max(bytes, (size_t)1 << 18) + 0xFFFF) & ~(size_t)0xFFFF;
    HIPCHK(hipHostMalloc(&c->pin_up, want, hipHostMallocDefault));
    c->pin_up_cap = want;
    return VSGPU_OK;
}
// 16 bytes per thread from pinned host memory (read over PCIe, uncached) into device memory
typedef unsigned int upload_u32x4 __attribute__((ext_vector_type(4)));
static __global__ __launch_bounds__(256) void k_upload_block(upload_u32x4 *__restrict__ dst, const upload_u32x4 *__restrict__ src, uint32_t n16) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = __builtin_nontemporal_load(src + i);
}
int upload_block(vsgpu_ctx *c, void *dev, const void *pinned_src, size_t bytes) {
    if (bytes == 0) return VSGPU_OK;
    if (!c->opt_upload_kernel || bytes > ((size_t)4 << 20) || (bytes & 15) || ((uintptr_t)dev & 15) || ((uintptr_t)pinned_src & 15)) {
        HIPCHK(hipMemcpyAsync(dev, pinned_src, bytes, hipMemcpyHostToDevice, c->stream));
        return VSGPU_OK;
    }
    const uint32_t n16 = (uint32_t)(bytes / 16);
    hipLaunchKernelGGL(k_upload_block, dim3(std::min<uint32_t>((n16 + 255) / 256, 512)), dim3(256), 0, c->stream, (upload_u32x4 *)dev,
                       (const upload_u32x4 *)pinned_src, n16);
    HIPCHK(hipGetLastError());
    return VSGPU_OK;
}
int ensure_pinned(vsgpu_ctx *c, size_t bytes) {
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
        // timing events WITHOUT the system-scope release a default event performs (a cache write-back so that the HOST sees device
        // writes: nothing reads through these events, the stream's own synchronisation orders the download): a pair around a kernel
        // costs its stream 2.8 us instead of 6.7 (tools/stream/event_cost.hip, profiles/r06_event_cost.txt)
        hipEventCreateWithFlags(&c->ev_a, hipEventDisableSystemFence) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_b, hipEventDisableSystemFence) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_c, hipEventDisableSystemFence) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_d, hipEventDisableSystemFence) != hipSuccess) {
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
    for (DevBuf *b : {&c->qperm, &c->qnorm, &c->dense, &c->tau, &c->counts, &c->cand, &c->ids, &c->qfrag, &c->qfrag2, &c->qn2, &c->sel, &c->selcnt, &c->qmeta, &c->klist, &c->qblock}) {
        if (b->p && !b->alias) (void)hipFree(b->p);
        if (b->own_p && (b->alias || b->own_p != b->p)) (void)hipFree(b->own_p);
    }
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pin_up) (void)hipHostFree(c->pin_up);
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
extern "C" void vsgpu_set_poll(vsgpu_ctx *c, int (*poll)(void *), void *user) {
    c->poll = poll;
    c->poll_user = user;
}
extern "C" int vsgpu_set_option(vsgpu_ctx *c, const char *name, long value) {
    std::string n(name);
    if (n == "mfma") c->opt_mfma = value;
    else if (n == "mfma_variant") c->opt_mfma_variant = value;
    else if (n == "lowp_variant") c->opt_lowp_variant = value;
    else if (n == "lowp_dbg") c->opt_lowp_dbg = value;
    else if (n == "wide_blocks") c->opt_wide_blocks = value;
    else if (n == "wide_gx") c->opt_wide_gx = value;
    else if (n == "lowp_wg_per_cu") c->opt_lowp_wg_per_cu = std::max(1L, value);
    else if (n == "lowp_ksplit") c->opt_lowp_ksplit = value;
    else if (n == "lowp_x32") c->opt_lowp_x32 = value;
    else if (n == "hnsw_slots") c->opt_hnsw_slots = std::max(1L, std::min(64L, value));
    else if (n == "lowp_qsplit") c->opt_lowp_qsplit = value;
    else if (n == "lowp_narrow") c->opt_lowp_narrow = value;
    else if (n == "chain_early") c->opt_chain_early = value;
    else if (n == "stream_tau") c->opt_stream_tau = value;
    else if (n == "stream_probe_tiles") c->opt_stream_probe_tiles = std::max(16L, value);
    else if (n == "stream_stride") c->opt_stream_stride = std::min(std::max(value, (long)MF_KLIST), 65536L);
    else if (n == "stream_early") c->opt_stream_early = std::max(0L, value);
    else if (n == "stream_refresh") {   // tiles between two re-reads of a query's threshold, rounded down to a power of two
        long r = 1;
        while (r * 2 <= std::min(std::max(value, 1L), 1024L)) r *= 2;
        c->opt_stream_refresh = r;
    }
    else if (n == "upload_kernel") c->opt_upload_kernel = value;
    else if (n == "sel_mapped") c->opt_sel_mapped = value;
    else if (n == "events") c->opt_events = value & 3;
    else if (n == "probe_rt16") c->opt_probe_rt16 = value;
    else if (n == "sq8_block") c->opt_sq8_block = value;
    else if (n == "wg_per_cu") c->opt_wg_per_cu = c->opt_lowp_wg_per_cu = std::max(1L, value);   // (both filter families)
    else if (n == "mfma_min_q") c->opt_mfma_min_q = std::max(1L, value);
    else if (n == "dense_pairs") c->opt_dense_pairs = value;
    else if (n == "dense_sliced_bytes") c->opt_dense_sliced_bytes = std::max(0L, value);
    else if (n == "dense_small_q") c->opt_dense_small_q = std::max(1L, value);
    else if (n == "probe_div") c->opt_probe_div = std::max(0L, value);
    else if (n == "probe_cap") c->opt_probe_cap = std::min(1L << 20, std::max(64L, value));
    else if (n == "probe_run") c->opt_probe_run = std::min(8L, std::max(-1L, value));
    else if (n == "cand_cap") c->opt_cand_cap = std::max(16L, value);
    else return fail(VSGPU_ERR_ARG, "unknown option %s", name);
    return VSGPU_OK;
}

// ------------------------------------------------------------------ table

extern "C" vsgpu_table *vsgpu_table_create(vsgpu_ctx *c, int type, int metric, int tier, size_t dim,
                                           size_t row_bytes) {
    if (!c) {
        fail(VSGPU_ERR_ARG, "null context");
        return nullptr;
    }
    if (type < VSGPU_F32 || type > VSGPU_SQ8H || metric < VSGPU_L2 || metric > VSGPU_COSINE || dim == 0 ||
        row_bytes < dim * (size_t)elem_bytes_of(type) ||
        ((type == VSGPU_SQ8 || type == VSGPU_SQ8H) && row_bytes != dim + (metric == VSGPU_L2 ? 16 : 12) &&
         !(metric == VSGPU_IP && row_bytes == dim + 16))) {   // (IP rows of dim + 16 bytes: mean-centred, see vsgpu.h)
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
    // bf16 IP on the avx512_bf16 tier: the vdpbf16ps step (odd element, then even, each with FTZ)
    if (t->prog.dpbf16) t->opk = OP_IP_DPBF16;
    // fp16 rows on the AVX512-FP16 tier: half-precision accumulators (exact kernels only: prog.reduce = 2 keeps the MFMA filters off)
    if (t->prog.f16acc) t->opk = l2 ? OP_L2_F16ACC : OP_IP_F16ACC;
    if (is_int) t->epi = l2 ? EPI_INT_L2 : (metric == VSGPU_IP ? EPI_INT_IP : EPI_INT_COS);
    else if (type == VSGPU_SQ8 || type == VSGPU_SQ8H) {
        t->epi = l2 ? EPI_SQ8_L2 : EPI_SQ8_IP;
        t->sq8_centred = metric == VSGPU_IP && row_bytes == dim + 16;
    }
    else t->epi = l2 ? EPI_L2 : (t->prog.f16acc ? EPI_ONE_MINUS_H16 : EPI_ONE_MINUS);
    // SQ8 accumulates the code dot product in the IP order whatever the metric (L2 is algebraic: L2.cpp:30-45)
    if (type == VSGPU_SQ8 || type == VSGPU_SQ8H) t->opk = t->prog.fused ? OP_IP_FMA : OP_IP_MULADD;
    // LDS budget: offs + BT query images.  A CU has 160 KiB; up to 152 KiB go to one workgroup when a wide row needs them (the
    // default 64 KiB held fp32 rows to ~7 K elements and left rows beyond ~1.8 K with one query per pass)
    size_t offs_b = ((size_t)t->prog.steps * t->prog.vl * 4 + 15) & ~(size_t)15;
    size_t q_b = (size_t)t->prog.steps * t->prog.vl * acc_bytes(type);
    size_t budget = VSG_EXACT_LDS_BUDGET;
    // The reference takes any dim (spaces/L2_space.cpp:185-241, spaces.h:57-66): rows whose lane table + one query image do
    // not fit the LDS run on the global-table variant of the exact kernels, one query per pass (slow, correct).
    t->gtab = offs_b + q_b > budget;
    // as many queries per pass as fit, but an 8-query tile only while two workgroups still share a CU (80 KiB each)
    size_t fit = t->gtab ? 1 : (budget - offs_b) / q_b;
    t->bt_max = (fit >= 8 && offs_b + 8 * q_b <= 80 * 1024) ? 8 : (fit >= 4 ? 4 : 1);
    if (!t->prog.fused || t->opk == OP_IP_DPBF16) t->bt_max = 1;  // these orders are only instantiated for BT=1
    {
        // fp32 MFMA filter: any dim up to 3072.  The kernel instance is the next compiled width (k-steps of 32
        // elements); the columns past `dim` hold the start of the next row in LDS and zeros in the query fragments
        // (finite x 0 = 0; a NaN there only makes the filter pass the row on to the exact re-rank).
        // (widths 128 / 192 / 256 -- dims up to 4096 / 6144 / 8192 -- run on k_mfma_filter_wide, mfma_wide_kernels.hpp)
        static const int kInst[] = {4, 6, 8, 10, 12, 16, 20, 24, 28, 30, 32, 40, 48, 64, 80, 96, 128, 192, 256};
        size_t ks = 0;
        for (int v : kInst)
            if ((size_t)v * 32 >= dim) {
                ks = (size_t)v;
                break;
            }
        if (type == VSGPU_F64) {   // fp64 rows on the same filter (mfma_kernels.hpp EB = 8): fewer compiled widths
            static const int kInst64[] = {4, 8, 16, 24, 32, 48, 64};
            ks = 0;
            for (int v : kInst64)
                if ((size_t)v * 32 >= dim) {
                    ks = (size_t)v;
                    break;
                }
        }
        t->mfma_ok = ((type == VSGPU_F32 || type == VSGPU_F64) && !t->prog.scalar_tier &&
                      row_bytes == dim * (size_t)elem_bytes_of(type) && ks != 0);
        t->ksteps = (int)ks;
        const size_t data_bytes = dim * (size_t)elem_bytes_of(type);
        // low-precision rows: like fp32, any dim runs at the next compiled width (bf16/fp16: 256, 512, 768, 1024,
        // 1536 elements; int8/uint8: 512, 768, 1024) with zero query columns past `dim`
        // (the avx512_bf16 tier rides the same filter: E covers any accumulation order of the exact bf16 products, its
        // absolute term the flushed subnormals; survivors are re-scored in the vdpbf16ps order)
        if (!t->prog.scalar_tier && t->prog.reduce == 0 && (type == VSGPU_BF16 || type == VSGPU_F16) && dim <= 2048 &&
            row_bytes == data_bytes) {
            // (width 2048: 4 waves x 16 queries, the 64 fragments of a wave in AGPRs -- a 64-query tile)
            static const int w16[] = {8, 16, 24, 32, 48, 64};
            static const int rt16[] = {64, 32, 32, 16, 16, 16};
            for (int i = 0; i < 6; i++)
                if ((size_t)w16[i] * 32 >= dim) {
                    t->lowp_ok = true;
                    t->lp_kind = type == VSGPU_BF16 ? LP_BF16 : LP_F16;
                    t->lp_ksteps = w16[i];
                    t->lp_rt = rt16[i];
                    t->lp_qtile = w16[i] == 64 ? 64 : 128;
                    break;
                }
        }
        if (!t->prog.scalar_tier && t->prog.reduce == 0 && (type == VSGPU_BF16 || type == VSGPU_F16) && dim > 2048 && dim <= 8192 &&
            row_bytes == data_bytes) {
            // 16 queries per workgroup, the row's k range split over the four waves by ring stage (mfma_wide_kernels.hpp)
            static const int ww[] = {96, 128, 192, 256};
            for (int i = 0; i < 4; i++)
                if ((size_t)ww[i] * 32 >= dim) {
                    t->lowp_ok = true;
                    t->lp_wide = true;
                    t->lp_kind = type == VSGPU_BF16 ? LP_BF16 : LP_F16;
                    t->lp_ksteps = ww[i];
                    t->lp_rt = 16;
                    t->lp_qtile = 16;
                    break;
                }
        }
        if ((type == VSGPU_SQ8 || type == VSGPU_SQ8H) && !t->prog.scalar_tier && dim <= 1024) {
            // SQ8 x FP32 on the int8 MFMA filter (mfma_lowp_kernels.hpp LP_SQ8): 8 waves x 16 queries, 64-row tiles
            t->lowp_ok = true;
            t->lp_kind = LP_SQ8;
            t->lp_ksteps = dim <= 512 ? 8 : (dim <= 768 ? 12 : 16);
            t->lp_rt = 64;
            t->lp_qtile = 128;
            t->aux_bytes = 16;
        }
        if ((type == VSGPU_I8 || type == VSGPU_U8) && dim > 4096 && dim <= 16384) {
            // the k-split filter (mfma_wide_kernels.hpp, EK = 3 / 4 / 5): 16 (or 32) queries per workgroup, 1 KiB of the row per ring
            // stage and wave.  uint8 Cosine (EK = 5, round 5) needs two per-row values: the 16-byte aux records of the narrower widths.
            static const int ww[] = {96, 128, 192, 256};
            for (int i = 0; i < 4; i++)
                if ((size_t)ww[i] * 64 >= dim) {
                    t->lowp_ok = true;
                    t->lp_wide = true;
                    t->lp_kind = type == VSGPU_I8 ? LP_I8 : (metric == VSGPU_COSINE ? LP_U8C : LP_U8);
                    if (t->lp_kind == LP_U8C) t->aux_bytes = 16;
                    t->lp_ksteps = ww[i];
                    t->lp_rt = 16;
                    t->lp_qtile = 16;
                    break;
                }
        }
        if ((type == VSGPU_I8 || type == VSGPU_U8) && dim > 1024 && dim <= 4096) {
            // widths 2048 / 3072: 8 waves x 16 queries (128 / 192 registers of fragments per wave), 16-row tiles, a 128-query
            // tile; width 4096: 4 waves x 16 queries, the 64 fragments of a wave in AGPRs, a 64-query tile
            t->lowp_ok = true;
            t->lp_kind = type == VSGPU_I8 ? LP_I8 : (metric == VSGPU_COSINE ? LP_U8C : LP_U8);
            if (t->lp_kind == LP_U8C) t->aux_bytes = 16;
            // (round 5: width 1536 -- 24 k-steps on 32-row tiles of 512-byte stage segments, the shape 16-bit rows of 768 elements use;
            // before, 1025 .. 1536 elements ran at width 2048, a quarter of every row image the next row's bytes)
            t->lp_ksteps = dim <= 1536 ? 24 : (dim <= 2048 ? 32 : (dim <= 3072 ? 48 : 64));
            t->lp_rt = t->lp_ksteps == 24 ? 32 : 16;
            t->lp_qtile = dim <= 3072 ? 128 : 64;
            // (uint8 Cosine at width 3072: the 8-wave kernel -- 192 registers of fragments + the two-value epilogue in a wave's 256 --
            // spilled 60 registers and ran at 1.8 TB/s where int8 Cosine runs 4.9: the 4-wave shape of width 4096, fragments in AGPRs)
            if (t->lp_kind == LP_U8C && t->lp_ksteps == 48) t->lp_qtile = 64;
        }
        if ((type == VSGPU_I8 || type == VSGPU_U8) && dim <= 1024) {
            t->lowp_ok = true;
            // uint8 Cosine needs two aux values per row (sum x', the stored norm): the 16-byte aux records
            t->lp_kind = type == VSGPU_I8 ? LP_I8 : (metric == VSGPU_COSINE ? LP_U8C : LP_U8);
            if (t->lp_kind == LP_U8C) t->aux_bytes = 16;
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
    if (t->lowp_ok && t->lp_kind == LP_SQ8 &&
        (hipMalloc((void **)&t->d_sq8_max, 16) != hipSuccess || hipMemset(t->d_sq8_max, 0, 16) != hipSuccess)) {
        fail(VSGPU_ERR_HIP, "SQ8 aux allocation failed");
        delete t;
        return nullptr;
    }
    if (t->lowp_ok && (t->lp_kind == LP_I8 || t->lp_kind == LP_U8)) {
        // int8 / uint8 rows: {min, max} of the per-row aux value over every row ever stored (k_row_aux_i8; signed order of the bit
        // patterns: the sums are ints, the Cosine norms non-negative floats).  The 32 x 32 x 32 filter screens a unit of rows with
        // ONE integer threshold per query derived from them (mfma_i8x32_kernels.hpp, k_i8_filter_x32l).
        const int init[4] = {0x7FFFFFFF, (int)0x80000000, 0, 0};
        if (hipMalloc((void **)&t->d_sq8_max, 16) != hipSuccess || hipMemcpy(t->d_sq8_max, init, 16, hipMemcpyHostToDevice) != hipSuccess) {
            fail(VSGPU_ERR_HIP, "int8 aux allocation failed");
            delete t;
            return nullptr;
        }
    }
    return t;
}

// A view of `parent` for another context (reader lane): same rows, own stream + scratch.  The caller keeps writers away
// while a view is in use and calls vsgpu_table_view_sync after the parent changed.
extern "C" vsgpu_table *vsgpu_table_view_create(vsgpu_table *parent, vsgpu_ctx *ctx) {
    if (!parent || !ctx || parent->parent || ctx->device != parent->ctx->device) {
        fail(VSGPU_ERR_ARG, "bad view parameters");
        return nullptr;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    if (!parent->chain) {
        parent->chain = new ScanChain();
        if (hipEventCreateWithFlags(&parent->chain_ev, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&parent->scan_ev, hipEventDisableTiming) != hipSuccess)
            return nullptr;
    }
    vsgpu_table *v = new vsgpu_table(*parent);
    v->ctx = ctx;
    v->parent = parent;
    v->chain_ev = nullptr;
    v->scan_ev = nullptr;
    if (hipEventCreateWithFlags(&v->chain_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&v->scan_ev, hipEventDisableTiming) != hipSuccess) {
        delete v;
        return nullptr;
    }
    parent->chain->users++;
    return v;
}
extern "C" int vsgpu_table_view_sync(vsgpu_table *v) {
    if (!v || !v->parent) return fail(VSGPU_ERR_ARG, "not a view");
    const vsgpu_table *p = v->parent;
    v->slabs = p->slabs;
    v->d_slabs = p->d_slabs;
    v->d_slabs_cap = p->d_slabs_cap;
    v->norm_slabs = p->norm_slabs;
    v->d_norm_slabs = p->d_norm_slabs;
    v->d_sq8_max = p->d_sq8_max;
    v->n = p->n;
    for (int i = 0; i < 8; i++) v->sq8_blk[i] = p->sq8_blk[i];
    v->sq8_blk_set = p->sq8_blk_set;
    return VSGPU_OK;
}

extern "C" void vsgpu_table_destroy(vsgpu_table *t) {
    if (!t) return;
    (void)hipSetDevice(t->ctx->device);
    (void)hipStreamSynchronize(t->ctx->stream);
    if (t->chain_ev) (void)hipEventDestroy(t->chain_ev);
    if (t->scan_ev) (void)hipEventDestroy(t->scan_ev);
    if (t->chain && --t->chain->users == 0) delete t->chain;
    if (t->h_i8_ext) (void)hipHostFree(t->h_i8_ext);
    if (t->parent) {   // a view owns nothing else
        delete t;
        return;
    }
    for (char *s : t->slabs) (void)hipFree(s);
    for (float *s : t->norm_slabs) (void)hipFree(s);
    if (t->d_slabs) (void)hipFree(t->d_slabs);
    if (t->d_norm_slabs) (void)hipFree(t->d_norm_slabs);
    if (t->d_offs) (void)hipFree(t->d_offs);
    if (t->d_sq8_max) (void)hipFree(t->d_sq8_max);
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
        // + slack: the MFMA filters read a row out to their compiled kernel width (the next width at or above dim: up to
        // 511 floats more at the fp32 widths 64 / 80 / 96 k-steps), so the last row of a slab is over-read by less than
        // that width
        const size_t width = t->mfma_ok ? (size_t)t->ksteps * 32 * (size_t)elem_bytes_of(t->type) : (t->lowp_ok ? (size_t)t->lp_ksteps * 64 : 0);
        HIPCHK(hipMalloc((void **)&p, slab_rows * t->row_bytes + std::max<size_t>(1024, width + 256)));
        poison(p, slab_rows * t->row_bytes);
        t->slabs.push_back(p);
        if (t->mfma_ok || t->lowp_ok) {
            float *np = nullptr;
            HIPCHK(hipMalloc((void **)&np, slab_rows * t->aux_bytes));
            poison(np, slab_rows * t->aux_bytes);
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

// SQ8 tables hold code ^ 0x80 (= code - 128 as int8, the filter's MFMA operand; exact_kernels.hpp sq8_code): flip the codes of
// rows [first, first+n) that just arrived from the host.  The metadata behind the codes stays as it is.
static __global__ __launch_bounds__(256) void k_sq8_flip_codes(char *rows, uint32_t row_stride, uint32_t dim, uint32_t n) {
    const uint32_t per_row = (dim + 3) / 4;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < (uint64_t)n * per_row; i += (uint64_t)gridDim.x * 256) {
        const uint32_t row = (uint32_t)(i / per_row), e = (uint32_t)(i % per_row) * 4;
        unsigned char *p = reinterpret_cast<unsigned char *>(rows) + (size_t)row * row_stride + e;
        for (uint32_t j = 0; j < 4 && e + j < dim; j++) p[j] ^= 0x80u;
    }
}
static int sq8_flip_codes(vsgpu_table *t, size_t first, size_t n) {
    if ((t->type != VSGPU_SQ8 && t->type != VSGPU_SQ8H) || n == 0) return VSGPU_OK;
    const size_t slab_rows = (size_t)1 << t->slab_shift;
    size_t id = first, left = n;
    while (left) {
        const size_t in_slab = std::min(left, slab_rows - (id & (slab_rows - 1)));
        const uint64_t work = (uint64_t)in_slab * ((t->dim + 3) / 4);
        hipLaunchKernelGGL(k_sq8_flip_codes, dim3((unsigned)std::min<uint64_t>((work + 255) / 256, 65536)), dim3(256), 0, t->ctx->stream,
                           row_ptr(t, id), (uint32_t)t->row_bytes, (uint32_t)t->dim, (uint32_t)in_slab);
        id += in_slab;
        left -= in_slab;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(t->ctx->stream));   // (reader lanes query on their own streams)
    return VSGPU_OK;
}

// recompute |x|^2 of rows [first, first+n) (fp32 tables on the MFMA path only)
static int update_norms(vsgpu_table *t, size_t first, size_t n) {
    if (!(t->mfma_ok || t->lowp_ok) || n == 0) return VSGPU_OK;
    const size_t slab_rows = (size_t)1 << t->slab_shift;
    size_t id = first, left = n;
    while (left) {
        size_t in_slab = std::min(left, slab_rows - (id & (slab_rows - 1)));
        float *np = t->norm_slabs[id >> t->slab_shift] + (id & (slab_rows - 1)) * (t->aux_bytes / 4);
        const dim3 g((unsigned)((in_slab + 3) / 4));
        if (t->lowp_ok && t->lp_kind == LP_U8C)
            hipLaunchKernelGGL(k_row_aux_u8c, g, dim3(256), 0, t->ctx->stream, (const char *)row_ptr(t, id),
                               (uint32_t)t->row_bytes, (uint32_t)t->dim, (uint32_t)in_slab, (uint4 *)np);
        else if (t->lowp_ok && t->lp_kind == LP_SQ8)
            hipLaunchKernelGGL(k_row_aux_sq8, dim3((unsigned)((in_slab + 255) / 256)), dim3(256), 0, t->ctx->stream,
                               (const char *)row_ptr(t, id), (uint32_t)t->row_bytes, (uint32_t)t->dim, (uint32_t)in_slab,
                               t->epi == EPI_SQ8_L2 ? 1 : 0, (uint32_t)(id & (slab_rows - 1)),
                               (uint32_t *)t->norm_slabs[id >> t->slab_shift], t->d_sq8_max);
        else if (t->mfma_ok && t->type == VSGPU_F64)
            hipLaunchKernelGGL(k_row_norms_f64, g, dim3(256), 0, t->ctx->stream, (const char *)row_ptr(t, id),
                               (uint32_t)t->row_bytes, (uint32_t)t->dim, (uint32_t)in_slab, np);
        else if (t->mfma_ok)
            hipLaunchKernelGGL(k_row_norms_f32, g, dim3(256), 0, t->ctx->stream, (const char *)row_ptr(t, id),
                               (uint32_t)t->row_bytes, (uint32_t)t->dim, (uint32_t)in_slab, np);
        else if (t->lp_kind == LP_I8 || t->lp_kind == LP_U8)
            hipLaunchKernelGGL(k_row_aux_i8, g, dim3(256), 0, t->ctx->stream, (const char *)row_ptr(t, id),
                               (uint32_t)t->row_bytes, (uint32_t)t->dim, (uint32_t)in_slab,
                               t->lp_kind == LP_U8 ? (t->metric == VSGPU_L2 ? 2 : 3) : (t->metric == VSGPU_COSINE ? 1 : 0),
                               (uint32_t *)np, (int *)t->d_sq8_max);
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
    rc = sq8_flip_codes(t, t->n, n);
    if (rc) return rc;
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
    const int rc = sq8_flip_codes(t, id, 1);
    if (rc) return rc;
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
    if (t->type == VSGPU_SQ8 || t->type == VSGPU_SQ8H)   // (the device keeps code ^ 0x80: sq8_flip_codes)
        for (size_t i = 0; i < t->dim; i++) reinterpret_cast<unsigned char *>(host_row)[i] ^= 0x80u;
    return VSGPU_OK;
}
// rows [first, first + n) as stored, one D2H copy per slab segment (measurement / test hook: the CPU baseline of bench.py
// reads the synthetic rows back instead of regenerating them on the host)
extern "C" int vsgpu_table_read_range(vsgpu_table *t, size_t first, size_t n, void *host_rows) {
    if (first + n > t->n) return fail(VSGPU_ERR_ARG, "rows [%zu, %zu) out of range", first, first + n);
    HIPCHK(hipSetDevice(t->ctx->device));
    const size_t per_slab = (size_t)1 << t->slab_shift;
    char *dst = static_cast<char *>(host_rows);
    for (size_t r = first; r < first + n;) {
        const size_t run = std::min(first + n - r, per_slab - (r & (per_slab - 1)));
        HIPCHK(hipMemcpyAsync(dst, row_ptr(t, r), run * t->row_bytes, hipMemcpyDeviceToHost, t->ctx->stream));
        dst += run * t->row_bytes;
        r += run;
    }
    HIPCHK(hipStreamSynchronize(t->ctx->stream));
    if (t->type == VSGPU_SQ8 || t->type == VSGPU_SQ8H)   // (the device keeps code ^ 0x80: sq8_flip_codes)
        for (size_t i = 0; i < n; i++)
            for (size_t j = 0; j < t->dim; j++) static_cast<unsigned char *>(host_rows)[i * t->row_bytes + j] ^= 0x80u;
    return VSGPU_OK;
}
extern "C" int vsgpu_table_append_synthetic(vsgpu_table *t, size_t n, uint64_t seed) {
    if (t->type == VSGPU_F64 || t->type == VSGPU_U8 || t->type == VSGPU_SQ8 || t->type == VSGPU_SQ8H)
        return fail(VSGPU_ERR_UNSUPPORTED, "synthetic fill: fp32/bf16/fp16/int8 only");
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

// Build [nq][steps][vl] permuted/widened query images in pinned memory and upload them.
size_t staged_query_bytes(const vsgpu_table *t, size_t nq) { return nq * (size_t)t->prog.steps * t->prog.vl * acc_bytes(t->type); }
// host_dst / dev_dst given: the images are written to host_dst and ctx->qperm becomes dev_dst -- a region of a block the CALLER
// uploads (one copy per batch: vsgpu_mfma.hip); fp32 / fp64 / bf16 / fp16 tables only (no per-query metadata beside the images)
int stage_queries(vsgpu_table *t, const void *queries, size_t nq, size_t qstride, void *host_dst, void *dev_dst) {
    vsgpu_ctx *c = t->ctx;
    const LaneProgram &pg = t->prog;
    const size_t per_q = (size_t)pg.steps * pg.vl;
    const size_t ab = acc_bytes(t->type);
    const size_t bytes = nq * per_q * ab;
    int rc = VSGPU_OK;
    if (host_dst) {
        alias_into(c->qperm, dev_dst, bytes);
    } else {
        rc = ensure_pinned(c, bytes + nq * 8);
        if (rc) return rc;
        rc = ensure(c, c->qperm, bytes);
        if (rc) return rc;
    }
    char *dst = host_dst ? (char *)host_dst : (char *)c->pinned;
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
        case VSGPU_SQ8H: {   // fp16 query elements, widened exactly (types/float16.h:33-52)
            float *of = (float *)o;
            for (size_t i = 0; i < per_q; i++) {
                float v = 0;
                if (offs[i] >= 0) {
                    uint16_t h;
                    memcpy(&h, src + 2 * (size_t)offs[i], 2);
                    v = widen_f16(h);
                }
                of[i] = v;
            }
            break;
        }
        case VSGPU_SQ8: {   // table offsets address the one-byte codes: element e of the fp32 query sits at 4 e
            float *of = (float *)o;
            for (size_t i = 0; i < per_q; i++) {
                float v = 0;
                if (offs[i] >= 0) memcpy(&v, src + 4 * (size_t)offs[i], 4);
                of[i] = v;
            }
            break;
        }
        default: {
            int *oi = (int *)o;
            for (size_t i = 0; i < per_q; i++) oi[i] = offs[i] >= 0 ? (int)*(const uint8_t *)(src + offs[i]) : 0;
            break;
        }
        }
    }
    if (host_dst) return VSGPU_OK;
    HIPCHK(hipMemcpyAsync(c->qperm.p, c->pinned, bytes, hipMemcpyHostToDevice, c->stream));
    if (t->type == VSGPU_SQ8 || t->type == VSGPU_SQ8H) {   // {y_sum, y_sum_squares} of every query blob (the second only exists for L2)
        rc = ensure_pinned(c, bytes + nq * 8);
        if (rc) return rc;
        float *qm = (float *)((char *)c->pinned + bytes);
        for (size_t q = 0; q < nq; q++) {
            const char *src = (const char *)queries + q * qstride + (t->type == VSGPU_SQ8H ? 2 : 4) * t->dim;
            memcpy(&qm[2 * q], src, 4);
            qm[2 * q + 1] = 0.f;   // IP: the value the score is shifted by (y_mean_ip of a mean-centred table, else nothing)
            if (t->epi == EPI_SQ8_L2 || t->sq8_centred) memcpy(&qm[2 * q + 1], src + 4, 4);
        }
        rc = ensure(c, c->qnorm, nq * 8);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(c->qnorm.p, qm, nq * 8, hipMemcpyHostToDevice, c->stream));
    }
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
    if (lds == 0) {   // tables from global memory (vsgpu_table::gtab); callers pass one query per pass
        if constexpr (BT == 1) hipLaunchKernelGGL((k_exact_scan<EK, OPK, 1, true>), grid, dim3(256), 0, s, P);
        return;
    }
    if (lds > 64 * 1024)   // beyond the default dynamic-LDS limit: raised per kernel instance
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_exact_scan<EK, OPK, BT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
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
    case OP_IP_DPBF16:
        if constexpr (EK == EK_BF16) launch_scan_t<EK, OP_IP_DPBF16, 1>(P, grid, lds, s);
        break;
    case OP_L2_F16ACC:
        if constexpr (EK == EK_F16) launch_scan_bt<EK, OP_L2_F16ACC>(bt, P, grid, lds, s);
        break;
    case OP_IP_F16ACC:
        if constexpr (EK == EK_F16) launch_scan_bt<EK, OP_IP_F16ACC>(bt, P, grid, lds, s);
        break;
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
    case EK_SQ8:
        if (opk == OP_IP_FMA) launch_scan_bt<EK_SQ8, OP_IP_FMA>(bt, P, grid, lds, s);
        else launch_scan_t<EK_SQ8, OP_IP_MULADD, 1>(P, grid, lds, s);
        break;
    case EK_SQ8H:
        if (opk == OP_IP_FMA) launch_scan_bt<EK_SQ8H, OP_IP_FMA>(bt, P, grid, lds, s);
        else launch_scan_t<EK_SQ8H, OP_IP_MULADD, 1>(P, grid, lds, s);
        break;
    default: launch_scan_op<EK_U8>(opk, bt, P, grid, lds, s); break;
    }
}
int tile_rows_of(int ek) { return (ek == EK_F64 || ek == EK_BF16) ? (256 / 16) * 4 : (ek == EK_SQ8H ? (256 / 64) * 4 : (256 / 32) * 4); }

static int pick_bt(const vsgpu_table *t, size_t nq) {
    int bt = t->bt_max;
    while (bt > 1 && (size_t)bt / 2 >= nq) bt /= 2;  // 8 -> 4 -> ... while the smaller tile still covers nq
    if (bt == 2) bt = 4;
    if (bt > 1 && nq == 1) bt = 1;
    return bt;
}

// Fill the table/program part of ScanParams and launch over compact rows.
int run_scan(vsgpu_table *t, ScanParams &P, size_t nq, bool timed) {
    vsgpu_ctx *c = t->ctx;
    P.slabs = t->d_slabs;
    P.slab_shift = t->slab_shift;
    P.slab_mask = (uint32_t)(((size_t)1 << t->slab_shift) - 1);
    P.row_stride = (uint32_t)t->row_bytes;
    P.offs = t->d_offs;
    P.steps = t->prog.steps;
    P.reduce = t->prog.reduce;
    P.full_from = t->prog.full_from();
    P.full_to = t->prog.full_to();
    P.qperm = c->qperm.p;
    P.nq = (int)nq;
    P.epilogue = t->epi;
    P.norm_off = (uint32_t)t->dim;
    P.qnorm = (const float *)c->qnorm.p;
    P.sq8_fused = t->prog.fused ? 1 : 0;
    const int bt = t->gtab ? 1 : pick_bt(t, nq);
    const int tile_rows = tile_rows_of(t->ek);
    const uint32_t n_tiles = (P.n_compact + tile_rows - 1) / tile_rows;
    if (n_tiles == 0) return VSGPU_OK;
    const size_t offs_b = ((size_t)t->prog.steps * t->prog.vl * 4 + 15) & ~(size_t)15;
    const size_t lds = t->gtab ? 0 : offs_b + (size_t)bt * t->prog.steps * t->prog.vl * acc_bytes(t->type);
    const uint32_t q_tiles = (uint32_t)((nq + bt - 1) / bt);
    uint32_t gx = std::min<uint32_t>(n_tiles, (uint32_t)c->n_cu * 8);
    timed = timed && (c->opt_events & 1);
    if (timed) HIPCHK(hipEventRecord(c->ev_a, c->stream));
    launch_scan(t->ek, t->opk, bt, P, dim3(gx, q_tiles), lds, c->stream);
    HIPCHK(hipGetLastError());
    if (timed) HIPCHK(hipEventRecord(c->ev_b, c->stream));
    return VSGPU_OK;
}

void account_scan(vsgpu_ctx *c, vsgpu_table *t, uint64_t rows, uint64_t passes, const char *name) {
    float ms = 0;
    // (without the events -- option "events" bit 0 off -- launches, rows and the kernel's name are still counted; scan_ms stays put)
    if ((c->opt_events & 1) && hipEventElapsedTime(&ms, c->ev_a, c->ev_b) == hipSuccess) c->stats.scan_ms += ms;
    c->stats.scan_launches += 1;
    c->stats.scan_rows += rows;
    c->stats.scan_bytes += rows * t->row_bytes;
    (void)passes;
    snprintf(c->stats.scan_kernel, sizeof c->stats.scan_kernel, "%s", name);
}

// ------------------------------------------------------------------ dense scores
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
    // the ids travel through the pinned staging block, behind the query image stage_queries puts at its start: no copy from
    // pageable caller memory and no synchronisation before the kernel (the HNSW batch iterator's walk calls this once per group
    // of expansions, host/hnsw_iter.cpp: one round trip per call)
    const size_t qbytes = (staged_query_bytes(t, 1) + 8 + 15) & ~(size_t)15;
    int rc = ensure_pinned(c, qbytes + n * 4);
    if (rc) return rc;
    rc = stage_queries(t, query, 1, 0);
    if (rc) return rc;
    rc = ensure(c, c->ids, n * 4);
    if (rc) return rc;
    memcpy((char *)c->pinned + qbytes, ids, n * 4);
    HIPCHK(hipMemcpyAsync(c->ids.p, (char *)c->pinned + qbytes, n * 4, hipMemcpyHostToDevice, c->stream));
    return dense_to_host(t, 1, (const uint32_t *)c->ids.p, 0, n, scores);
}

// ------------------------------------------------------------------ SQ8 symmetric distances
// SQ8_SQ8_InnerProduct / _Cosine / _L2Sqr between two stored rows (IP.cpp:146-183, L2.cpp:185-201; AVX-512 VNNI tier from
// dim 64: IP_AVX512F_BW_VL_VNNI_SQ8_SQ8.h:38-65).  One wave per pair.  mode 0: VNNI tier (exact int32 dot, epilogue fused
// the way gcc fuses it, see oracle/vso_sq8.c); 1: scalar tier with dim < 64 (float accumulation of integers below 2^24 is
// exact, so the same int dot serves); 2: scalar tier beyond the 32-bit bound (dim > 33025): sequential float accumulation.
static __global__ __launch_bounds__(256) void k_sq8_pairs(const char *const *slabs, uint32_t slab_shift, uint32_t slab_mask,
                                                         uint32_t row_stride, uint32_t dim, int is_l2, int mode,
                                                         const uint32_t *ids_a, const uint32_t *ids_b, uint32_t n, float *out,
                                                         int centred, float mss) {
    const int lane = threadIdx.x & 63;
    const uint32_t pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= n) return;
    const uint32_t ra = ids_a[pair], rb = ids_b[pair];
    const unsigned char *a = reinterpret_cast<const unsigned char *>(slabs[ra >> slab_shift] + (size_t)(ra & slab_mask) * row_stride);
    const unsigned char *b = reinterpret_cast<const unsigned char *>(slabs[rb >> slab_shift] + (size_t)(rb & slab_mask) * row_stride);
    float fdot;
    if (mode == 2) {
        float product = 0.f;
        if (lane == 0)
            for (uint32_t i = 0; i < dim; i++) product = __fadd_rn(product, (float)((int)(a[i] ^ 0x80u) * (int)(b[i] ^ 0x80u)));   // (codes are stored ^ 0x80)
        fdot = product;
    } else {
        int dot = 0;
        for (uint32_t i = lane; i < dim; i += 64) dot += (int)(a[i] ^ 0x80u) * (int)(b[i] ^ 0x80u);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) dot += __shfl_xor(dot, o);
        fdot = (float)dot;
    }
    if (lane != 0) return;
    const char *ma = reinterpret_cast<const char *>(a) + dim, *mb = reinterpret_cast<const char *>(b) + dim;
    const float min1 = load_f32_unaligned(ma), delta1 = load_f32_unaligned(ma + 4), sum1 = load_f32_unaligned(ma + 8);
    const float min2 = load_f32_unaligned(mb), delta2 = load_f32_unaligned(mb + 4), sum2 = load_f32_unaligned(mb + 8);
    const float fdim = (float)dim;
    float ip;
    if (mode == 0) {
        const float A = __fmaf_rn(min1, sum2, __fmul_rn(min2, sum1));
        const float B = __fmaf_rn(__fmul_rn(delta1, delta2), fdot, A);
        ip = __fmaf_rn(-__fmul_rn(fdim, min1), min2, B);
    } else {
        const float t0 = __fmul_rn(min1, sum2), t1 = __fmul_rn(min2, sum1);
        const float t2 = __fmul_rn(__fmul_rn(fdim, min1), min2);
        const float t3 = __fmul_rn(__fmul_rn(delta1, delta2), fdot);
        ip = __fadd_rn(__fsub_rn(__fadd_rn(t0, t1), t2), t3);
    }
    float sc;
    if (!is_l2) sc = __fsub_rn(1.0f, ip);
    else sc = __fsub_rn(__fadd_rn(load_f32_unaligned(ma + 12), load_f32_unaligned(mb + 12)), __fmul_rn(2.0f, ip));
    // mean-centred IP rows (calculator.h:168-186): base - x_mean_ip - y_mean_ip + mean_sum_squares, left to right
    if (centred) sc = __fadd_rn(__fsub_rn(__fsub_rn(sc, load_f32_unaligned(ma + 12)), load_f32_unaligned(mb + 12)), mss);
    out[pair] = sc;
}

extern "C" int vsgpu_table_set_sq8_mean_sum_squares(vsgpu_table *t, float mean_sum_squares) {
    if (!t || (t->type != VSGPU_SQ8 && t->type != VSGPU_SQ8H)) return fail(VSGPU_ERR_ARG, "not an SQ8 table");
    t->sq8_mss = mean_sum_squares;
    return VSGPU_OK;
}

extern "C" int vsgpu_table_set_sq8_block_bounds(vsgpu_table *t, const float bounds[8]) {
    if (!t || (t->type != VSGPU_SQ8 && t->type != VSGPU_SQ8H)) return fail(VSGPU_ERR_ARG, "not an SQ8 table");
    for (int i = 0; i < 8; i++) t->sq8_blk[i] = bounds ? bounds[i] : 0.f;
    t->sq8_blk_set = bounds != nullptr;
    return VSGPU_OK;
}

extern "C" int vsgpu_sq8_pair_scores(vsgpu_table *t, const uint32_t *ids_a, const uint32_t *ids_b, size_t n, double *scores) {
    if (t->type != VSGPU_SQ8 && t->type != VSGPU_SQ8H) return fail(VSGPU_ERR_ARG, "not an SQ8 table");
    if (n == 0) return VSGPU_OK;
    for (size_t i = 0; i < n; i++)
        if (ids_a[i] >= t->n || ids_b[i] >= t->n) return fail(VSGPU_ERR_ARG, "row id beyond table size %zu", t->n);
    vsgpu_ctx *c = t->ctx;
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure(c, c->ids, 2 * n * 4);
    if (rc) return rc;
    rc = ensure(c, c->dense, n * 4);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->ids.p, ids_a, n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync((uint32_t *)c->ids.p + n, ids_b, n * 4, hipMemcpyHostToDevice, c->stream));
    const bool scalar = t->tier == VSGPU_TIER_SCALAR || t->dim < 64 || t->dim > 33025;   // L2_space.cpp:529-566
    const int mode = !scalar ? 0 : (t->dim > 33025 ? 2 : 1);
    hipLaunchKernelGGL(k_sq8_pairs, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, (const char *const *)t->d_slabs,
                       t->slab_shift, (uint32_t)(((size_t)1 << t->slab_shift) - 1), (uint32_t)t->row_bytes, (uint32_t)t->dim,
                       t->epi == EPI_SQ8_L2 ? 1 : 0, mode, (const uint32_t *)c->ids.p, (const uint32_t *)c->ids.p + n,
                       (uint32_t)n, (float *)c->dense.p, t->sq8_centred ? 1 : 0, t->sq8_mss);
    HIPCHK(hipGetLastError());
    std::vector<float> h(n);
    HIPCHK(hipMemcpyAsync(h.data(), c->dense.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; i++) scores[i] = (double)h[i];
    return VSGPU_OK;
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
// The fallback's form of it: a table-wide pass like any scan, so it is ordered behind the other reader lanes' scans through the
// scan chain (round-4 advisor finding: it ran beside them), and the score matrix it grew -- up to 1 GiB per context, every reader
// lane has one -- is released again instead of staying with the lane for the life of the index.
constexpr size_t DENSE_KEEP_BYTES = (size_t)1 << 28;   // what the small-problem path of vsgpu_topk uses routinely
static int fallback_dense_path(vsgpu_table *t, size_t nq, size_t k, size_t cap, uint32_t *ids, double *scores, uint32_t *counts,
                               size_t q_first, size_t q_count, const void *queries, size_t qstride) {
    vsgpu_ctx *c = t->ctx;
    ScanChainGuard g(t);
    g.before_scan();
    const bool was_plain = c->dense_plain;
    c->dense_plain = true;   // (the sliced dense path ends in collect_candidates, whose fallback is this function)
    const int rc = topk_dense_path(t, nq, k, cap, ids, scores, counts, q_first, q_count, queries, qstride);
    c->dense_plain = was_plain;
    (void)hipStreamSynchronize(c->stream);   // (the pass ends drained on success; an error may have left work queued)
    g.submitted();
    if (c->dense.p && !c->dense.alias && c->dense.cap > DENSE_KEEP_BYTES) {
        (void)hipFree(c->dense.p);
        c->dense = DevBuf{};
    }
    return rc;
}
// fp64 tables: the exact pair scores are doubles in c->dense ([nq][ccap], launch_exact_pairs), selection on 64-bit keys
static int collect_candidates_f64(vsgpu_table *t, const void *queries, size_t nq, size_t qstride, size_t k, size_t cap,
                                  size_t ccap, uint32_t *ids, double *scores, uint32_t *counts, const char *scan_name,
                                  ScanChainGuard *chain) {
    vsgpu_ctx *c = t->ctx;
    const size_t n = t->n, ocap = cap;
    // one block {selected counts [nq], raw counts [nq], records [nq][ocap]}: one download
    const size_t hdr = (nq * 8 + 15) & ~(size_t)15;
    int rc = ensure(c, c->sel, hdr + nq * ocap * sizeof(SelRec64));
    if (rc) return rc;
    hipLaunchKernelGGL(k_select_upto_kth_f64, dim3((unsigned)nq), dim3(256), 0, c->stream, (const uint2 *)c->cand.p,
                       (const double *)c->dense.p, (const uint32_t *)c->counts.p, (uint32_t)ccap, (uint32_t)std::min(k, n),
                       (SelRec64 *)((char *)c->sel.p + hdr), (uint32_t *)c->sel.p, (uint32_t)ocap);
    HIPCHK(hipGetLastError());
    if (chain) chain->submitted();
    rc = ensure_pinned(c, hdr + nq * ocap * sizeof(SelRec64));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->pinned, c->sel.p, hdr + nq * ocap * sizeof(SelRec64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    // (copied out of the pinned staging area: the dense fallback below reuses it)
    std::vector<uint32_t> hsel((const uint32_t *)c->pinned, (const uint32_t *)c->pinned + 2 * nq);
    std::vector<SelRec64> hrec((const SelRec64 *)((const char *)c->pinned + hdr), (const SelRec64 *)((const char *)c->pinned + hdr) + nq * ocap);
    {
        account_scan(c, t, n, 1, scan_name);
        float ms = 0;
        if ((c->opt_events & 2) && hipEventElapsedTime(&ms, c->ev_c, c->ev_d) == hipSuccess) c->stats.other_ms += ms;
    }
    const uint32_t *hraw = hsel.data() + nq;
    std::vector<Hit> hits;
    for (size_t q = 0; q < nq; q++) {
        if (hraw[q] > ccap || hraw[q] < std::min(k, n)) {   // more candidates than slots: exact dense fallback
            c->stats.fallbacks++;
            rc = fallback_dense_path(t, nq, k, cap, ids, scores, counts, q, 1, queries, qstride);
            if (rc) return rc;
            continue;
        }
        c->stats.candidates += hraw[q];
        if (hsel[q] == VSGPU_COUNT_OVERFLOW) {
            counts[q] = VSGPU_COUNT_OVERFLOW;
            continue;
        }
        hits.resize(hsel[q]);
        for (size_t i = 0; i < hsel[q]; i++) {
            double d;
            memcpy(&d, &hrec[q * ocap + i].bits, 8);
            hits[i] = Hit{(uint32_t)hrec[q * ocap + i].row, d};
        }
        std::sort(hits.begin(), hits.end(), [](const Hit &a, const Hit &b) { return a.id < b.id; });
        emit(hits, q, cap, ids, scores, counts);
    }
    return VSGPU_OK;
}

int collect_candidates(vsgpu_table *t, const void *queries, size_t nq, size_t qstride, size_t k, size_t cap,
                              size_t ccap, uint32_t *ids, double *scores, uint32_t *counts, const char *scan_name,
                              ScanChainGuard *chain) {
    if (t->type == VSGPU_F64) return collect_candidates_f64(t, queries, nq, qstride, k, cap, ccap, ids, scores, counts, scan_name, chain);
    vsgpu_ctx *c = t->ctx;
    const size_t n = t->n;
    // GPU: keep, per query, the candidates with exact score <= T_k; only those travel to the host
    const size_t ocap = cap;
    // one block {selected counts [nq], raw counts [nq], records [nq][ocap]}: one download
    const size_t hdr = (nq * 8 + 15) & ~(size_t)15;
    // (round 6) the select kernel writes its block straight into the pinned reply block -- host memory the device sees at the same
    // address -- so the batch has no download operation (a blit kernel + its queue hop: 9 us of every batch); only the records a
    // query kept cross PCIe.  Option sel_mapped = 0: device block + hipMemcpyAsync as before.
    const bool mapped = c->opt_sel_mapped != 0;
    int rc = mapped ? ensure_pinned(c, hdr + nq * ocap * sizeof(uint2)) : ensure(c, c->sel, hdr + nq * ocap * sizeof(uint2));
    if (rc) return rc;
    char *selblk = mapped ? (char *)c->pinned : (char *)c->sel.p;
    hipLaunchKernelGGL(k_select_upto_kth, dim3((unsigned)nq), dim3(256), 0, c->stream, (const uint2 *)c->cand.p,
                       (const uint32_t *)c->counts.p, (uint32_t)ccap, (uint32_t)std::min(k, n), (uint2 *)(selblk + hdr),
                       (uint32_t *)selblk, (uint32_t)ocap);
    HIPCHK(hipGetLastError());
    // the last kernel of this batch is in the stream: the next reader lane's kernels may follow (its probe and scan then
    // overlap with this lane's downloads and host replay, not with its kernels -- a re-rank or select kernel sharing the
    // CUs with another lane's scan cost that scan more than the overlap saved: bf16 config 4, 3.18 -> 3.33 ms)
    if (chain) chain->submitted();
    rc = ensure_pinned(c, hdr + nq * ocap * sizeof(uint2));
    if (rc) return rc;
    uint32_t *hsel = (uint32_t *)c->pinned;
    uint2 *hrec = (uint2 *)((char *)c->pinned + hdr);
    if (!mapped) HIPCHK(hipMemcpyAsync(c->pinned, c->sel.p, hdr + nq * ocap * sizeof(uint2), hipMemcpyDeviceToHost, c->stream));
    WallMarks wm;
    HIPCHK(hipStreamSynchronize(c->stream));
    wm.mark("wait_gpu");
    {
        account_scan(c, t, n, 1, scan_name);
        float ms = 0;
        if ((c->opt_events & 2) && hipEventElapsedTime(&ms, c->ev_c, c->ev_d) == hipSuccess) c->stats.other_ms += ms;
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
                size_t oob = 0;
                for (uint32_t j = 0; j < cn; j++) oob += hc[q * ccap + j].x >= n;
                fprintf(stderr,
                        "VSGPU_VERIFY MISS %s q=%zu row=%zu exact=%.9g T_k=%.9g tau=%.9g cand_slot=%ld stored=%.9g count=%u "
                        "raw=%u sel=%u oob_rows=%zu n=%zu\n",
                        scan_name, q, i, row[i], T, (double)htau[q], at, (double)stored, hcnt[q], hraw[q], hsel[q], oob, n);
            }
        }
    }
    std::vector<Hit> hits;
    std::vector<size_t> retry_q, dense_q;
    std::vector<float> retry_tau;
    for (size_t q = 0; q < nq; q++) {
        // (only behind an MFMA filter: `chain` is theirs.  The exact-kernel filter path of a table without one -- tiny dims, the scalar
        // tier -- has no second filter to run and goes to the dense pass below; round 3 sent it into topk_mfma, a fault on such a
        // table: found by the round-4 soak, fp16 dim 13, 400 K rows, k 37)
        if (chain != nullptr && hraw[q] > ccap && !c->in_retry && qstride != 0 && hsel[q] != VSGPU_COUNT_OVERFLOW && hsel[q] >= std::min(k, n)) {
            // More candidates than slots (near-duplicate clusters: the probe's threshold sits inside the cluster).  The slots
            // that were filled hold real rows with exact scores, so the k-th smallest of THEM -- the largest score the selection
            // kept -- bounds the true k-th score from above and is far tighter than the probe's: one more filter pass over all
            // such queries of the batch with those thresholds, instead of a dense exact pass per query (10 M x 768: ~4.5 ms
            // per batch against ~125 ms per query, tools/bench_overflow.py).
            float tmax = -INFINITY;
            for (size_t i = 0; i < hsel[q]; i++) {
                float f;
                memcpy(&f, &hrec[q * ocap + i].y, 4);
                tmax = std::max(tmax, f);
            }
            if (std::isfinite(tmax)) {
                retry_q.push_back(q);
                retry_tau.push_back(tmax);
                continue;
            }
        }
        if (hraw[q] > ccap || hraw[q] < std::min(k, n)) {
            // still more candidates than slots (massive ties) or a short list: exact dense fallback, below, runs of neighbouring
            // queries together
            c->stats.fallbacks++;
            dense_q.push_back(q);
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
    for (size_t i = 0; i < dense_q.size();) {
        size_t j = i + 1;
        while (j < dense_q.size() && dense_q[j] == dense_q[j - 1] + 1 && qstride != 0) j++;
        rc = fallback_dense_path(t, nq, k, cap, ids, scores, counts, dense_q[i], j - i, queries, qstride);
        if (rc) return rc;
        i = j;
    }
    if (!retry_q.empty()) {
        const size_t m = retry_q.size();
        std::vector<char> sub(m * qstride);
        for (size_t j = 0; j < m; j++) memcpy(sub.data() + j * qstride, (const char *)queries + retry_q[j] * qstride, qstride);
        std::vector<uint32_t> sid(m * cap), scnt(m);
        std::vector<double> ssc(m * cap);
        c->tau_override = retry_tau.data();
        c->in_retry = true;
        rc = t->lowp_ok ? topk_lowp(t, sub.data(), m, qstride, k, cap, sid.data(), ssc.data(), scnt.data())
                        : topk_mfma(t, sub.data(), m, qstride, k, cap, sid.data(), ssc.data(), scnt.data());
        c->tau_override = nullptr;
        c->in_retry = false;
        if (rc) return rc;
        c->stats.retries += m;
        for (size_t j = 0; j < m; j++) {
            const size_t q = retry_q[j];
            counts[q] = scnt[j];
            if (scnt[j] == VSGPU_COUNT_OVERFLOW) continue;
            memcpy(ids + q * cap, sid.data() + j * cap, (size_t)scnt[j] * 4);
            memcpy(scores + q * cap, ssc.data() + j * cap, (size_t)scnt[j] * 8);
        }
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
// Probe tiles can come in runs of 2^shift consecutive tiles (MfmaParams::tile_run_shift, option "probe_run").  Measured
// with ~2 MiB runs on the three BASELINE shapes: no faster than one tile every tile_step tiles (the probe is not bound by
// address translation), and evenly spread tiles sample a table that was ingested in clusters better: default 0.
uint32_t probe_run_shift(const vsgpu_ctx *c, size_t tile_bytes, uint32_t probe_tiles) {
    (void)tile_bytes;
    uint32_t s = c->opt_probe_run > 0 ? (uint32_t)c->opt_probe_run : 0;
    while (s > 0 && (probe_tiles >> s) < 64) s--;
    return s;
}
uint32_t probe_divisor(const vsgpu_ctx *c, size_t n, size_t nq, size_t k, bool rerank) {
    if (c->opt_probe_div > 0) return (uint32_t)c->opt_probe_div;
    if (!rerank) return 48;
    const double d = std::sqrt(0.15 * (double)n / (double)(std::max<size_t>(nq, 1) * std::max<size_t>(k, 1)));
    return (uint32_t)std::min(64.0, std::max(8.0, d));
}

size_t candidate_capacity(const vsgpu_ctx *c, size_t k, size_t n, size_t probe_rows) {
    double expect = (double)k * (double)n / (double)std::max<size_t>(probe_rows, 1);
    size_t want = (size_t)std::min(expect * 3.0 + 64.0, 1048576.0);
    want = std::max<size_t>((size_t)c->opt_cand_cap, want);
    // the retry pass of overflowed queries (collect_candidates): a handful of queries, 32 x the room -- a cluster of near-duplicates
    // that the bf16 bound cannot tell apart passes the filter whatever the threshold, and is then settled by the exact re-rank
    if (c->in_retry) want = std::min<size_t>(want * 32, std::max<size_t>(n, 64));
    return want;
}


// (Round 4 tried the selection in this kernel's tail -- the last workgroup of a query to finish, by ticket, selects -- to save a
// launch.  Correct (568 GPU tests) and 0.8-1.2 ms SLOWER per batch of 64-128 queries: every workgroup needs a device-scope
// release before its ticket, which on this multi-XCD part writes back its L2 (buffer_wbl2 sc1); a kernel boundary pays that once.)
int launch_exact_pairs(vsgpu_table *t, size_t nq, size_t ccap) {
    vsgpu_ctx *c = t->ctx;
    ScanParams S{};
    S.slabs = t->d_slabs;
    S.slab_shift = t->slab_shift;
    S.slab_mask = (uint32_t)(((size_t)1 << t->slab_shift) - 1);
    S.row_stride = (uint32_t)t->row_bytes;
    S.offs = t->d_offs;
    S.steps = t->prog.steps;
    S.reduce = t->prog.reduce;
    S.qperm = c->qperm.p;
    S.nq = (int)nq;
    S.epilogue = t->epi;
    S.counts = (uint32_t *)c->counts.p;
    S.cand = (uint2 *)c->cand.p;
    S.cap = (uint32_t)ccap;
    dim3 grid(64, (unsigned)nq);
    const bool l2 = (t->opk == OP_L2_FMA);
    S.norm_off = (uint32_t)t->dim;
    S.qnorm = (const float *)c->qnorm.p;
    S.sq8_fused = t->prog.fused ? 1 : 0;
    if (t->type == VSGPU_F64) {   // double scores beside the candidate list (topk_mfma sized c->dense for them)
        S.out = c->dense.p;
        if (l2) hipLaunchKernelGGL((k_exact_pairs<EK_F64, OP_L2_FMA>), grid, dim3(256), 0, c->stream, S);
        else hipLaunchKernelGGL((k_exact_pairs<EK_F64, OP_IP_FMA>), grid, dim3(256), 0, c->stream, S);
    } else if (t->type == VSGPU_SQ8H) {
        hipLaunchKernelGGL((k_exact_pairs<EK_SQ8H, OP_IP_FMA>), grid, dim3(256), 0, c->stream, S);
    } else if (t->type == VSGPU_SQ8) {
        hipLaunchKernelGGL((k_exact_pairs<EK_SQ8, OP_IP_FMA>), grid, dim3(256), 0, c->stream, S);
    } else if (t->type == VSGPU_F32) {
        if (l2) hipLaunchKernelGGL((k_exact_pairs<EK_F32, OP_L2_FMA>), grid, dim3(256), 0, c->stream, S);
        else hipLaunchKernelGGL((k_exact_pairs<EK_F32, OP_IP_FMA>), grid, dim3(256), 0, c->stream, S);
    } else if (t->type == VSGPU_BF16) {
        if (l2) hipLaunchKernelGGL((k_exact_pairs<EK_BF16, OP_L2_FMA>), grid, dim3(256), 0, c->stream, S);
        else if (t->opk == OP_IP_DPBF16) hipLaunchKernelGGL((k_exact_pairs<EK_BF16, OP_IP_DPBF16>), grid, dim3(256), 0, c->stream, S);
        else hipLaunchKernelGGL((k_exact_pairs<EK_BF16, OP_IP_FMA>), grid, dim3(256), 0, c->stream, S);
    } else if (t->prog.f16acc) {   // AVX512-FP16 tier: half-precision accumulators
        if (l2) hipLaunchKernelGGL((k_exact_pairs<EK_F16, OP_L2_F16ACC>), grid, dim3(256), 0, c->stream, S);
        else hipLaunchKernelGGL((k_exact_pairs<EK_F16, OP_IP_F16ACC>), grid, dim3(256), 0, c->stream, S);
    } else {
        if (l2) hipLaunchKernelGGL((k_exact_pairs<EK_F16, OP_L2_FMA>), grid, dim3(256), 0, c->stream, S);
        else hipLaunchKernelGGL((k_exact_pairs<EK_F16, OP_IP_FMA>), grid, dim3(256), 0, c->stream, S);
    }
    HIPCHK(hipGetLastError());
    return VSGPU_OK;
}
int launch_probe_threshold(vsgpu_ctx *c, size_t nq, uint32_t probe_tiles, size_t k, uint32_t M, bool seed_list) {
    if (seed_list) {   // the streaming filter's lists (M <= 2048 there: the sorting kernel)
        hipLaunchKernelGGL(k_probe_threshold, dim3((unsigned)nq), dim3(1024), M * sizeof(float), c->stream,
                           (const float *)c->dense.p, (size_t)probe_tiles, probe_tiles, (uint32_t)k, M, (float *)c->tau.p,
                           (uint32_t *)c->klist.p, (uint32_t)c->opt_stream_stride);
        HIPCHK(hipGetLastError());
        return VSGPU_OK;
    }
    if (M > 2048)
        hipLaunchKernelGGL(k_probe_threshold_wide, dim3((unsigned)nq), dim3(1024), 0, c->stream,
                           (const float *)c->dense.p, (size_t)probe_tiles, probe_tiles, (uint32_t)k, M, (float *)c->tau.p);
    else
        hipLaunchKernelGGL(k_probe_threshold, dim3((unsigned)nq), dim3(1024), M * sizeof(float), c->stream,
                           (const float *)c->dense.p, (size_t)probe_tiles, probe_tiles, (uint32_t)k, M, (float *)c->tau.p);
    HIPCHK(hipGetLastError());
    return VSGPU_OK;
}

// ------------------------------------------------------------------ top-K
// Exact scores of ALL rows for nq queries in one dense matrix on the device, the k-th smallest per query by radix selection there
// (k_select_dense_upto_kth), only the rows at or below it to the host.  The small-problem path of vsgpu_topk and the fallback of
// queries whose candidate lists overflowed twice (collect_candidates).  fp32-scored tables; the caller bounds nq * n.
static bool dense_sliced_ok(const vsgpu_table *t, size_t nq) {
    const vsgpu_ctx *c = t->ctx;
    return !c->dense_plain && c->opt_dense_sliced_bytes > 0 && nq <= (size_t)c->opt_dense_small_q && t->n >= 8192 &&
           (t->type == VSGPU_F32 || t->type == VSGPU_BF16 || t->type == VSGPU_F16 || t->type == VSGPU_I8 || t->type == VSGPU_U8);
}
// The dense path with its selection dealt over slices of the rows (k_select_dense_slices): ONE staged block (zeroed candidate counters +
// the exact-order query images) uploaded by the copy kernel, the exact scan into the dense score matrix, the slice select appending
// each slice's rows at or below its own k-th score to the query's candidate list, and collect_candidates' final select + reply.
static int dense_sliced_topk(vsgpu_table *t, const void *queries, size_t nq, size_t qstride, size_t k, size_t cap, uint32_t *ids,
                             double *scores, uint32_t *counts) {
    vsgpu_ctx *c = t->ctx;
    const size_t n = t->n, kk = std::min(k, n);
    const size_t S = std::min<size_t>(256, (n + 4095) / 4096), L = (n + S - 1) / S;
    const size_t ccap = S * (2 * kk + 64);
    // the block: {zeroed candidate counters, query norms (int8 / uint8 Cosine: the float behind a query's elements), exact-order images}
    const size_t ab = (nq * 4 + 255) & ~(size_t)255, qb = (staged_query_bytes(t, nq) + 255) & ~(size_t)255;
    int rc = ensure(c, c->qblock, 2 * ab + qb);
    if (rc) return rc;
    rc = ensure_pin_up(c, 2 * ab + qb);
    if (rc) return rc;
    memset(c->pin_up, 0, 2 * ab);
    if (t->epi == EPI_INT_COS)
        for (size_t q = 0; q < nq; q++) memcpy((char *)c->pin_up + ab + 4 * q, (const char *)queries + q * qstride + t->dim, 4);
    rc = stage_queries(t, queries, nq, qstride, (char *)c->pin_up + 2 * ab, (char *)c->qblock.p + 2 * ab);
    if (rc) return rc;
    alias_into(c->counts, c->qblock.p, ab);
    alias_into(c->qnorm, (char *)c->qblock.p + ab, ab);
    rc = upload_block(c, c->qblock.p, c->pin_up, 2 * ab + qb);
    if (rc) return rc;
    rc = ensure(c, c->dense, nq * n * 4);
    if (rc) return rc;
    rc = ensure(c, c->cand, nq * ccap * sizeof(uint2));
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
    rc = run_scan(t, P, nq, true);
    if (rc) return rc;
    hipLaunchKernelGGL(k_select_dense_slices, dim3((unsigned)nq, (unsigned)S), dim3(1024), 0, c->stream, (const float *)c->dense.p, n,
                       (uint32_t)n, (uint32_t)kk, (uint2 *)c->cand.p, (uint32_t *)c->counts.p, (uint32_t)ccap, (uint32_t)L);
    HIPCHK(hipGetLastError());
    // final select among the slices' survivors, records into the pinned reply block, host emit; a list that ran over its room (massive
    // exact ties) or came up short (NaN scores) goes to the plain dense pass (collect_candidates -> fallback_dense_path)
    return collect_candidates(t, queries, nq, qstride, k, cap, ccap, ids, scores, counts, "k_exact_scan(dense)", nullptr);
}
static int dense_gpu_topk(vsgpu_table *t, const void *queries, size_t nq, size_t qstride, size_t k, size_t cap, uint32_t *ids,
                          double *scores, uint32_t *counts) {
    vsgpu_ctx *c = t->ctx;
    const size_t n = t->n;
    // (a k beyond a quarter of a 4 K-row slice would have every slice hand on most of its rows: the one-workgroup select then)
    if (dense_sliced_ok(t, nq) && std::min(k, n) <= 1024) return dense_sliced_topk(t, queries, nq, qstride, k, cap, ids, scores, counts);
    int rc = stage_queries(t, queries, nq, qstride);
    if (rc) return rc;
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
    rc = ensure(c, c->selcnt, nq * 8);   // (selected counts, then the raw counts the kernel passes on)
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

static int topk_dense_path(vsgpu_table *t, size_t nq, size_t k, size_t cap, uint32_t *ids, double *scores,
                           uint32_t *counts, size_t q_first, size_t q_count, const void *queries, size_t qstride) {
    // queries [q_first, q_first+q_count) answered from full score vectors
    const size_t n = t->n;
    if (t->type != VSGPU_F64) {
        // on the device, a group of queries per pass over the rows (the dense matrix stays below 1 GiB): rounds 1-3 brought every
        // query's n scores to the host one query at a time -- 74 ms per query at 10 M x 768, tools/bench_overflow.py
        const size_t group = std::max<size_t>(1, std::min<size_t>(q_count, ((size_t)1 << 30) / (n * 4)));
        for (size_t q0 = q_first; q0 < q_first + q_count; q0 += group) {
            const size_t g = std::min(group, q_first + q_count - q0);
            int rc = dense_gpu_topk(t, (const char *)queries + q0 * qstride, g, qstride, k, cap, ids + q0 * cap, scores + q0 * cap, counts + q0);
            if (rc) return rc;
        }
        return VSGPU_OK;
    }
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

    // small problems (and fp64 without the MFMA filter: narrow batches, scalar-tier dims): one dense score matrix
    const bool f64_filter = f64 && t->mfma_ok && c->opt_mfma && nq >= (size_t)c->opt_mfma_min_q;
    const bool small_sliced = c->opt_dense_pairs > 0 /* (dense_pairs = 0 is how callers ask for the filter path) */ && dense_sliced_ok(t, nq) && (double)n * (double)t->row_bytes * (double)nq <= (double)c->opt_dense_sliced_bytes;
    if ((f64 && !f64_filter) || (double)n * (double)nq <= (double)c->opt_dense_pairs || n <= 4 * k || small_sliced) {
        if (!f64 && n * nq * 4 <= ((size_t)1 << 28)) return dense_gpu_topk(t, queries, nq, qstride, k, cap, ids, scores, counts);
        if (f64) {
            // fp64: dense double scores per group of queries, 64-bit selection on the GPU, survivors to the host
            const size_t ocap = cap;
            const size_t group = std::max<size_t>(1, std::min<size_t>(nq, ((size_t)1 << 30) / (n * 8)));
            int rc = ensure(c, c->dense, group * n * 8);
            if (rc) return rc;
            rc = ensure(c, c->sel, group * ocap * sizeof(SelRec64));
            if (rc) return rc;
            rc = ensure(c, c->selcnt, group * 8);
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

    if (c->opt_events & 2) HIPCHK(hipEventRecord(c->ev_c, c->stream));
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
    if (c->opt_events & 2) HIPCHK(hipEventRecord(c->ev_d, c->stream));
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
