// vsgpu_internal.hpp -- shared by the translation units of libvsgpu.so (vsgpu.hip, vsgpu_mfma.hip,
// vsgpu_lowp.hip, vsgpu_hnsw.hip, vsgpu_comm.hip): context / table objects and the host helpers around
// the kernels.  Not part of any ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "vsgpu.h"
#include "lane_program.h"
#include "exact_kernels.hpp"

int vsg_fail(int code, const char *fmt, ...);
#define fail vsg_fail
#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return fail(_e == hipErrorOutOfMemory ? VSGPU_ERR_OOM : VSGPU_ERR_HIP, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(_e), __FILE__, __LINE__);                        \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool alias = false;   // p points into another buffer (vsgpu_ctx::qblock): never freed through this handle
    // the buffer's OWN allocation while p is an alias (alias_into parks it here, ensure() takes it back): a context whose calls alternate
    // between a path that aliases (one staged block per batch) and one that does not neither frees and reallocates per call nor --
    // round 6, found by the sliced dense path's test -- keeps writing through a stale alias into a block the next path lays out anew
    void *own_p = nullptr;
    size_t own_cap = 0;
};

struct vsgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_d = nullptr;
    DevBuf qperm, qnorm, dense, tau, counts, cand, ids, qfrag, qfrag2, qn2, sel, selcnt, qmeta, klist;
    // one upload per batch: {query fragments, |q|^2, thresholds, zeroed counters} are regions of qblock, staged in pin_up
    DevBuf qblock;
    void *pin_up = nullptr;
    size_t pin_up_cap = 0;
    void *pinned = nullptr;
    size_t pinned_cap = 0;
    vsgpu_stats stats{};
    // options
    long opt_mfma = 1;
    long opt_mfma_variant = 0;
    long opt_lowp_variant = 0;
    long opt_hnsw_slots = 16;  // resident search waves (= visited-tag slots) per CU: 8 -> 264 K QPS, 12-32 -> 314-319 K (200 K x 768)
    long opt_wide_blocks = 0;  // k_mfma_filter_wide: 0 = as many 16-query column blocks per workgroup as the registers hold (4 at width 96 k-steps, 2 up to 192), 1 = always one, 2 = at most two
    long opt_wide_gx = 0;      // k_mfma_filter_wide: workgroups per query tile (diagnosis); 0 = every query tile of a row tile resident at once
    long opt_sq8_block = 1;    // (rounds 1-2: the SQ8 filter's block pre-screen; accepted, without effect since round 3)
    // timing events in the batch's stream (each costs the GPU's timeline 3-5 us, profiles/r04_event_cost.txt): bit 0 = around the
    // scan kernel (stats.scan_ms: what bench.py's roofline reads), bit 1 = around probe + threshold (stats.other_ms)
    long opt_events = 1;
    long opt_probe_rt16 = 1;   // fp32 / fp64 probe on 16-row tiles where the filter uses them
    // round 6 (the seven stream operations of a batch): 1 = the batch's query block reaches the device through a copy KERNEL reading the
    // pinned staging block (a kernel behind a kernel starts 0 us later; a kernel behind an SDMA copy 8 us later, profiles/r06_timelines.txt),
    // 0 = hipMemcpyAsync.  Blocks beyond 4 MiB always take hipMemcpyAsync.
    long opt_upload_kernel = 1;
    // 1 = the select kernel writes the batch's records straight into the pinned (device-visible) reply block: no download operation
    long opt_sel_mapped = 1;
    // (round 6) the streaming threshold of the fp32 / fp64 filter (mfma_kernels.hpp MF_STREAM): a probe of opt_stream_probe_tiles tiles
    // instead of n / div rows, the filter tightens tau while it streams.  0 = the full probe + fixed thresholds
    long opt_stream_tau = 0;
    long opt_stream_stride = 128;  // words between two queries' lists
    long opt_stream_probe_tiles = 512;
    long opt_stream_early = 0;     // ... and after each of a workgroup's first this many tiles (the threshold falls fastest at the start)
    long opt_stream_refresh = 8;   // the filter re-reads a query's threshold every this many tiles (a device-scope load each: behind the XCD's L2)
    long opt_chain_early = 1;  // reader lanes: the next lane's probe may follow this lane's SCAN (1; its scan still waits for the select kernel) or only the select kernel (0)
    long opt_lowp_narrow = 1;  // batches of <= 64 queries on 4-wave workgroups (SQ8, bf16 / fp16 up to 768 elements)
    long opt_lowp_qsplit = 0;  // int8: 1 = two 128-query workgroups per row tile instead of one 256-query one
    long opt_lowp_x32 = 1; // int8/uint8 1 KiB rows, batches wider than 128: the 32x32x32 filters (mfma_i8x32_kernels.hpp).  1 = by the table's
                           // aux spread (lean stream k_i8_filter_x32l when one integer threshold per query screens well, else the per-value
                           // screen of k_i8_filter_x32); 0 = the 16x16x64 filter; other values = VAR + 1 of a kernel variant (tuning)
                               // kernel, value - 1 = VAR bits (tuning build; the shipped build has 32769 only)
    long opt_lowp_ksplit = 0;  // int8/uint8 1 KiB rows: K-split filter kernel (mfma_i8ks_kernels.hpp); 2 = with s_setprio
    long opt_lowp_dbg = 0;   // diagnosis only: bit0 skip epilogue, bit1 skip LDS reads + MFMA, bit2 skip row DMA
    long opt_wg_per_cu = 2;
    // the bf16 / fp16 / int8 / SQ8 filters hold ONE workgroup per CU (query fragments fill the register file): a grid of one
    // workgroup per CU avoids the tail a second round of workgroups leaves (bf16 config-4 shape: 6.3 against 6.0 TB/s,
    // profiles/r01_tuning_lowp.txt round-2 block)
    long opt_lowp_wg_per_cu = 1;
    long opt_mfma_min_q = 1;          // batches narrower than this stay on the exact kernel.  Measured (tools/bench_small_batches.py):
                                      // the MFMA filter wins from one query up (10M x 768: 4.7 ms vs 5.8-8.5 ms for 1-8 queries)
    long opt_dense_pairs = 1L << 16;  // nq*n at or below this: one dense score matrix + one select kernel
    // (round 6) fp32 / bf16 / fp16 tables, at most opt_dense_small_q queries: the dense path while rows x storedDataSize x queries stays
    // at or below THIS many bytes, its selection dealt over slices of the rows (k_select_dense_slices) -- upload, exact scan, slice
    // select, final select: four stream operations instead of the filter path's six, for single queries on small tables (BASELINE
    // config 1: 100 K x 128 fp32, one query: 51 -> 39 us; 400 K rows 80 -> 71; 800 K rows: the filter path wins, profiles/r06_c1_dense.txt).  0 = off
    long opt_dense_sliced_bytes = 256L << 20;
    long opt_dense_small_q = 4;
    bool dense_plain = false;         // inside the overflow fallback: the one-workgroup select, no slices (it must not come back here)
    long opt_probe_div = 0;           // probe ~ n / probe_div rows; 0 = chosen per call by probe_divisor()
    long opt_probe_cap = 32768;       // ... but at most this many probe tiles
    long opt_probe_run = -1;          // probe tiles per contiguous run, as a shift; -1 = about 2 MiB per run (probe_run_shift())
    long opt_cand_cap = 8192;         // candidate slots per query
    int n_cu = 256;
    // second filter pass of queries whose candidate list overflowed (collect_candidates): thresholds handed over instead of probed
    const float *tau_override = nullptr;
    bool in_retry = false;
    int (*poll)(void *) = nullptr;   // vsgpu_set_poll
    void *poll_user = nullptr;
};
// between the launches of a top-k call: drain the stream and ask the caller's poll function (no-op without one)
#define VSG_POLL_POINT(c)                                                          \
    do {                                                                           \
        if ((c)->poll) {                                                           \
            HIPCHK(hipStreamSynchronize((c)->stream));                             \
            if ((c)->poll((c)->poll_user)) return VSGPU_ERR_TIMEOUT;               \
        }                                                                          \
    } while (0)

int poison_byte();
void poison(void *p, size_t bytes);
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
int ensure(vsgpu_ctx *c, DevBuf &b, size_t bytes);
int ensure_pinned(vsgpu_ctx *c, size_t bytes);

// Reader lanes (vsgpu_table_view_create): several contexts -- each with its own stream and scratch -- query one set of
// rows.  The big scan kernels of the lanes are chained on the GPU in submission order (each waits for the event the previous
// one recorded), so they run one after the other at full bandwidth while a lane's small kernels, copies and host work
// overlap with another lane's scan.
struct ScanChain {
    // two gates, always taken in this order.  mu_scan / last_scan: from a lane's probe to its scan kernel -- the next lane's PROBE
    // waits for last_scan (recorded behind the scan).  mu_batch / last_batch: from a lane's scan to its select kernel -- the next
    // lane's SCAN waits for last_batch (recorded behind the select kernel).  So a lane's probe runs beside the other lane's re-rank
    // and select (small grids), but nothing ever runs beside a scan: the scans deal their tiles statically over the workgroups
    // resident at launch, and a kernel still holding CUs when a scan starts leaves late workgroups that finish late
    // (SQ8 10 M x 768, batch 128, two readers: scan 1.74 -> 2.67 ms when the other lane's re-rank overlapped its start).
    std::mutex mu_scan, mu_batch;
    hipEvent_t last_scan = nullptr, last_batch = nullptr;   // (owned by the recording lane's table)
    int users = 1;
};

struct vsgpu_table {
    vsgpu_ctx *ctx = nullptr;
    vsgpu_table *parent = nullptr;   // non-null: a view (shares the parent's slabs, lane table and aux arrays)
    ScanChain *chain = nullptr;
    hipEvent_t chain_ev = nullptr, scan_ev = nullptr;
    int type = 0, metric = 0, tier = 0;
    size_t dim = 0, row_bytes = 0;
    vsg::LaneProgram prog;
    int32_t *d_offs = nullptr;
    std::vector<char *> slabs;
    char **d_slabs = nullptr;
    size_t d_slabs_cap = 0;
    uint32_t slab_shift = 0;
    size_t n = 0;
    int ek = 0, opk = 0, epi = 0;
    int bt_max = 1;  // largest query tile whose LDS image fits
    bool gtab = false;   // lane table + one query image exceed the LDS budget: k_exact_scan<..., GT = true> reads both from global memory
    // MFMA filter path (fp32, AVX-512-order tier, dim a multiple of 64): |x|^2 per row, slab-parallel
    bool mfma_ok = false;
    int ksteps = 0;
    // low-precision MFMA filter (bf16/fp16/int8 rows): kernel shape picked at create time
    bool lowp_ok = false;
    int lp_kind = 0, lp_ksteps = 0, lp_rt = 0, lp_qtile = 0;
    bool lp_wide = false;   // bf16 / fp16 rows of 2049 .. 8192 elements: k_mfma_filter_wide<.., EK = 1 | 2> (16 queries per workgroup)
    bool sq8_centred = false;   // mean-centred IP rows (dim + 16 bytes: x_mean_ip behind the three base slots), queries carry y_mean_ip
    float sq8_mss = 0.f;        // sum mean_i^2, the symmetric IP correction constant
    float sq8_blk[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // extremes of the rows' metadata for the filter's block pre-screen (vsgpu.h)
    bool sq8_blk_set = false;
    size_t aux_bytes = 4;   // per-row aux of the MFMA filters: 4 B, or 16 B for SQ8 rows (k_row_aux_sq8: four arrays per 64 rows)
    int *h_i8_ext = nullptr;         // int8 / uint8 tables: pinned host copy of {min, max} of the aux values (d_sq8_max), refreshed behind a
    size_t i8_ext_n = (size_t)-1;    // batch whenever rows were added since (a heuristic input only: which filter kernel to launch)
    uint32_t *d_sq8_max = nullptr;   // SQ8: {max delta, max |min|, max sum_squares} over the rows ever stored (k_row_aux_sq8)
    std::vector<float *> norm_slabs;
    float **d_norm_slabs = nullptr;
};

static inline size_t acc_bytes(int type) { return type == VSGPU_F64 ? 8 : 4; }
constexpr size_t VSG_EXACT_LDS_BUDGET = 152 * 1024;   // dynamic LDS of k_exact_scan: lane table + query images

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
static inline uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
// around the launch of a table-wide scan kernel (between the timing events): orders it behind the other lanes' scans
// (round 4, late: putting probe / threshold / scan of all lanes into ONE stream while several batches are in flight -- no
// cross-queue event between one lane's scan and the next lane's probe -- measured slower on every config, +20 us per batch on
// config 2, +8 on config 1: tools/tuning_tests/chain_shared.patch, profiles/r04_chain_shared.txt)
struct ScanChainGuard {
    vsgpu_table *t;
    bool early;                      // option chain_early: the two gates apart (else both from the start to the select kernel)
    bool held_scan = false, held_batch = false;
    explicit ScanChainGuard(vsgpu_table *tt) : t(tt), early(tt->ctx->opt_chain_early != 0) {
        if (!t->chain) return;
        t->chain->mu_scan.lock();
        held_scan = true;
        if (t->chain->last_scan && t->chain->last_scan != t->scan_ev) (void)hipStreamWaitEvent(t->ctx->stream, t->chain->last_scan, 0);
        if (!early) before_scan();
    }
    void before_scan() {   // in front of the scan kernel's launch
        if (!t->chain || held_batch) return;
        t->chain->mu_batch.lock();
        held_batch = true;
        if (t->chain->last_batch && t->chain->last_batch != t->chain_ev) (void)hipStreamWaitEvent(t->ctx->stream, t->chain->last_batch, 0);
    }
    void scan_submitted() {   // the scan is in the stream
        if (!held_scan) return;
        (void)hipEventRecord(t->scan_ev, t->ctx->stream);
        t->chain->last_scan = t->scan_ev;
        t->chain->mu_scan.unlock();
        held_scan = false;
    }
    void scan_submitted_if_early() {
        if (early) scan_submitted();
    }
    void submitted() {   // the batch's last kernel is in the stream
        if (!t->chain) return;
        scan_submitted();
        if (!held_batch) return;
        (void)hipEventRecord(t->chain_ev, t->ctx->stream);
        t->chain->last_batch = t->chain_ev;
        t->chain->mu_batch.unlock();
        held_batch = false;
    }
    ~ScanChainGuard() {
        if (held_scan) t->chain->mu_scan.unlock();
        if (held_batch) t->chain->mu_batch.unlock();
    }
};
int stage_queries(vsgpu_table *t, const void *queries, size_t nq, size_t qstride, void *host_dst = nullptr, void *dev_dst = nullptr);
size_t staged_query_bytes(const vsgpu_table *t, size_t nq);
int ensure_pin_up(vsgpu_ctx *c, size_t bytes);
// the batch's staged block (pinned host memory, device-visible) -> device, on the batch's stream: a copy kernel or hipMemcpyAsync
int upload_block(vsgpu_ctx *c, void *dev, const void *pinned_src, size_t bytes);
void alias_into(DevBuf &b, void *p, size_t bytes);
int tile_rows_of(int ek);
int run_scan(vsgpu_table *t, vsg::ScanParams &P, size_t nq, bool timed);
void account_scan(vsgpu_ctx *c, vsgpu_table *t, uint64_t rows, uint64_t passes, const char *name);
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
int collect_candidates(vsgpu_table *t, const void *queries, size_t nq, size_t qstride, size_t k, size_t cap,
                       size_t ccap, uint32_t *ids, double *scores, uint32_t *counts, const char *scan_name,
                       ScanChainGuard *chain = nullptr);
uint32_t probe_run_shift(const vsgpu_ctx *c, size_t tile_bytes, uint32_t probe_tiles);
uint32_t probe_divisor(const vsgpu_ctx *c, size_t n, size_t nq, size_t k, bool rerank);
size_t candidate_capacity(const vsgpu_ctx *c, size_t k, size_t n, size_t probe_rows);
// stage 2 of the filter paths: reference-order exact re-score of the candidate lists in ctx->cand (in place)
int launch_exact_pairs(vsgpu_table *t, size_t nq, size_t ccap);
// threshold of every query from the probe's per-tile minima (ctx->dense -> ctx->tau)
int launch_probe_threshold(vsgpu_ctx *c, size_t nq, uint32_t probe_tiles, size_t k, uint32_t M, bool seed_list = false);
int topk_mfma(vsgpu_table *t, const void *queries, size_t nq, size_t qstride, size_t k, size_t cap, uint32_t *ids,
              double *scores, uint32_t *counts);
int topk_lowp(vsgpu_table *t, const void *queries, size_t nq, size_t qstride, size_t k, size_t cap, uint32_t *ids,
              double *scores, uint32_t *counts);
