// vsgpu_mfma.hip -- fp32 MFMA filter path of vsgpu_topk (kernels: mfma_kernels.hpp)
#include "vsgpu_internal.hpp"
#include "mfma_kernels.hpp"
#include "mfma_wide_kernels.hpp"

using namespace vsg;

// Default shapes: the probe always uses 64-row tiles;
// the filter uses 16-row x 1-KiB stages with non-temporal DMA when dim % 256 == 0 and the tile is at
// least as long as the ring (dim >= 512), else 64-row x 256-B stages.  Measured on 10M x 768 (profiles/):
// 64-row default policy 5.5 ms, 64-row nt 5.16 ms, 16-row nt 4.99 ms.
template <int KS, int EB = 4> static uint32_t launch_filter_ks(MfmaParams Q, size_t n, uint32_t wgs, unsigned q_tiles, hipStream_t s) {
    if constexpr (KS % 8 == 0 && KS >= 16) {
        Q.n_tiles = (uint32_t)((n + 15) / 16);
        hipLaunchKernelGGL((k_mfma_filter<KS, MF_FILTER, 3, 2, 1, 16, 0, EB>), dim3(std::min(Q.n_tiles, wgs), q_tiles), dim3(256),
                           mf_lds_bytes(3), s, Q);
    } else {
        Q.n_tiles = (uint32_t)((n + 63) / 64);
        hipLaunchKernelGGL((k_mfma_filter<KS, MF_FILTER, 3, 2, 1, 64, 0, EB>), dim3(std::min(Q.n_tiles, wgs), q_tiles), dim3(256),
                           mf_lds_bytes(3), s, Q);
    }
    return Q.n_tiles;
}
// rows per probe tile: the filter's own 16-row x 1-KiB stages where the width allows them (round 4: the 64-row probe streamed its
// 640 MB at 4.3 TB/s where the filter reaches 7; the probe sits between two scans in full, profiles/r04_c2_timeline.txt)
static inline bool probe_rt16(const vsgpu_ctx *c, int ksteps) { return ksteps % 8 == 0 && ksteps >= 16 && c->opt_probe_rt16 != 0; }
template <int KS, int EB = 4> static void launch_probe_ks(const MfmaParams &P, dim3 grid, hipStream_t s, bool rt16) {
    if constexpr (KS % 8 == 0 && KS >= 16) {
        if (rt16) {
            hipLaunchKernelGGL((k_mfma_filter<KS, MF_PROBE, 3, 2, 1, 16, 0, EB>), grid, dim3(256), mf_probe_lds_bytes(3), s, P);
            return;
        }
    }
    hipLaunchKernelGGL((k_mfma_filter<KS, MF_PROBE, 3, 0, 1, 64, 0, EB>), grid, dim3(256), mf_probe_lds_bytes(3), s, P);
}
// fp64 rows: the widths vsgpu_table_create picks from for VSGPU_F64 (k-steps of 32 doubles)
static void launch_filter_f64(int ksteps, const MfmaParams &P, size_t n, uint32_t wgs, unsigned q_tiles, hipStream_t s) {
    switch (ksteps) {
    case 4: launch_filter_ks<4, 8>(P, n, wgs, q_tiles, s); break;
    case 8: launch_filter_ks<8, 8>(P, n, wgs, q_tiles, s); break;
    case 16: launch_filter_ks<16, 8>(P, n, wgs, q_tiles, s); break;
    case 24: launch_filter_ks<24, 8>(P, n, wgs, q_tiles, s); break;
    case 32: launch_filter_ks<32, 8>(P, n, wgs, q_tiles, s); break;
    case 48: launch_filter_ks<48, 8>(P, n, wgs, q_tiles, s); break;
    default: launch_filter_ks<64, 8>(P, n, wgs, q_tiles, s); break;
    }
}
static void launch_probe_f64(int ksteps, const MfmaParams &P, dim3 grid, hipStream_t s, bool rt16) {
    switch (ksteps) {
    case 4: launch_probe_ks<4, 8>(P, grid, s, rt16); break;
    case 8: launch_probe_ks<8, 8>(P, grid, s, rt16); break;
    case 16: launch_probe_ks<16, 8>(P, grid, s, rt16); break;
    case 24: launch_probe_ks<24, 8>(P, grid, s, rt16); break;
    case 32: launch_probe_ks<32, 8>(P, grid, s, rt16); break;
    case 48: launch_probe_ks<48, 8>(P, grid, s, rt16); break;
    default: launch_probe_ks<64, 8>(P, grid, s, rt16); break;
    }
}
#ifdef VSGPU_TUNING
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
#else
static int launch_mfma_variant(int, MfmaParams, size_t, uint32_t, unsigned, hipStream_t) { return 0; }
#endif
// the streaming-threshold filter (MF_STREAM): the 16-row x 1-KiB shapes only
template <int KS, int EB = 4> static void launch_stream_ks(MfmaParams Q, size_t n, uint32_t wgs, unsigned q_tiles, hipStream_t s) {
    Q.n_tiles = (uint32_t)((n + 15) / 16);
    hipLaunchKernelGGL((k_mfma_filter<KS, MF_STREAM, 3, 2, 1, 16, 0, EB>), dim3(std::min(Q.n_tiles, wgs), q_tiles), dim3(256), mf_lds_bytes(3) + 1024 /* the waves' threshold mailboxes */, s, Q);
}
static bool stream_shape(int ksteps) { return ksteps == 16 || ksteps == 24 || ksteps == 32 || ksteps == 48 || ksteps == 64 || ksteps == 96; }
static void launch_stream(int ksteps, const MfmaParams &P, size_t n, uint32_t wgs, unsigned q_tiles, hipStream_t s) {
    switch (ksteps) {
    case 16: launch_stream_ks<16>(P, n, wgs, q_tiles, s); break;
    case 24: launch_stream_ks<24>(P, n, wgs, q_tiles, s); break;
    case 32: launch_stream_ks<32>(P, n, wgs, q_tiles, s); break;
    case 48: launch_stream_ks<48>(P, n, wgs, q_tiles, s); break;
    case 64: launch_stream_ks<64>(P, n, wgs, q_tiles, s); break;
    default: launch_stream_ks<96>(P, n, wgs, q_tiles, s); break;
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
static void launch_probe(int ksteps, const MfmaParams &P, dim3 grid, hipStream_t s, bool rt16) {
    switch (ksteps) {
    case 4: launch_probe_ks<4>(P, grid, s, rt16); break;
    case 6: launch_probe_ks<6>(P, grid, s, rt16); break;
    case 8: launch_probe_ks<8>(P, grid, s, rt16); break;
    case 10: launch_probe_ks<10>(P, grid, s, rt16); break;
    case 12: launch_probe_ks<12>(P, grid, s, rt16); break;
    case 16: launch_probe_ks<16>(P, grid, s, rt16); break;
    case 20: launch_probe_ks<20>(P, grid, s, rt16); break;
    case 24: launch_probe_ks<24>(P, grid, s, rt16); break;
    case 28: launch_probe_ks<28>(P, grid, s, rt16); break;
    case 30: launch_probe_ks<30>(P, grid, s, rt16); break;
    case 40: launch_probe_ks<40>(P, grid, s, rt16); break;
    case 48: launch_probe_ks<48>(P, grid, s, rt16); break;
    case 80: launch_probe_ks<80>(P, grid, s, rt16); break;
    case 64: launch_probe_ks<64>(P, grid, s, rt16); break;
    case 96: launch_probe_ks<96>(P, grid, s, rt16); break;
    default: launch_probe_ks<32>(P, grid, s, rt16); break;
    }
}

// rows wider than 3072 elements: 16 queries per workgroup, the k range split over the waves (mfma_wide_kernels.hpp)
template <int KS, int MODE, int NQ, int NS> static void launch_wide_ks(const MfmaParams &P, dim3 grid, hipStream_t s) {
    constexpr int lds_bytes = mfw_lds_bytes(MODE == MF_PROBE, NQ, NS);
    static_assert(lds_bytes <= 160 * 1024, "LDS ring does not fit");
#ifdef VSGPU_TUNING
    if (getenv("VSGPU_WIDE_NT")) {
        auto k2 = k_mfma_filter_wide<KS, MODE, 2, 0, NQ, NS>;
        if (lds_bytes > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        hipLaunchKernelGGL(k2, grid, dim3(256), lds_bytes, s, P);
        return;
    }
#endif
    auto kern = k_mfma_filter_wide<KS, MODE, 0, 0, NQ, NS>;
    if (lds_bytes > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, s, P);
}
// nq_blocks: 16-query column blocks per workgroup (2 up to width 6144: the fragments of 32 queries fit the registers of a wave).
// WIDE_NS_ALONE: ring slots of a workgroup that is alone on its CU (two blocks, or width 8192): see k_mfma_filter_wide
// (profiles/r05_wide_ring_depth.txt; two stages per ring barrier, while the loop still had one, measured no faster: _depth2.txt)
#ifndef WIDE_NS_ALONE
#define WIDE_NS_ALONE 5
#endif
template <int MODE> static void launch_wide(int ksteps, int nq_blocks, const MfmaParams &P, dim3 grid, hipStream_t s) {
    if (nq_blocks == 8) {   // 128 queries per workgroup at widths 512 / 768 / 1024 (ring depth: at most one row tile ahead)
        switch (ksteps) {
        case 16: launch_wide_ks<16, MODE, 8, 3>(P, grid, s); break;
        case 24: launch_wide_ks<24, MODE, 8, 4>(P, grid, s); break;
        default: launch_wide_ks<32, MODE, 8, 5>(P, grid, s); break;
        }
        return;
    }
    switch (ksteps) {
    case 128:
        if (nq_blocks == 3) launch_wide_ks<128, MODE, 3, WIDE_NS_ALONE>(P, grid, s);
        else if (nq_blocks == 2) launch_wide_ks<128, MODE, 2, WIDE_NS_ALONE>(P, grid, s);
        else launch_wide_ks<128, MODE, 1, 3>(P, grid, s);
        break;
    case 192:
        if (nq_blocks == 2) launch_wide_ks<192, MODE, 2, WIDE_NS_ALONE>(P, grid, s);
        else launch_wide_ks<192, MODE, 1, 3>(P, grid, s);
        break;
    default: launch_wide_ks<256, MODE, 1, WIDE_NS_ALONE>(P, grid, s); break;
    }
}

int topk_mfma(vsgpu_table *t, const void *queries, size_t nq, size_t qstride, size_t k, size_t cap,
                     uint32_t *ids, double *scores, uint32_t *counts) {
    vsgpu_ctx *c = t->ctx;
    const size_t n = t->n, dim = t->dim;
    const int KS = t->ksteps;
    // (round 5, late) fp32 rows of 257 .. 1024 elements (kernel widths 512 / 768 / 1024), MORE THAN 64 QUERIES: k_mfma_filter holds 64 queries per workgroup, so a batch of 128
    // crosses the rows twice (3.4 TB/s effective at any dim).  The k-split filter holds EIGHT column blocks at these widths -- 8 x KS registers
    // of fragments per wave, 128-256 (at width 1536 the 384 + two blocks' epilogue spill) -- i.e. 128 queries per pass.  Option wide_blocks = 8 / 0 allow it, any other value keeps the 64-query tiles.
    const bool k8 = t->type == VSGPU_F32 && (KS == 16 || KS == 24 || KS == 32) && nq > 64 && (c->opt_wide_blocks == 0 || c->opt_wide_blocks == 8);
    const bool wide = KS > 96 || k8;   // (fp32 only: vsgpu_table_create offers fp64 rows no width beyond 64)
    // (three column blocks at width 128 k-steps -- 4096 elements -- where 48-query tiles cross the rows less often than 32-query tiles: vsgpu_lowp.hip)
    const bool three = wide && KS == 128 && (c->opt_wide_blocks == 0 || c->opt_wide_blocks == 3) && (nq + 47) / 48 < (nq + 31) / 32;
    const int wide_blocks = k8 ? 8 : three ? 3 : ((wide && KS <= 192 && nq > (size_t)MFW_QTILE && c->opt_wide_blocks != 1) ? 2 : 1);
    const bool rt16 = !wide && probe_rt16(c, KS);
    const size_t QT = wide ? (size_t)MFW_QTILE * wide_blocks : (size_t)MF_QTILE, TILE_ROWS = (wide || rt16) ? 16 : (size_t)MF_TILE_ROWS;
    // (padded queries: whole 64-query tiles, and whole 48-query tiles where three column blocks run: 128 queries are three of those = 144)
    const size_t q_tiles = (nq + QT - 1) / QT, nqp = std::max((nq + MF_QTILE - 1) / MF_QTILE * MF_QTILE, q_tiles * QT);
    const bool l2 = (t->metric == VSGPU_L2);

    // (1) bf16 B-operand fragments + |q|^2 for the filter, (2) exact-order query images for the re-rank -- staged behind the
    // filter launch, under the scan, since only the re-rank reads them
    int rc = VSGPU_OK;
    WallMarks wm0;
    const size_t kdim = (size_t)KS * 32;       // kernel width >= dim
    // ONE upload per batch: {fragments [q_tile][wave][kstep][lane][8], |q|^2, thresholds, zeroed candidate counters} are built in a
    // pinned staging block and land in regions of ctx->qblock (three copies and a fill were four operations on the stream -- 4-7 us
    // of blit kernel each -- in front of every batch; a single query's whole GPU time is 31 us)
    // (round 4: the exact-order query images of the re-rank ride in the same block -- a second upload cost a single query 5-8 us)
    const size_t fb = (nqp * kdim * 2 + 255) & ~(size_t)255, ab = (nqp * 4 + 255) & ~(size_t)255;
    const size_t qb = (staged_query_bytes(t, nq) + 255) & ~(size_t)255;
    rc = ensure(c, c->qblock, fb + 3 * ab + qb);
    if (rc) return rc;
    rc = ensure_pin_up(c, fb + 3 * ab + qb);
    if (rc) return rc;
    uint16_t *frag = reinterpret_cast<uint16_t *>(c->pin_up);
    float *qn2 = reinterpret_cast<float *>((char *)c->pin_up + fb), *tau0 = reinterpret_cast<float *>((char *)c->pin_up + fb + ab);
    memset(c->pin_up, 0, fb + ab);
    for (size_t q = 0; q < nqp; q++) tau0[q] = -INFINITY;
    memset((char *)c->pin_up + fb + 2 * ab, 0, ab);
    const bool f64 = t->type == VSGPU_F64;
    std::vector<float> narrow(f64 ? dim : 0);
    for (size_t q = 0; q < nq; q++) {
        const float *src = (const float *)((const char *)queries + q * qstride);
        double ss = 0;
        if (f64) {   // the filter sees the query through float, then bf16; the re-rank sees the doubles (stage_queries)
            const double *d = (const double *)((const char *)queries + q * qstride);
            for (size_t i = 0; i < dim; i++) {
                ss += d[i] * d[i];
                narrow[i] = (float)d[i];
            }
            src = narrow.data();
        } else {
            for (size_t i = 0; i < dim; i++) ss += (double)src[i] * (double)src[i];
        }
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
    wm0.mark("frag_build");
    alias_into(c->qfrag, c->qblock.p, fb);
    alias_into(c->qn2, (char *)c->qblock.p + fb, ab);
    alias_into(c->tau, (char *)c->qblock.p + fb + ab, ab);
    alias_into(c->counts, (char *)c->qblock.p + fb + 2 * ab, ab);
    const uint32_t total_tiles = (uint32_t)((n + TILE_ROWS - 1) / TILE_ROWS);
    uint32_t probe_tiles = std::max<uint32_t>(total_tiles / probe_divisor(c, n, nq, k, true), (uint32_t)(4 * k * MF_TILE_ROWS / TILE_ROWS));   // (at least 256 k rows)
    probe_tiles = std::min<uint32_t>(std::min<uint32_t>(probe_tiles, total_tiles), (uint32_t)(c->opt_probe_cap * (long)(MF_TILE_ROWS / TILE_ROWS)));   // (the cap counts 64-row tiles)
    // (round 6) streaming threshold: a SMALL probe seeds tau and the per-query lists, the filter tightens them while it streams
    // (mfma_kernels.hpp MF_STREAM).  fp32 rows on the 16-row shapes, first passes only (a retry pass brings its thresholds), k <= MF_KLIST
    const bool stream_kernel_only = c->opt_stream_tau == 4 && !wide && rt16 && t->type == VSGPU_F32 && stream_shape(KS);
    const bool stream = c->opt_stream_tau != 0 && c->opt_stream_tau != 4 && !wide && rt16 && t->type == VSGPU_F32 && stream_shape(KS) && c->tau_override == nullptr && k <= (size_t)MF_KLIST &&
                        (size_t)c->opt_stream_probe_tiles * 4 < probe_tiles;
    const uint32_t full_probe_tiles = probe_tiles;
    if (stream) probe_tiles = std::max<uint32_t>((uint32_t)c->opt_stream_probe_tiles, (uint32_t)(4 * k * MF_TILE_ROWS / TILE_ROWS));
    // room per query: the streaming filter lets through ~2.3 k ln(n / probe rows) rows (the harmonic sum of a tightening k-th) -- 6 x that
    // -- and never less than the full probe's pass is given (clustered rows: many lower bounds within E of the k-th score)
    const size_t ccap = stream ? std::max<size_t>(candidate_capacity(c, k, n, (size_t)full_probe_tiles * TILE_ROWS),
                                                  (size_t)(14.0 * (double)k * std::log((double)n / ((double)probe_tiles * TILE_ROWS)) + 256.0))
                               : candidate_capacity(c, k, n, (size_t)probe_tiles * TILE_ROWS);
    rc = ensure(c, c->cand, nqp * ccap * sizeof(uint2));
    if (rc) return rc;
    const bool have_tau = c->tau_override != nullptr;   // (retry pass: thresholds from the first pass's exact scores, no probe)
    if (have_tau)
        for (size_t q = 0; q < nq; q++) tau0[q] = c->tau_override[q];
    rc = stage_queries(t, queries, nq, qstride, (char *)c->pin_up + fb + 3 * ab, (char *)c->qblock.p + fb + 3 * ab);
    if (rc) return rc;
    wm0.mark("stage_queries");
    rc = upload_block(c, c->qblock.p, c->pin_up, fb + 3 * ab + qb);
    if (rc) return rc;
    wm0.mark("uploads");

    // rigorous |a - s_ref| <= cE*(|x|^2+|q|^2) + absE   (derivation: DESIGN.md §5.2)
    const double u = std::ldexp(1.0, -24);
    const double cq = std::ldexp(1.0, -8) * (1.0 + std::ldexp(1.0, -10)) + (double)kdim * std::ldexp(1.0, -22) * 1.01;
    const double gref = ((double)kdim / 32.0 + 12.0) * u;
    const float cE = (float)(((cq + 2.0 * gref) * 1.001 + 16.0 * u) * (1.0 + 1e-6));
    const float absE = l2 ? 1e-30f : 1e-6f;

    uint32_t tile_step = total_tiles / probe_tiles, stream_shift = 0;
    if (stream) {   // a power-of-two stride: the filter tells the probe's tiles by a mask (the sample then spans more than half the table)
        while ((2u << stream_shift) <= tile_step) stream_shift++;
        tile_step = 1u << stream_shift;
    }
    uint32_t M = 64;  // group minima sorted per query (more probe tiles than that are grouped, see topk_lowp)
    while (M < probe_tiles && M < 8192 && M < 64 * k) M <<= 1;   // (64 k groups: two of the k best rows rarely share one)

    // (fp64: the exact pair scores -- doubles, [nq][ccap] -- reuse this buffer once the thresholds are out)
    rc = ensure(c, c->dense, std::max(nqp * (size_t)probe_tiles * 4, f64 ? nqp * ccap * 8 : (size_t)0));
    if (rc) return rc;

    if (stream) {
        rc = ensure(c, c->klist, nqp * (size_t)c->opt_stream_stride * sizeof(uint32_t));
        if (rc) return rc;
    }
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

    // behind the other reader lanes' probe + scan (no-op for a table without views): the table-wide kernels of all lanes run
    // one after the other, so each streams at full bandwidth and its HIP-event time is its own; what overlaps with another
    // lane's scan is this lane's query upload before, and its re-rank, selection, download and host replay after
    ScanChainGuard chain(t);
    if (c->opt_events & 2) HIPCHK(hipEventRecord(c->ev_c, c->stream));
    if (!have_tau) {   // probe: strided tiles -> per (tile, query) upper bounds
        MfmaParams Q = P;
        Q.tile_first = 0;
        Q.tile_step = tile_step;
        Q.tile_run_shift = stream ? 0u : probe_run_shift(c, TILE_ROWS * t->row_bytes, probe_tiles);   // (MF_STREAM tells the probe's tiles by their stride)
        Q.n_tiles = probe_tiles;
        Q.tilemin = (float *)c->dense.p;
        Q.tilemin_stride = probe_tiles;
        if (wide) launch_wide<MF_PROBE>(KS, wide_blocks, Q, dim3(std::min(probe_tiles, wg_cap), (unsigned)q_tiles), c->stream);
        else if (f64) launch_probe_f64(KS, Q, dim3(std::min(probe_tiles, wg_cap), (unsigned)q_tiles), c->stream, rt16);
        else launch_probe(KS, Q, dim3(std::min(probe_tiles, wg_cap), (unsigned)q_tiles), c->stream, rt16);
        HIPCHK(hipGetLastError());
        rc = launch_probe_threshold(c, nq, probe_tiles, k, M, stream);
        if (rc) return rc;
    }
    VSG_POLL_POINT(c);
    if (c->opt_events & 2) HIPCHK(hipEventRecord(c->ev_d, c->stream));
    chain.before_scan();   // behind the other lane's select kernel (ScanChain)
    if (c->opt_events & 1) HIPCHK(hipEventRecord(c->ev_a, c->stream));
    {   // filter: every tile once
        MfmaParams Q = P;
        Q.tile_first = 0;
        Q.tile_step = 1;
        Q.n_tiles = total_tiles;
        const uint32_t wgs = (uint32_t)c->n_cu * (uint32_t)c->opt_wg_per_cu;
        if (wide) {
            // one workgroup per CU and query tile in flight at a time per XCD slot: gridDim.x a multiple of 8, so the query
            // tiles of a row tile (blockIdx.y) land on one XCD and share its L2
            // (as many workgroups as are resident at once -- two per CU while a wave's fragments leave room for it, else one --
            // so that the query tiles of a row tile run at the same time)
            const uint32_t per_cu = (wide_blocks >= 2 || KS > 192) ? 1u : 2u;
            uint32_t gx = std::max<uint32_t>(8, std::min<uint32_t>(total_tiles, (uint32_t)c->n_cu * per_cu / (uint32_t)std::min<size_t>(q_tiles, 4)) / 8 * 8);
            if (c->opt_wide_gx > 0) gx = (uint32_t)c->opt_wide_gx;   // (diagnosis: option wide_gx)
            launch_wide<MF_FILTER>(KS, wide_blocks, Q, dim3(std::min(total_tiles, gx), (unsigned)q_tiles), c->stream);
        } else if (stream || stream_kernel_only) {
            // (measurement: 2 = no insertions, tau stays the small probe's; 4 = the full probe's thresholds, this kernel without
            // insertions and re-reads -- what its epilogue alone costs)
            Q.klist = (uint32_t *)c->klist.p;
            Q.klist_stride = (uint32_t)c->opt_stream_stride;
            Q.k = (c->opt_stream_tau == 2 || stream_kernel_only) ? 0xFFFFFFFFu : (uint32_t)k;
            Q.probe_shift = stream_shift;
            Q.refresh_early = (uint32_t)std::max<long>(c->opt_stream_early, 0);
            Q.refresh_mask = (uint32_t)std::max<long>(c->opt_stream_refresh, 1) - 1;   // (a power of two)
            Q.probe_tiles = stream ? probe_tiles : 0;
            launch_stream(KS, Q, n, wgs, (unsigned)q_tiles, c->stream);
        } else if (f64) launch_filter_f64(KS, Q, n, wgs, (unsigned)q_tiles, c->stream);
        else if (!(KS == 24 && launch_mfma_variant((int)c->opt_mfma_variant, Q, n, wgs, (unsigned)q_tiles, c->stream)))
            launch_filter(KS, Q, n, wgs, (unsigned)q_tiles, c->stream);
        HIPCHK(hipGetLastError());
    }
    if (c->opt_events & 1) HIPCHK(hipEventRecord(c->ev_b, c->stream));
    // the scan is in the stream: the next reader lane's probe may follow it and run beside this lane's re-rank and select
    // kernels (small grids both); its SCAN waits for them (ScanChain)
    chain.scan_submitted_if_early();
    VSG_POLL_POINT(c);
    wm0.mark("launches");
    wm0.flush("mfma_pre");
    rc = launch_exact_pairs(t, nq, ccap);  // exact re-rank of the survivors, in place
    if (rc) return rc;
    return collect_candidates(t, queries, nq, qstride, k, cap, ccap, ids, scores, counts, wide ? "k_mfma_filter_wide" : stream ? "k_mfma_filter(stream)" : "k_mfma_filter", &chain);
}