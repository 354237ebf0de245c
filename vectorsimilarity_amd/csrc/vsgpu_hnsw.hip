// vsgpu_hnsw.hip -- HNSW graph snapshot + search entry points of include/vsgpu.h (kernel: hnsw_kernels.hpp)
#include "vsgpu_internal.hpp"
#include "hnsw_kernels.hpp"

using namespace vsg;

// ------------------------------------------------------------------ HNSW graph snapshot + search
struct vsgpu_graph {
    vsgpu_table *t = nullptr;
    // non-null: a view for another reader (vsgpu_graph_view_create): the snapshot -- links, deletion marks, labels, entry point --
    // is the parent's; visited tags, result buffers, query staging and the stream are this view's own (its table is a view too)
    vsgpu_graph *parent = nullptr;
    uint32_t M = 16, M0 = 32;
    int multi = 0;   // labels may repeat: the search keeps one entry per label (vsgpu_graph_set_multi)
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
    vsgpu_graph *g = new vsgpu_graph();
    g->t = t;
    g->M = (uint32_t)M;
    g->M0 = (uint32_t)(2 * M);
    return g;
}
extern "C" vsgpu_graph *vsgpu_graph_view_create(vsgpu_graph *parent, vsgpu_table *view_table) {
    if (!parent || parent->parent || !view_table || view_table->parent != parent->t) {
        fail(VSGPU_ERR_ARG, "graph view: needs a graph and a view of its table");
        return nullptr;
    }
    vsgpu_graph *g = new vsgpu_graph();
    g->t = view_table;
    g->parent = parent;
    g->M = parent->M;
    g->M0 = parent->M0;
    return g;
}
extern "C" void vsgpu_graph_set_multi(vsgpu_graph *g, int multi) {
    if (g) g->multi = multi ? 1 : 0;
}
extern "C" void vsgpu_graph_destroy(vsgpu_graph *g) {
    if (!g) return;
    (void)hipSetDevice(g->t->ctx->device);
    (void)hipStreamSynchronize(g->t->ctx->stream);
    for (DevBuf *b : {&g->links0, &g->cnt0, &g->upper_off, &g->upper, &g->deleted, &g->labels, &g->tags, &g->slot_epoch,
                      &g->out_labels, &g->out_scores, &g->out_counts, &g->stat})
        if (b->p) (void)hipFree(b->p);   // (a view never allocated the snapshot buffers)
    delete g;
}
extern "C" int vsgpu_graph_upload(vsgpu_graph *g, size_t n, const uint32_t *links0, const uint16_t *cnt0,
                                  const uint32_t *upper_off, const uint32_t *upper, size_t upper_words,
                                  const uint8_t *deleted, const uint64_t *labels, uint32_t entry, int max_level) {
    if (g->parent) return fail(VSGPU_ERR_ARG, "graph upload through a view");
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
    else if (opk == OP_IP_DPBF16) {
        if constexpr (EK == EK_BF16) hipLaunchKernelGGL((k_hnsw_search<EK, OP_IP_DPBF16>), grid, dim3(64), lds, s, P);
    }
    else if (opk == OP_L2_F16ACC) {
        if constexpr (EK == EK_F16) hipLaunchKernelGGL((k_hnsw_search<EK, OP_L2_F16ACC>), grid, dim3(64), lds, s, P);
    }
    else if (opk == OP_IP_F16ACC) {
        if constexpr (EK == EK_F16) hipLaunchKernelGGL((k_hnsw_search<EK, OP_IP_F16ACC>), grid, dim3(64), lds, s, P);
    }
    else hipLaunchKernelGGL((k_hnsw_search<EK, OP_IP_MULADD>), grid, dim3(64), lds, s, P);
}

// top-k search (range == nullptr) or range search (range = {radius, epsilon}; k is then the result capacity per query)
static int graph_run(vsgpu_graph *g, const void *queries, size_t nq, size_t qstride, size_t k, size_t ef, const double *range,
                     uint64_t *labels, double *scores, uint32_t *counts, uint64_t *dist_evals) {
    vsgpu_table *t = g->t;
    vsgpu_ctx *c = t->ctx;
    const vsgpu_graph *snap = g->parent ? g->parent : g;   // whose links / marks / labels / entry point are searched
    if (dist_evals) *dist_evals = 0;
    if (nq == 0) return VSGPU_OK;
    if (k == 0 || snap->n == 0 || snap->entry == 0xFFFFFFFFu) {
        for (size_t q = 0; q < nq; q++) counts[q] = 0;
        return VSGPU_OK;
    }
    HIPCHK(hipSetDevice(c->device));
    ef = range ? 1 : std::max(ef, k);
    if (ef > 4096) return fail(VSGPU_ERR_UNSUPPORTED, "ef %zu too large for the LDS heaps", ef);
    int rc = stage_queries(t, queries, nq, qstride);
    if (rc) return rc;
    const size_t ab = acc_bytes(t->type);
    const size_t db = t->type == VSGPU_F64 ? 8 : 4;   // distance bytes: float scores, double for fp64 rows (DistType = double)
    size_t ccap = 2 * ef;
    if (range) {
        // the reference's candidate set is unbounded: give the window what LDS allows (overflow is reported)
        const size_t fixed = 2 * (((size_t)t->prog.steps * t->prog.vl * std::max<size_t>(ab, 4) + 15) & ~(size_t)15) + 1024;
        ccap = 64;
        while (ccap < 3072 && fixed + (2 * (2 * ccap) + 2) * (4 + db) + 64 <= 60 * 1024) ccap *= 2;
    }
    size_t lds = (((size_t)t->prog.steps * t->prog.vl * 4 + 15) & ~(size_t)15);
    lds += (((size_t)t->prog.steps * t->prog.vl * ab + 15) & ~(size_t)15);
    lds += (ef + 2) * 8;
    lds += (((ef + 2) * db + 15) & ~(size_t)15);
    lds += (((2 * ccap + 2) * db + 15) & ~(size_t)15);
    lds += (((2 * ccap + 2) * 4 + 15) & ~(size_t)15);
    lds += 64 * db + 64 * 4;
    if (lds > 64 * 1024) return fail(VSGPU_ERR_UNSUPPORTED, "ef %zu / dim %zu need %zu B of LDS per query", ef, t->dim, lds);
    // resident search waves = tag slots
    const size_t slots = std::min<size_t>(nq, (size_t)c->n_cu * (size_t)c->opt_hnsw_slots);
    if (slots > g->tag_slots || snap->n > g->tag_n) {
        const size_t ns = std::max(slots, g->tag_slots), nn = std::max(snap->n, g->tag_n);
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
    if ((rc = ensure(c, g->out_scores, nq * k * db))) return rc;
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
    P.reduce = t->prog.reduce;
    P.qperm = c->qperm.p;
    P.nq = (int)nq;
    P.epilogue = t->epi;
    P.norm_off = (uint32_t)t->dim;
    P.qnorm = (const float *)c->qnorm.p;
    P.links0 = (const uint32_t *)snap->links0.p;
    P.cnt0 = (const uint16_t *)snap->cnt0.p;
    P.upper_off = (const uint32_t *)snap->upper_off.p;
    P.upper = (const uint32_t *)snap->upper.p;
    P.deleted = (const uint8_t *)snap->deleted.p;
    P.labels = (const uint64_t *)snap->labels.p;
    P.M0 = g->M0;
    P.M = g->M;
    P.entry = snap->entry;
    P.max_level = snap->max_level;
    P.multi = snap->multi;
    P.n = (uint32_t)g->tag_n;  // tag row pitch
    P.tags = (uint16_t *)g->tags.p;
    P.slot_epoch = (uint32_t *)g->slot_epoch.p;
    P.ef = (uint32_t)ef;
    P.k = (uint32_t)k;
    P.ccap = (uint32_t)ccap;
    P.out_labels = (uint64_t *)g->out_labels.p;
    P.out_scores = g->out_scores.p;
    P.out_counts = (uint32_t *)g->out_counts.p;
    P.stat_dists = (uint64_t *)g->stat.p;
    P.next_query = (uint32_t *)((char *)g->stat.p + 8);
    if (range) {
        P.range = 1;
        P.radius = range[0];
        P.epsilon = range[1];
        P.rcap = (uint32_t)k;
    }
    if (c->opt_events & 1) HIPCHK(hipEventRecord(c->ev_a, c->stream));
    const dim3 grid((unsigned)slots);
    switch (t->ek) {
    case EK_F32: launch_hnsw_ek<EK_F32>(t->opk, P, grid, lds, c->stream); break;
    case EK_F64: launch_hnsw_ek<EK_F64>(t->opk, P, grid, lds, c->stream); break;
    case EK_BF16: launch_hnsw_ek<EK_BF16>(t->opk, P, grid, lds, c->stream); break;
    case EK_F16: launch_hnsw_ek<EK_F16>(t->opk, P, grid, lds, c->stream); break;
    case EK_I8: launch_hnsw_ek<EK_I8>(t->opk, P, grid, lds, c->stream); break;
    default: launch_hnsw_ek<EK_U8>(t->opk, P, grid, lds, c->stream); break;
    }
    HIPCHK(hipGetLastError());
    if (c->opt_events & 1) HIPCHK(hipEventRecord(c->ev_b, c->stream));
    std::vector<float> hs(db == 4 ? nq * k : 0);
    uint64_t hstat = 0;
    HIPCHK(hipMemcpyAsync(labels, g->out_labels.p, nq * k * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(db == 4 ? (void *)hs.data() : (void *)scores, g->out_scores.p, nq * k * db, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(counts, g->out_counts.p, nq * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&hstat, g->stat.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < hs.size(); i++) scores[i] = (double)hs[i];
    if (dist_evals) *dist_evals = hstat;
    {
        float ms = 0;
        if ((c->opt_events & 1) && hipEventElapsedTime(&ms, c->ev_a, c->ev_b) == hipSuccess) c->stats.scan_ms += ms;
        c->stats.scan_launches += 1;
        c->stats.scan_rows += hstat;                    // rows gathered = distance evaluations
        c->stats.scan_bytes += hstat * t->row_bytes;
        snprintf(c->stats.scan_kernel, sizeof c->stats.scan_kernel, range ? "k_hnsw_search(range)" : "k_hnsw_search");
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