/*
 * vso_hnsw.c -- CPU restatement of the reference's HNSW *query* loops.  TEST INFRASTRUCTURE ONLY.
 *
 *   searchBottomLayerEP / greedySearchLevel<true>   algorithms/hnsw/hnsw.h:1967-1981, 1210-1258
 *   searchBottomLayer_WithTimeout                   hnsw.h:1983-2035
 *   processCandidate                                hnsw.h:530-613
 *   topKQuery (ef = max(ef, k), drain ascending)    hnsw.h:2037-2084
 *
 * The graph is an input (exported from the index under test), distances come from vso_distance (the
 * pinned kernel oracle).  The two heaps are kept as plain arrays with linear-scan "pop the maximum",
 * which is trivially the same element std::priority_queue<pair> pops: candidate_set orders by
 * (-dist, id), top_candidates by (dist, label).
 *
 * Parity pin: the reference's deterministic HNSW unit tests (tests/unit/test_hnsw.cpp:225-293, 1580-1632, 1801-1845:
 * closed forms over {i,i,i,i} vectors, the Cosine ranking, the range query with two epsilons) restated as data in
 * tests/golden/kat_hnsw.json; tests/test_gpu_hnsw.py asserts them on this restatement searching the graph the index under
 * test exports, and on the GPU search itself.  Beyond those: GPU search == this restatement on the same graph, bit for bit
 * (labels, order, scores, number of distance evaluations), plus recall against the exact Flat answer.  The graph is built by
 * the product's host builder, so those known answers pin the search loops on that graph, not the reference's build order.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "vso.h"

/* distances are kept as doubles: a float score widens exactly and orders the same; fp64 indexes have DistType = double
 * (index_factories/hnsw_factory.cpp:47) and keep every bit */
typedef struct { double d; uint32_t id; } cand_t;
typedef struct { double d; uint64_t label; } top_t;
#define NARROW(type, v) ((type) == VSO_F64 ? (double)(v) : (double)(float)(v))

static const uint32_t *links_at(uint32_t node, int level, const uint32_t *links0, const uint16_t *cnt0, uint32_t M0,
                                const uint32_t *upper_off, const uint32_t *upper, uint32_t M, uint32_t *cnt) {
    if (level == 0) {
        *cnt = cnt0[node];
        return links0 + (size_t)node * M0;
    }
    const uint32_t *blk = upper + ((size_t)upper_off[node] + (size_t)(level - 1)) * (M + 1);
    *cnt = blk[0];
    return blk + 1;
}

/* multi != 0: top_candidates is the label-keyed updatable_max_heap of a multi-value index (hnsw_multi.h:108-112,
 * utils/updatable_heap.h:93-113): a label that is already in keeps the lower of its two distances, the size counts labels */
static size_t hnsw_search_impl(int type, int metric, int tier, size_t dim, const void *rows, size_t stride, uint32_t n,
                       const uint32_t *links0, const uint16_t *cnt0, uint32_t M0, const uint32_t *upper_off,
                       const uint32_t *upper, uint32_t M, const uint8_t *deleted, const uint64_t *labels,
                       uint32_t entry, int max_level, const void *query, size_t k, size_t ef, uint64_t *out_labels,
                       double *out_scores, uint64_t *dist_evals, int multi) {
    if (n == 0 || k == 0 || entry == 0xFFFFFFFFu) return 0;
    if (ef < k) ef = k;
    const char *base = rows;
    uint64_t evals = 0;
#define DIST(node) (evals++, NARROW(type, vso_distance(type, metric, tier, dim, base + (size_t)(node)*stride, query)))
    uint32_t cur = entry;
    double curd = DIST(cur);
    for (int level = max_level; level > 0; level--) {
        int changed = 1;
        while (changed) {
            changed = 0;
            uint32_t cnt;
            const uint32_t *lk = links_at(cur, level, links0, cnt0, M0, upper_off, upper, M, &cnt);
            for (uint32_t i = 0; i < cnt; i++) {   /* walks the ORIGINAL node's list to its end */
                double d = DIST(lk[i]);
                if (d < curd) { curd = d; cur = lk[i]; changed = 1; }
            }
        }
    }
    uint8_t *visited = calloc(n, 1);
    cand_t *cand = malloc(((size_t)n + 1) * sizeof(cand_t));
    top_t *top = malloc((ef + 2) * sizeof(top_t));
    size_t nc = 0, nt = 0;
    double lower;
    if (!deleted[cur]) {
        double d = DIST(cur);   /* the reference re-evaluates dist(ep) here */
        lower = d;
        top[nt].d = d; top[nt].label = labels[cur]; nt++;
        cand[nc].d = d; cand[nc].id = cur; nc++;
    } else {
        lower = type == VSO_F64 ? 1.7976931348623157e308 : 3.402823466e+38f;   /* numeric_limits<DistType>::max() */
        cand[nc].d = lower; cand[nc].id = cur; nc++;
    }
    visited[cur] = 1;
    while (nc) {
        /* candidate_set.top(): max of (-d, id) == min d, ties -> larger id */
        size_t bi = 0;
        for (size_t i = 1; i < nc; i++)
            if (cand[i].d < cand[bi].d || (cand[i].d == cand[bi].d && cand[i].id > cand[bi].id)) bi = i;
        cand_t c = cand[bi];
        if (c.d > lower && nt >= ef) break;
        cand[bi] = cand[--nc];
        uint32_t cnt;
        const uint32_t *lk = links_at(c.id, 0, links0, cnt0, M0, upper_off, upper, M, &cnt);
        for (uint32_t j = 0; j < cnt; j++) {
            uint32_t nb = lk[j];
            if (visited[nb]) continue;
            visited[nb] = 1;
            double d = DIST(nb);
            if (lower > d || nt < ef) {
                cand[nc].d = d; cand[nc].id = nb; nc++;
                if (!deleted[nb]) {
                    size_t f = nt;
                    if (multi)
                        for (size_t i = 0; i < nt; i++)
                            if (top[i].label == labels[nb]) { f = i; break; }
                    if (f == nt) { top[nt].d = d; top[nt].label = labels[nb]; nt++; }
                    else if (top[f].d > d) top[f].d = d;
                }
                if (nt > ef) {   /* pop the max (d, label) */
                    size_t mi = 0;
                    for (size_t i = 1; i < nt; i++)
                        if (top[i].d > top[mi].d || (top[i].d == top[mi].d && top[i].label > top[mi].label)) mi = i;
                    top[mi] = top[--nt];
                }
                if (nt) {
                    double mx = top[0].d;
                    for (size_t i = 1; i < nt; i++) if (top[i].d > mx) mx = top[i].d;
                    lower = mx;
                }
            }
        }
    }
#undef DIST
    /* pop down to k, then ascending (d, label) */
    while (nt > k) {
        size_t mi = 0;
        for (size_t i = 1; i < nt; i++)
            if (top[i].d > top[mi].d || (top[i].d == top[mi].d && top[i].label > top[mi].label)) mi = i;
        top[mi] = top[--nt];
    }
    for (size_t i = 0; i < nt; i++)
        for (size_t j = i + 1; j < nt; j++)
            if (top[j].d < top[i].d || (top[j].d == top[i].d && top[j].label < top[i].label)) { top_t t = top[i]; top[i] = top[j]; top[j] = t; }
    for (size_t i = 0; i < nt; i++) { out_labels[i] = top[i].label; out_scores[i] = top[i].d; }
    if (dist_evals) *dist_evals = evals;
    free(visited); free(cand); free(top);
    return nt;
}
size_t vso_hnsw_search(int type, int metric, int tier, size_t dim, const void *rows, size_t stride, uint32_t n,
                       const uint32_t *links0, const uint16_t *cnt0, uint32_t M0, const uint32_t *upper_off,
                       const uint32_t *upper, uint32_t M, const uint8_t *deleted, const uint64_t *labels,
                       uint32_t entry, int max_level, const void *query, size_t k, size_t ef, uint64_t *out_labels,
                       double *out_scores, uint64_t *dist_evals) {
    return hnsw_search_impl(type, metric, tier, dim, rows, stride, n, links0, cnt0, M0, upper_off, upper, M, deleted, labels, entry,
                            max_level, query, k, ef, out_labels, out_scores, dist_evals, 0);
}
size_t vso_hnsw_search_multi(int type, int metric, int tier, size_t dim, const void *rows, size_t stride, uint32_t n,
                       const uint32_t *links0, const uint16_t *cnt0, uint32_t M0, const uint32_t *upper_off,
                       const uint32_t *upper, uint32_t M, const uint8_t *deleted, const uint64_t *labels,
                       uint32_t entry, int max_level, const void *query, size_t k, size_t ef, uint64_t *out_labels,
                       double *out_scores, uint64_t *dist_evals) {
    return hnsw_search_impl(type, metric, tier, dim, rows, stride, n, links0, cnt0, M0, upper_off, upper, M, deleted, labels, entry,
                            max_level, query, k, ef, out_labels, out_scores, dist_evals, 1);
}

/*
 * Range search: rangeQuery (hnsw.h:2153-2187) -> searchBottomLayerEP -> searchRangeBottomLayer_WithTimeout
 * (hnsw.h:2087-2150) with processCandidate_RangeSearch (hnsw.h:616-680).  candidate_set is unbounded; the
 * dynamic range shrinks towards `radius` as closer candidates are expanded and the search stops when the best
 * candidate is more than (1 + epsilon) times the dynamic range away.  Results come out in discovery order
 * (the C API sorts them afterwards).  dyn * (1.0 + epsilon) is evaluated in double and rounded to float, as the
 * reference's DistType = float assignment does.
 */
size_t vso_hnsw_range(int type, int metric, int tier, size_t dim, const void *rows, size_t stride, uint32_t n,
                      const uint32_t *links0, const uint16_t *cnt0, uint32_t M0, const uint32_t *upper_off,
                      const uint32_t *upper, uint32_t M, const uint8_t *deleted, const uint64_t *labels,
                      uint32_t entry, int max_level, const void *query, double radius_d, double epsilon,
                      uint64_t *out_labels, double *out_scores, size_t out_cap, uint64_t *dist_evals) {
    if (n == 0 || entry == 0xFFFFFFFFu) return 0;
    const char *base = rows;
    const double radius = NARROW(type, radius_d);
    uint64_t evals = 0;
#define RDIST(node) (evals++, NARROW(type, vso_distance(type, metric, tier, dim, base + (size_t)(node)*stride, query)))
    uint32_t cur = entry;
    double curd = RDIST(cur);
    for (int level = max_level; level > 0; level--) {
        int changed = 1;
        while (changed) {
            changed = 0;
            uint32_t cnt;
            const uint32_t *lk = links_at(cur, level, links0, cnt0, M0, upper_off, upper, M, &cnt);
            for (uint32_t i = 0; i < cnt; i++) {
                double d = RDIST(lk[i]);
                if (d < curd) { curd = d; cur = lk[i]; changed = 1; }
            }
        }
    }
    uint8_t *visited = calloc(n, 1);
    cand_t *cand = malloc(((size_t)n + 1) * sizeof(cand_t));
    size_t nc = 0, nres = 0;
    double ep_dist, dyn, bound;
    if (deleted[cur]) {
        ep_dist = type == VSO_F64 ? 1.7976931348623157e308 : 3.402823466e+38f;
        dyn = bound = ep_dist;
    } else {
        ep_dist = RDIST(cur);
        dyn = ep_dist;
        if (ep_dist <= radius) {
            if (nres < out_cap) { out_labels[nres] = labels[cur]; out_scores[nres] = ep_dist; }
            nres++;
            dyn = radius;
        }
        bound = NARROW(type, (double)dyn * (1.0 + epsilon));
    }
    cand[nc].d = ep_dist; cand[nc].id = cur; nc++;
    visited[cur] = 1;
    while (nc) {
        /* top of a max-heap on (-dist, id): smallest dist, ties -> largest id */
        size_t b = 0;
        for (size_t i = 1; i < nc; i++)
            if (cand[i].d < cand[b].d || (cand[i].d == cand[b].d && cand[i].id > cand[b].id)) b = i;
        const cand_t c = cand[b];
        if (c.d > bound) break;
        cand[b] = cand[--nc];
        if (c.d < dyn && c.d >= radius) {
            dyn = c.d;
            bound = NARROW(type, (double)dyn * (1.0 + epsilon));
        }
        uint32_t cnt;
        const uint32_t *lk = links_at(c.id, 0, links0, cnt0, M0, upper_off, upper, M, &cnt);
        for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t id = lk[j];
            if (visited[id]) continue;
            visited[id] = 1;
            const double d = RDIST(id);
            if (d < bound) {
                cand[nc].d = d; cand[nc].id = id; nc++;
                if (d <= radius && !deleted[id]) {
                    if (nres < out_cap) { out_labels[nres] = labels[id]; out_scores[nres] = d; }
                    nres++;
                }
            }
        }
    }
#undef RDIST
    free(visited);
    free(cand);
    if (dist_evals) *dist_evals = evals;
    return nres;
}
