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

/*
 * Batch iterator: the incremental walk of hnsw_batch_iterator.h:96-230 with the single-value heaps of
 * hnsw_single_batch_iterator.h:36-80 or the label-keyed ones of hnsw_multi_batch_iterator.h:38-96.  One call plays a whole
 * iteration: batch b asks for sizes[b] results (getNextResults: ef is raised to the batch size for that call), until the
 * iterator reports depletion or `max_batches` were taken.  Batches are written one after the other: out_counts[b] results of batch b
 * at offset sum(out_counts[0..b)).  Containers as plain arrays with linear scans: candidates and extras pop their smallest
 * (dist, id | label), top_candidates its largest (dist, label); the multi-value top keeps one entry per label with its lowest
 * distance (updatable_heap.h:93-113).  Returns the number of batches taken; *depleted_out = isDepleted() after the last one.
 */
typedef struct { top_t *v; size_t n, cap; } tvec_t;
static void tv_push(tvec_t *t, double d, uint64_t label) {
    if (t->n == t->cap) { t->cap = t->cap ? 2 * t->cap : 64; t->v = realloc(t->v, t->cap * sizeof(top_t)); }
    t->v[t->n].d = d; t->v[t->n].label = label; t->n++;
}
static size_t tv_extreme(const tvec_t *t, int want_max) {   /* index of the largest / smallest (d, label) */
    size_t b = 0;
    for (size_t i = 1; i < t->n; i++) {
        const int less = t->v[i].d < t->v[b].d || (t->v[i].d == t->v[b].d && t->v[i].label < t->v[b].label);
        const int more = t->v[b].d < t->v[i].d || (t->v[i].d == t->v[b].d && t->v[b].label < t->v[i].label);
        if (want_max ? more : less) b = i;
    }
    return b;
}
static void tv_remove(tvec_t *t, size_t i) { t->v[i] = t->v[t->n - 1]; t->n--; }
/* top_candidates->emplace: multi-value tops keep a label's lower distance */
static void top_emplace(tvec_t *top, int multi, double d, uint64_t label) {
    if (multi)
        for (size_t i = 0; i < top->n; i++)
            if (top->v[i].label == label) {
                if (top->v[i].d > d) top->v[i].d = d;
                return;
            }
    tv_push(top, d, label);
}
static int label_in(const uint64_t *set, size_t n, uint64_t label) {
    for (size_t i = 0; i < n; i++)
        if (set[i] == label) return 1;
    return 0;
}

size_t vso_hnsw_iterate(int type, int metric, int tier, size_t dim, const void *rows, size_t stride, uint32_t n,
                        const uint32_t *links0, const uint16_t *cnt0, uint32_t M0, const uint32_t *upper_off,
                        const uint32_t *upper, uint32_t M, const uint8_t *deleted, const uint64_t *labels,
                        uint32_t entry, int max_level, const void *query, size_t ef0, int multi, size_t n_labels,
                        const size_t *sizes, size_t max_batches, uint64_t *out_labels, double *out_scores, size_t *out_counts,
                        int *depleted_out) {
    const char *base = rows;
#define IDIST(node) (NARROW(type, vso_distance(type, metric, tier, dim, base + (size_t)(node)*stride, query)))
    uint8_t *visited = calloc(n ? n : 1, 1);
    cand_t *cand = malloc(((size_t)n + 1) * sizeof(cand_t));
    size_t nc = 0;
    tvec_t extras = {0, 0, 0};
    uint64_t *returned = malloc(((size_t)n + 1) * sizeof(uint64_t));
    size_t n_returned = 0, results = 0, written = 0, b = 0;
    double lower = INFINITY;
    int depleted = 0;
    uint32_t ep = 0xFFFFFFFFu;
    size_t ef = ef0;
    for (; b < max_batches; b++) {
        const size_t n_res = sizes[b];
        const size_t orig_ef = ef;
        if (orig_ef < n_res) ef = n_res;
        if (results == 0) {   /* searchBottomLayerEP */
            ep = entry;
            if (n == 0) ep = 0xFFFFFFFFu;
            if (ep != 0xFFFFFFFFu) {
                double curd = IDIST(ep);
                for (int level = max_level; level > 0; level--) {
                    int changed = 1;
                    while (changed) {
                        changed = 0;
                        uint32_t cnt;
                        const uint32_t *lk = links_at(ep, level, links0, cnt0, M0, upper_off, upper, M, &cnt);
                        for (uint32_t i = 0; i < cnt; i++) {
                            const double d = IDIST(lk[i]);
                            if (d < curd) { curd = d; ep = lk[i]; changed = 1; }
                        }
                    }
                }
            }
        }
        /* scanGraph */
        tvec_t top = {0, 0, 0};
        if (ep == 0xFFFFFFFFu) {
            depleted = 1;
        } else {
            if (results == 0 && extras.n == 0 && nc == 0) {
                lower = deleted[ep] ? (type == VSO_F64 ? 1.7976931348623157e308 : (double)3.402823466e+38f) : IDIST(ep);
                visited[ep] = 1;
                cand[nc].d = lower; cand[nc].id = ep; nc++;
            }
            /* fillFromExtras */
            while (top.n < ef && extras.n) {
                const size_t m = tv_extreme(&extras, 0);
                if (!multi || !label_in(returned, n_returned, extras.v[m].label)) top_emplace(&top, multi, extras.v[m].d, extras.v[m].label);
                tv_remove(&extras, m);
            }
            if (top.n != ef) {
                /* scanGraphInternal */
                while (nc) {
                    size_t m = 0;
                    for (size_t i = 1; i < nc; i++)
                        if (cand[i].d < cand[m].d || (cand[i].d == cand[m].d && cand[i].id < cand[m].id)) m = i;
                    const double cd = cand[m].d;
                    const uint32_t cnode = cand[m].id;
                    if (cd > lower && top.n >= ef) break;
                    if (!deleted[cnode]) {   /* updateHeaps */
                        if (!multi) {
                            if (top.n < ef) {
                                tv_push(&top, cd, labels[cnode]);
                                lower = top.v[tv_extreme(&top, 1)].d;
                            } else if (lower > cd) {
                                tv_push(&top, cd, labels[cnode]);
                                const size_t w = tv_extreme(&top, 1);
                                tv_push(&extras, top.v[w].d, top.v[w].label);
                                tv_remove(&top, w);
                                lower = top.v[tv_extreme(&top, 1)].d;
                            }
                        } else if (lower > cd || top.n < ef) {
                            if (!label_in(returned, n_returned, labels[cnode])) {
                                top_emplace(&top, 1, cd, labels[cnode]);
                                if (top.n > ef) {
                                    const size_t w = tv_extreme(&top, 1);
                                    tv_push(&extras, top.v[w].d, top.v[w].label);
                                    tv_remove(&top, w);
                                }
                                lower = top.v[tv_extreme(&top, 1)].d;
                            }
                        }
                    }
                    cand[m] = cand[nc - 1];
                    nc--;
                    const uint32_t cnt = cnt0[cnode];
                    const uint32_t *lk = links0 + (size_t)cnode * M0;
                    for (uint32_t j = 0; j < cnt; j++) {
                        const uint32_t c = lk[j];
                        if (visited[c]) continue;
                        visited[c] = 1;
                        cand[nc].d = IDIST(c); cand[nc].id = c; nc++;
                    }
                }
                if (top.n < ef) depleted = 1;
            }
        }
        /* prepareResults */
        while (top.n > n_res) {
            const size_t w = tv_extreme(&top, 1);
            tv_push(&extras, top.v[w].d, top.v[w].label);
            tv_remove(&top, w);
        }
        const size_t got = top.n;
        for (size_t i = got; i-- > 0;) {
            const size_t w = tv_extreme(&top, 1);
            out_labels[written + i] = top.v[w].label;
            out_scores[written + i] = top.v[w].d;
            if (multi) returned[n_returned++] = top.v[w].label;
            tv_remove(&top, w);
        }
        free(top.v);
        out_counts[b] = got;
        written += got;
        results += got;
        if (results == n_labels) depleted = 1;
        ef = orig_ef;
        if (depleted && extras.n == 0) { b++; break; }
    }
    if (depleted_out) *depleted_out = depleted && extras.n == 0;
    free(visited); free(cand); free(extras.v); free(returned);
    return b;
#undef IDIST
}

/*
 * ---- the INSERT path (round 6) -------------------------------------------------------------------------------------------
 *   getRandomLevel                                   hnsw.h:418-422   (std::default_random_engine seed 100, hnsw.h:230)
 *   storeNewElement / indexVector / appendVector     hnsw.h:1857-1946 (entry point and max level move at STORE time)
 *   insertElementToGraph                             hnsw.h:1567-1610
 *   greedySearchLevel<false>                         hnsw.h:1210-1258
 *   searchLayer + processCandidate                   hnsw.h:680-720, 530-613
 *   getNeighborsByHeuristic2 (+ _internal)           hnsw.h:720-797
 *   mutuallyConnectNewElement                        hnsw.h:889-963
 *   revisitNeighborConnections                       hnsw.h:800-886
 * Serial inserts in id order, no deletions.  PINNED: tests/golden/ref_hnsw_graphs.npz holds two graphs the REFERENCE built and
 * serialized (its unit tests' own data files, tests/unit/data/ *.v3: 1001 x 4 fp32, L2, M 8, efConstruction 10); re-inserting
 * their vectors here reproduces every level, every link list IN ORDER, the entry point and the unidirectional-edge sets
 * (tests/test_oracle_hnsw_build.py).
 *
 * What is library-defined in the reference and how it is restated:
 *  - the level generator is libstdc++'s minstd_rand0 (x <- 16807 x mod 2^31-1) through generate_canonical<double, 53>: two
 *    draws a, b -> ((a - 1) + (b - 1) R) / R^2 with R = 2^31 - 2, in double (the 1001 golden levels check it);
 *  - candidate_set / top_candidates are std::priority_queue<pair<dist, id>>: pairs are totally ordered (ids are unique), so
 *    WHICH element is on top never depends on the heap's layout.  The layout matters in one place: a candidate list shorter
 *    than M is linked in the container's order (mutuallyConnectNewElement iterates it unsorted).  Such a heap has never
 *    popped (it would hold ef >= M entries otherwise), so its layout is that of push_heap's sift-up alone -- canonical;
 *  - getNeighborsByHeuristic2 sorts by distance ONLY with std::sort ("we don't mind the secondary order"): among EXACTLY equal
 *    distances the reference's order is libstdc++'s introsort's.  That one step is not restated: vso_stdsort.cpp hands the list to
 *    the same std::sort with the same comparator (ties do occur: ~1 % of the lists of a 30 K x 32 fp32 build hold one).
 */
void vso_std_sort_by_distance(void *pairs, size_t n);   /* vso_stdsort.cpp: std::sort by .d */
typedef struct { double d; uint32_t id; } pr_t;                       /* pair<DistType, idType> */
static int pr_less(pr_t a, pr_t b) { return a.d < b.d || (!(b.d < a.d) && a.id < b.id); }   /* std::less<pair> */
typedef struct { pr_t *v; size_t n, cap; } heap_t;                    /* max-heap under pr_less, array = container order */
static void heap_push(heap_t *h, pr_t x) {
    if (h->n == h->cap) { h->cap = h->cap ? 2 * h->cap : 64; h->v = realloc(h->v, h->cap * sizeof(pr_t)); }
    size_t i = h->n++;
    while (i > 0) {                                                   /* __push_heap: sift the hole up */
        size_t p = (i - 1) / 2;
        if (!pr_less(h->v[p], x)) break;
        h->v[i] = h->v[p];
        i = p;
    }
    h->v[i] = x;
}
static void heap_pop(heap_t *h) {
    pr_t x = h->v[--h->n];
    size_t i = 0, n = h->n;
    if (!n) return;
    for (;;) {
        size_t c = 2 * i + 1;
        if (c >= n) break;
        if (c + 1 < n && pr_less(h->v[c], h->v[c + 1])) c++;
        if (!pr_less(x, h->v[c])) break;
        h->v[i] = h->v[c];
        i = c;
    }
    h->v[i] = x;
}

typedef struct {
    int type, metric, tier, fast;
    size_t dim, stride;
    const char *rows;
    uint32_t n, M, M0, efc;
    uint32_t *links0; uint16_t *cnt0; uint8_t *levels; uint32_t *upper_off; uint32_t *upper;
    uint32_t *tag, epoch;
    uint64_t evals;
} bld_t;
static double bdist(bld_t *b, uint32_t x, uint32_t y) {
    b->evals++;
    const void *px = b->rows + (size_t)x * b->stride, *py = b->rows + (size_t)y * b->stride;
    double d = b->fast ? vso_distance_fast_tier(b->type, b->metric, b->tier, b->dim, px, py) : vso_distance(b->type, b->metric, b->tier, b->dim, px, py);
    return NARROW(b->type, d);
}
static uint32_t *blinks(bld_t *b, uint32_t node, int level, uint32_t *cnt, uint32_t **cnt_word, uint16_t **cnt0_word) {
    if (level == 0) {
        *cnt = b->cnt0[node];
        if (cnt0_word) *cnt0_word = &b->cnt0[node];
        if (cnt_word) *cnt_word = NULL;
        return b->links0 + (size_t)node * b->M0;
    }
    uint32_t *blk = b->upper + ((size_t)b->upper_off[node] + (size_t)(level - 1)) * (b->M + 1);
    *cnt = blk[0];
    if (cnt_word) *cnt_word = blk;
    if (cnt0_word) *cnt0_word = NULL;
    return blk + 1;
}
static void bset_count(bld_t *b, uint32_t node, int level, uint32_t cnt) {
    if (level == 0) b->cnt0[node] = (uint16_t)cnt;
    else b->upper[((size_t)b->upper_off[node] + (size_t)(level - 1)) * (b->M + 1)] = cnt;
}
/* searchLayer (hnsw.h:680-720): the top_candidates heap, container order kept */
static void bsearch_layer(bld_t *b, uint32_t ep, uint32_t q, int level, size_t ef, heap_t *top, heap_t *cand) {
    top->n = cand->n = 0;
    if (++b->epoch == 0) { memset(b->tag, 0, (size_t)b->n * 4); b->epoch = 1; }
    const uint32_t tg = b->epoch;
    double d = bdist(b, ep, q), lower = d;                           /* (no deleted nodes on this path) */
    heap_push(top, (pr_t){d, ep});
    heap_push(cand, (pr_t){-d, ep});
    b->tag[ep] = tg;
    while (cand->n) {
        pr_t c = cand->v[0];
        if (-c.d > lower && top->n >= ef) break;
        heap_pop(cand);
        uint32_t cnt;
        const uint32_t *lk = blinks(b, c.id, level, &cnt, NULL, NULL);
        for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t nb = lk[j];
            if (b->tag[nb] == tg) continue;
            b->tag[nb] = tg;
            const double dd = bdist(b, nb, q);
            if (lower > dd || top->n < ef) {
                heap_push(cand, (pr_t){-dd, nb});
                heap_push(top, (pr_t){dd, nb});
                if (top->n > ef) heap_pop(top);
                lower = top->v[0].d;
            }
        }
    }
}
/* getNeighborsByHeuristic2_internal (hnsw.h:743-797): list -> the kept candidates, in order; removed[] gets the others */
static size_t bheuristic(bld_t *b, pr_t *list, size_t n, size_t M, uint32_t *removed, size_t *n_removed) {
    if (n_removed) *n_removed = 0;
    if (n < M) return n;
    vso_std_sort_by_distance(list, n);                                /* std::sort, distance only (vso_stdsort.cpp) */
    pr_t *kept = malloc((M + 1) * sizeof(pr_t));
    size_t nk = 0, i = 0;
    for (; i < n && nk < M; i++) {
        int good = 1;
        for (size_t s = 0; s < nk; s++)
            if (bdist(b, kept[s].id, list[i].id) < list[i].d) { good = 0; break; }
        if (good) kept[nk++] = list[i];
        else if (removed) removed[(*n_removed)++] = list[i].id;
    }
    if (removed) for (; i < n; i++) removed[(*n_removed)++] = list[i].id;
    memcpy(list, kept, nk * sizeof(pr_t));
    free(kept);
    return nk;
}
/* mutuallyConnectNewElement + revisitNeighborConnections (hnsw.h:800-963), serial: returns the next entry point */
static uint32_t bconnect(bld_t *b, uint32_t node, const heap_t *top, int level) {
    const uint32_t maxM = level ? b->M : b->M0;
    size_t n = top->n;
    pr_t *list = malloc((n + 1) * sizeof(pr_t));
    memcpy(list, top->v, n * sizeof(pr_t));
    uint32_t next;
    if (n < b->M) {                                                   /* std::min_element by distance: the first smallest */
        size_t bi = 0;
        for (size_t i = 1; i < n; i++) if (list[i].d < list[bi].d) bi = i;
        next = list[bi].id;
    } else {
        n = bheuristic(b, list, n, b->M, NULL, NULL);
        next = list[0].id;
    }
    pr_t *cands = malloc(((size_t)maxM + 2) * sizeof(pr_t));
    uint32_t *removed = malloc(((size_t)maxM + 2) * sizeof(uint32_t));
    for (size_t i = 0; i < n; i++) {
        const uint32_t nb = list[i].id;
        uint32_t mycnt, nbcnt;
        uint32_t *mine = blinks(b, node, level, &mycnt, NULL, NULL);
        if (mycnt == maxM) break;
        uint32_t *nl = blinks(b, nb, level, &nbcnt, NULL, NULL);
        if (nbcnt < maxM) {
            mine[mycnt] = nb;
            bset_count(b, node, level, mycnt + 1);
            nl[nbcnt] = node;
            bset_count(b, nb, level, nbcnt + 1);
            continue;
        }
        size_t nc = 0, nrem = 0;
        cands[nc++] = (pr_t){list[i].d, node};
        for (uint32_t j = 0; j < nbcnt; j++) cands[nc++] = (pr_t){bdist(b, nl[j], nb), nl[j]};
        bheuristic(b, cands, nc, maxM, removed, &nrem);
        int chosen = 1;
        for (size_t r = 0; r < nrem; r++) chosen &= removed[r] != node;
        uint32_t w = 0;
        for (uint32_t j = 0; j < nbcnt; j++) {
            int gone = 0;
            for (size_t r = 0; r < nrem; r++) gone |= removed[r] == nl[j];
            if (!gone) nl[w++] = nl[j];
        }
        if (mycnt < maxM) {
            mine[mycnt] = nb;
            bset_count(b, node, level, mycnt + 1);
            if (chosen && w < maxM) nl[w++] = node;
        }
        bset_count(b, nb, level, w);
    }
    free(list); free(cands); free(removed);
    return next;
}
typedef struct { uint64_t x; } minstd_t;
static uint32_t minstd_next(minstd_t *g) { g->x = (g->x * 16807ull) % 2147483647ull; return (uint32_t)g->x; }
static double minstd_canonical(minstd_t *g) {
    const double R = 2147483646.0;
    double sum = 0.0, tmp = 1.0;
    for (int k = 0; k < 2; k++) { sum += (double)(minstd_next(g) - 1u) * tmp; tmp *= R; }
    double r = sum / tmp;
    return r >= 1.0 ? nextafter(1.0, 0.0) : r;
}
/* levels of n nodes inserted in id order into an index created with M (hnsw.h:418-422; mult = 1 / log(M), hnsw.h:1644) */
void vso_hnsw_levels(uint32_t n, uint32_t M, uint32_t seed, uint8_t *levels) {
    minstd_t g = {seed % 2147483647ull ? seed % 2147483647ull : 1};
    const double mult = 1.0 / log(1.0 * M);
    for (uint32_t i = 0; i < n; i++) {
        const double r = -log(minstd_canonical(&g)) * mult;
        const size_t l = (size_t)r;
        levels[i] = (uint8_t)(l > 255 ? 255 : l);
    }
}
/* Builds the graph of rows 0..n-1 inserted in id order.  Outputs in the index's export layout (VecSimGpu_HnswGraphCopy):
 * links0 [n][2M], cnt0 [n], upper_off [n] (block number or 0xFFFFFFFF), upper: blocks of {count, links[M]} per level >= 1 in id
 * order, levels [n].  upper must hold (sum of levels) * (M + 1) words (vso_hnsw_levels gives the levels first).  fast != 0: the
 * intrinsics twins of the kernels (bit-identical, tests/test_oracle_kats.py).  Returns the number of distance evaluations. */
uint64_t vso_hnsw_build(int type, int metric, int tier, size_t dim, const void *rows, size_t stride, uint32_t n, uint32_t M, uint32_t efc,
                        uint32_t seed, int fast, uint32_t *links0, uint16_t *cnt0, uint8_t *levels, uint32_t *upper_off, uint32_t *upper,
                        uint32_t *entry_out, int *max_level_out) {
    bld_t b = {0};
    b.type = type; b.metric = metric; b.tier = tier; b.dim = dim; b.stride = stride; b.rows = rows; b.n = n;
    b.fast = fast && vso_fast_available(type, metric, tier, dim);
    b.M = M; b.M0 = 2 * M; b.efc = efc < M ? M : efc;
    b.links0 = links0; b.cnt0 = cnt0; b.levels = levels; b.upper_off = upper_off; b.upper = upper;
    b.tag = calloc(n ? n : 1, 4);
    vso_hnsw_levels(n, M, seed, levels);
    size_t blocks = 0;
    for (uint32_t i = 0; i < n; i++) {
        cnt0[i] = 0;
        upper_off[i] = levels[i] ? (uint32_t)blocks : 0xFFFFFFFFu;
        blocks += levels[i];
    }
    memset(upper, 0, blocks * (M + 1) * 4);
    memset(links0, 0, (size_t)n * b.M0 * 4);
    heap_t top = {0}, cand = {0};
    uint32_t entry = 0xFFFFFFFFu;
    int max_level = -1;
    for (uint32_t id = 0; id < n; id++) {
        const int level = levels[id];
        if (entry == 0xFFFFFFFFu) { entry = id; max_level = level; continue; }
        const uint32_t prev_entry = entry;
        const int prev_max = max_level;
        if (level > prev_max) { entry = id; max_level = level; }      /* storeNewElement */
        uint32_t cur = prev_entry;
        int common;
        if (level < prev_max) {
            double cd = bdist(&b, cur, id);
            for (int l = prev_max; l > level; l--) {
                int changed = 1;
                while (changed) {
                    changed = 0;
                    uint32_t cnt;
                    const uint32_t *lk = blinks(&b, cur, l, &cnt, NULL, NULL);   /* the ORIGINAL node's list to its end */
                    for (uint32_t i = 0; i < cnt; i++) {
                        const double d = bdist(&b, lk[i], id);
                        if (d < cd) { cd = d; cur = lk[i]; changed = 1; }
                    }
                }
            }
            common = level;
        } else {
            common = prev_max;
        }
        for (int l = common; l >= 0; l--) {
            bsearch_layer(&b, cur, id, l, b.efc, &top, &cand);
            if (top.n) cur = bconnect(&b, id, &top, l);
        }
    }
    *entry_out = entry;
    *max_level_out = max_level;
    free(top.v); free(cand.v); free(b.tag);
    return b.evals;
}
