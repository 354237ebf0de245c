// hnsw_kernels.hpp -- HNSW query loops on the GPU.
//
// Restates, one wavefront per query, the reference's search:
//   searchBottomLayerEP + greedySearchLevel   algorithms/hnsw/hnsw.h:1967-1981, 1210-1258
//   searchBottomLayer_WithTimeout             hnsw.h:1983-2035
//   processCandidate                          hnsw.h:530-613
// with the SAME sequential admission rules, so on the same graph the GPU returns exactly what the
// reference's loop returns (tests compare against oracle/vso_hnsw.c, ids and scores bit for bit):
//   * candidate_set  = max-heap on (-dist, id): next = smallest dist, ties -> larger id;
//   * top_candidates = max-heap on (dist, label) capped at ef: evict the largest (dist, label);
//   * a neighbour is admitted iff lowerBound > d or |top| < ef; deleted nodes are traversed, not returned;
//   * stop when the best candidate is farther than lowerBound and |top| >= ef.
// Every distance is the reference-order exact distance (lane programs of exact_kernels.hpp): a group of
// VL lanes walks one neighbour's row, so a wave scores 64/VL unvisited neighbours at a time with all of
// a row's loads in flight together; the ~3 KB random row gathers are what the kernel waits for, and
// throughput comes from thousands of resident waves (one query each), not from one fast query.
// Both heaps are small sorted arrays in LDS updated wave-parallel (ballot-rank + shift).
// Visited marks are epoch tags in a per-wave slot of a global u16 array (visited_nodes_handler.h:23-57):
// no clearing between queries, no per-node locks (the device graph is a read-only snapshot).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "exact_kernels.hpp"

namespace vsg {

struct HnswParams {
    // rows
    const char *const *slabs;
    uint32_t slab_shift, slab_mask, row_stride;
    // lane program + queries
    const int32_t *offs;
    int steps;
    int reduce;            // LaneProgram::reduce
    const void *qperm;  // [nq][steps][VL] accumulator-typed
    int nq;
    int epilogue;
    uint32_t norm_off;     // byte offset of the row's trailing float norm (int8/uint8 Cosine rows)
    const float *qnorm;    // [nq] query norms (int8/uint8 Cosine)
    // graph snapshot
    const uint32_t *links0;     // [n][M0]
    const uint16_t *cnt0;       // [n]
    const uint32_t *upper_off;  // [n] index of the node's first upper block, 0xFFFFFFFF if level 0 only
    const uint32_t *upper;      // blocks of (1 + M) words: count, links
    const uint8_t *deleted;     // [n]
    const uint64_t *labels;     // [n]
    uint32_t M0, M;
    uint32_t entry;
    int max_level;
    uint32_t n;
    // visited tags
    uint16_t *tags;        // [slots][n]
    uint32_t *slot_epoch;  // [slots] last epoch used by the slot
    // search
    uint32_t ef, k, ccap;  // ccap = capacity of the candidate array (>= 2*ef)
    uint64_t *out_labels;  // [nq][k]
    void *out_scores;      // [nq][k] float (double for fp64 tables: DistType = double, hnsw_factory.cpp:47)
    uint32_t *out_counts;  // [nq]
    uint64_t *stat_dists;  // optional: total distance evaluations (atomicAdd), may be null
    uint32_t *next_query;  // work queue head (zeroed before the launch): waves draw queries one at a time
    // range mode (hnsw.h:616-680, 2087-2150): every result within `radius`, discovery order, up to rcap per query;
    // out_counts[q] = number found (bit 31: the candidate window overflowed, the caller must not trust the list)
    // multi-value index (hnsw_multi.h:108-112): top_candidates is the label-keyed updatable heap (utils/updatable_heap.h:20-113) --
    // a label keeps its lowest distance, the heap's size counts labels
    int multi;
    int range;
    double radius;
    double epsilon;
    uint32_t rcap;
};

// ---- wave-parallel sorted arrays in LDS (all 64 lanes call with uniform arguments) ----
// top: ascending by (dist, label); cand: ascending by (dist, then id DESCENDING) so that the front is
// the reference's candidate_set.top().
template <typename D> __device__ __forceinline__ bool top_less(D d1, uint64_t l1, D d2, uint64_t l2) {
    return d1 < d2 || (d1 == d2 && l1 < l2);
}
template <typename D> __device__ __forceinline__ bool cand_less(D d1, uint32_t i1, D d2, uint32_t i2) {
    return d1 < d2 || (d1 == d2 && i1 > i2);
}

template <typename DT> __device__ __forceinline__ uint32_t top_insert(DT *D, uint64_t *L, uint32_t n, DT d, uint64_t lab, int lane) {
    uint32_t pos = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        const bool lt = (i < n) && top_less(D[i], L[i], d, lab);
        pos += (uint32_t)__popcll(__ballot(lt));
    }
    // shift [pos, n) right by one, highest chunk first; within a chunk every read precedes every write
    for (int32_t base = (int32_t)((n - pos + 63) / 64 - 1) * 64; base >= 0; base -= 64) {
        const uint32_t i = pos + (uint32_t)base + lane;
        DT vd = (DT)0;
        uint64_t vl = 0;
        const bool act = i < n;
        if (act) { vd = D[i]; vl = L[i]; }
        if (act) { D[i + 1] = vd; L[i + 1] = vl; }
    }
    if (lane == 0) { D[pos] = d; L[pos] = lab; }
    return n + 1;
}
// updatable_max_heap::emplace (updatable_heap.h:93-113): a new label is inserted; a label already present keeps the lower of its
// two distances (and its place in the order moves with it)
template <typename DT> __device__ __forceinline__ uint32_t top_emplace_label(DT *D, uint64_t *L, uint32_t n, DT d, uint64_t lab, int lane) {
    uint32_t at = 0xFFFFFFFFu;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        const unsigned long long hit = __ballot(i < n && L[i] == lab);
        if (hit) at = base + (uint32_t)__ffsll((long long)hit) - 1u;   // (a label sits in the array at most once)
    }
    if (at != 0xFFFFFFFFu) {
        if (!(D[at] > d)) return n;   // "else if (existing priority > p)": only a strictly lower distance replaces
        __syncthreads();
        // close the gap at `at`: shift (at, n) left by one, lowest chunk first; within a chunk every read precedes every write
        for (uint32_t base = at + 1; base < n; base += 64) {
            const uint32_t i = base + lane;
            DT vd = (DT)0;
            uint64_t vl = 0;
            const bool act = i < n;
            if (act) { vd = D[i]; vl = L[i]; }
            __syncthreads();
            if (act) { D[i - 1] = vd; L[i - 1] = vl; }
            __syncthreads();
        }
        n--;
    }
    return top_insert(D, L, n, d, lab, lane);
}
template <typename DT> __device__ __forceinline__ uint32_t cand_insert(DT *D, uint32_t *I, uint32_t head, uint32_t tail, DT d, uint32_t id,
                                                int lane) {
    uint32_t pos = head;
    for (uint32_t base = head; base < tail; base += 64) {
        const uint32_t i = base + lane;
        const bool lt = (i < tail) && cand_less(D[i], I[i], d, id);
        pos += (uint32_t)__popcll(__ballot(lt));
    }
    for (int32_t base = (int32_t)((tail - pos + 63) / 64 - 1) * 64; base >= 0; base -= 64) {
        const uint32_t i = pos + (uint32_t)base + lane;
        DT vd = (DT)0;
        uint32_t vi = 0;
        const bool act = i < tail;
        if (act) { vd = D[i]; vi = I[i]; }
        if (act) { D[i + 1] = vd; I[i + 1] = vi; }
    }
    if (lane == 0) { D[pos] = d; I[pos] = id; }
    return tail + 1;
}

#ifndef HNSW_CH
#define HNSW_CH 12   // table steps loaded per round trip
#endif
#ifndef HNSW_U
#define HNSW_U 4     // rows per lane group in flight
#endif
template <int EK, int OPK>
__global__ __launch_bounds__(64) void k_hnsw_search(HnswParams P) {
    using E = Elem<EK>;
    using acc_t = typename E::acc_t;
    using dist_t = typename E::score_t;   // float; double for fp64 rows
    constexpr dist_t DIST_MAX = sizeof(dist_t) == 8 ? (dist_t)1.7976931348623157e308 : (dist_t)3.402823466e+38f;   // numeric_limits<DistType>::max()
    constexpr int VL = E::VL;
    constexpr int NG = 64 / VL;  // neighbour rows scored at a time
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int vl = lane % VL, grp = lane / VL;
    const int steps = P.steps;

    int32_t *offs_s = reinterpret_cast<int32_t *>(smem);
    size_t o = ((size_t)steps * VL * 4 + 15) & ~(size_t)15;
    acc_t *q_s = reinterpret_cast<acc_t *>(smem + o);
    o += ((size_t)steps * VL * sizeof(acc_t) + 15) & ~(size_t)15;
    uint64_t *top_l = reinterpret_cast<uint64_t *>(smem + o);
    o += (size_t)(P.ef + 2) * 8;
    dist_t *top_d = reinterpret_cast<dist_t *>(smem + o);
    o += ((size_t)(P.ef + 2) * sizeof(dist_t) + 15) & ~(size_t)15;
    dist_t *cand_d = reinterpret_cast<dist_t *>(smem + o);
    o += ((size_t)(2 * P.ccap + 2) * sizeof(dist_t) + 15) & ~(size_t)15;  // physical array = 2x the live window
    uint32_t *cand_i = reinterpret_cast<uint32_t *>(smem + o);
    o += ((size_t)(2 * P.ccap + 2) * 4 + 15) & ~(size_t)15;
    dist_t *nb_d = reinterpret_cast<dist_t *>(smem + o);
    o += 64 * sizeof(dist_t);
    uint32_t *nb_id = reinterpret_cast<uint32_t *>(smem + o);  // unvisited neighbours of the current node

    for (int i = lane; i < steps * VL; i += 64) offs_s[i] = P.offs[i];

    const uint32_t slot = blockIdx.x;
    uint16_t *tags = P.tags + (size_t)slot * P.n;
    uint32_t epoch = P.slot_epoch[slot];
    uint64_t n_dists = 0;
    float cur_qnorm = 0.f;  // norm of the query being searched (int8/uint8 Cosine)

    // exact distance of up to NG nodes (ids[g] for group g, valid when g < cnt); result in nb_d[out0 + g]
    // U * NG nodes per call: node (first + u*NG + grp) for u < U.  All U rows' loads of a chunk are in flight before the
    // first FMA: the search is latency-bound on random ~3 KB rows, and memory-level parallelism per wave is what buys
    // throughput (200 K x 768 rows, QPS at CH x U: 12x1 402 K, 24x1 461 K, 24x2 497 K, 12x4 550 K = 6.0 TB/s of gathered
    // rows; 6x8 and 8x8 the same, 12x6 and 24x4 lose occupancy)
    constexpr int U = HNSW_U;
    auto score_nodes = [&](const uint32_t *ids, uint32_t first, uint32_t cnt) {
        uint32_t idx[U];
        bool act[U];
        const char *rp[U];
        acc_t acc[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            idx[u] = first + (uint32_t)(u * NG) + grp;
            act[u] = idx[u] < cnt;
            const uint32_t node = act[u] ? ids[idx[u]] : ids[first];
            rp[u] = P.slabs[node >> P.slab_shift] + (size_t)(node & P.slab_mask) * P.row_stride;
            acc[u] = (acc_t)0;
        }
        // branch-free, CH steps at a time (idle table entries load offset 0 and leave the accumulator untouched)
        constexpr int CH = HNSW_CH;
        for (int s0 = 0; s0 < steps; s0 += CH) {
            int off[CH];
            acc_t xv[U][CH], qv[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) off[j] = (s0 + j < steps) ? offs_s[(s0 + j) * VL + vl] : -1;
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int j = 0; j < CH; j++) xv[u][j] = E::load(rp[u] + (off[j] >= 0 ? off[j] : 0));
#pragma unroll
            for (int j = 0; j < CH; j++) qv[j] = q_s[min(s0 + j, steps - 1) * VL + vl];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int j = 0; j < CH; j++) {
                    const acc_t t = acc_step<OPK>(xv[u][j], qv[j], acc[u]);
                    acc[u] = off[j] >= 0 ? t : acc[u];
                }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const typename Reduced<acc_t>::type a = lane_reduce<VL>((typename Reduced<acc_t>::type)acc[u], P.reduce);   // (integer rows: 64-bit total)
            if (vl == 0 && act[u]) {
                float nrow = 0.f;
                if (P.epilogue == EPI_INT_COS) {
                    const unsigned char *np = reinterpret_cast<const unsigned char *>(rp[u] + P.norm_off);
                    nrow = __uint_as_float((uint32_t)np[0] | ((uint32_t)np[1] << 8) | ((uint32_t)np[2] << 16) | ((uint32_t)np[3] << 24));
                }
                nb_d[idx[u]] = epilogue_score<dist_t>(a, P.epilogue, nrow, cur_qnorm);
            }
        }
    };

    // queries differ a lot in length (evaluations per query vary 2-3x), so they are drawn from a shared counter
    // instead of being dealt out in advance
    for (;;) {
        int q = 0;
        if (lane == 0) q = (int)atomicAdd(P.next_query, 1u);
        q = __shfl(q, 0);
        if (q >= P.nq) break;
        // ---- per-query state ----
        __syncthreads();
        {
            const acc_t *qg = reinterpret_cast<const acc_t *>(P.qperm) + (size_t)q * steps * VL;
            for (int i = lane; i < steps * VL; i += 64) q_s[i] = qg[i];
        }
        cur_qnorm = (P.epilogue == EPI_INT_COS) ? P.qnorm[q] : 0.f;
        epoch = epoch + 1;
        if ((epoch & 0xFFFFu) == 0) {  // u16 tag wrapped: clear this slot's tags once
            for (uint32_t i = lane; i < P.n; i += 64) tags[i] = 0;
            epoch = epoch + 1;
        }
        const uint16_t tag = (uint16_t)(epoch & 0xFFFFu);
        __syncthreads();

        // ---- entry point + greedy descent through the upper levels (hnsw.h:1967-1981, 1210-1258) ----
        uint32_t cur = P.entry;
        if (lane == 0) nb_id[0] = cur;
        __syncthreads();
        score_nodes(nb_id, 0, 1);
        n_dists += 1;
        __syncthreads();
        dist_t curd = nb_d[0];
        for (int level = P.max_level; level > 0; level--) {
            bool changed = true;
            while (changed) {
                changed = false;
                const uint32_t uo = P.upper_off[cur];
                const uint32_t *blk = P.upper + ((size_t)uo + (uint32_t)(level - 1)) * (P.M + 1);
                const uint32_t cnt = min(blk[0], P.M);
                __syncthreads();
                if ((uint32_t)lane < cnt) nb_id[lane] = blk[1 + lane];
                __syncthreads();
                for (uint32_t f = 0; f < cnt; f += U * NG) score_nodes(nb_id, f, cnt);
                n_dists += cnt;
                __syncthreads();
                // the reference walks the ORIGINAL node's link list to the end while updating the best
                for (uint32_t i = 0; i < cnt; i++) {
                    const dist_t d = nb_d[i];
                    if (d < curd) { curd = d; cur = nb_id[i]; changed = true; }
                }
            }
        }

        if (P.range) {
            // ---- level 0, range search (searchRangeBottomLayer_WithTimeout + processCandidate_RangeSearch) ----
            uint32_t chead = 0, ctail = 0, nres = 0;
            bool overflow = false;
            dist_t dyn, bound, epd;
            const dist_t radius = (dist_t)P.radius;
            auto emit = [&](uint32_t id, dist_t d) {
                if (lane == 0 && nres < P.rcap) {
                    P.out_labels[(size_t)q * P.rcap + nres] = P.labels[id];
                    reinterpret_cast<dist_t *>(P.out_scores)[(size_t)q * P.rcap + nres] = d;
                }
                nres++;
            };
            if (lane == 0) tags[cur] = tag;
            if (P.deleted[cur]) {
                epd = DIST_MAX;
                dyn = bound = epd;
            } else {
                epd = curd;
                dyn = epd;
                if (epd <= radius) {
                    emit(cur, epd);
                    dyn = radius;
                }
                bound = (dist_t)((double)dyn * (1.0 + P.epsilon));
            }
            ctail = cand_insert(cand_d, cand_i, chead, ctail, epd, cur, lane);
            __syncthreads();
            while (chead < ctail) {
                const dist_t cd = cand_d[chead];
                const uint32_t cnode = cand_i[chead];
                if (cd > bound) break;
                chead++;
                if (cd < dyn && cd >= radius) {
                    dyn = cd;
                    bound = (dist_t)((double)dyn * (1.0 + P.epsilon));
                }
                const uint32_t cnt = min((uint32_t)P.cnt0[cnode], P.M0);
                uint32_t nid = 0;
                bool fresh = false;
                if ((uint32_t)lane < cnt) {
                    nid = P.links0[(size_t)cnode * P.M0 + lane];
                    fresh = tags[nid] != tag;
                    if (fresh) tags[nid] = tag;
                }
                const unsigned long long fm = __ballot(fresh);
                const uint32_t nfresh = (uint32_t)__popcll(fm);
                __syncthreads();
                if (fresh) nb_id[__popcll(fm & ((1ull << lane) - 1ull))] = nid;
                __syncthreads();
                for (uint32_t f = 0; f < nfresh; f += U * NG) score_nodes(nb_id, f, nfresh);
                n_dists += nfresh;
                __syncthreads();
                for (uint32_t i = 0; i < nfresh; i++) {
                    const dist_t d = nb_d[i];
                    const uint32_t id = nb_id[i];
                    if (d < bound) {
                        bool keep = true;
                        if (ctail - chead >= P.ccap) {  // the reference's set is unbounded: report, do not guess
                            overflow = true;
                            if (cand_less(d, id, cand_d[ctail - 1], cand_i[ctail - 1])) ctail--;
                            else keep = false;
                        }
                        if (keep) {
                            if (ctail >= 2 * P.ccap) {
                                const uint32_t live = ctail - chead;
                                for (uint32_t b = 0; b < live; b += 64) {
                                    const uint32_t j = b + lane;
                                    dist_t vd = (dist_t)0;
                                    uint32_t vi = 0;
                                    if (j < live) { vd = cand_d[chead + j]; vi = cand_i[chead + j]; }
                                    __syncthreads();
                                    if (j < live) { cand_d[j] = vd; cand_i[j] = vi; }
                                    __syncthreads();
                                }
                                chead = 0;
                                ctail = live;
                            }
                            ctail = cand_insert(cand_d, cand_i, chead, ctail, d, id, lane);
                        }
                        if (d <= radius && !P.deleted[id]) emit(id, d);
                        __syncthreads();
                    }
                }
                __syncthreads();
            }
            if (lane == 0) P.out_counts[q] = nres | (overflow ? 0x80000000u : 0u);
            continue;
        }

        // ---- level 0: ef-bounded best-first search (hnsw.h:1983-2035) ----
        uint32_t top_n = 0, chead = 0, ctail = 0;
        dist_t lower;
        if (lane == 0) tags[cur] = tag;
        if (!P.deleted[cur]) {
            // (the reference recomputes dist(ep): same value as curd)
            lower = curd;
            top_n = top_insert(top_d, top_l, top_n, curd, P.labels[cur], lane);   // (the heap is empty: multi or not, one insert)
            ctail = cand_insert(cand_d, cand_i, chead, ctail, curd, cur, lane);
        } else {
            lower = DIST_MAX;
            ctail = cand_insert(cand_d, cand_i, chead, ctail, lower, cur, lane);
        }
        __syncthreads();

        while (chead < ctail) {
            const dist_t cd = cand_d[chead];
            const uint32_t cnode = cand_i[chead];
            if (cd > lower && top_n >= P.ef) break;
            chead++;
            // processCandidate(cnode, layer 0)
            const uint32_t cnt = min((uint32_t)P.cnt0[cnode], P.M0);
            uint32_t nid = 0;
            bool fresh = false;
            if ((uint32_t)lane < cnt) {
                nid = P.links0[(size_t)cnode * P.M0 + lane];
                fresh = tags[nid] != tag;
                if (fresh) tags[nid] = tag;
            }
            const unsigned long long fm = __ballot(fresh);
            const uint32_t nfresh = (uint32_t)__popcll(fm);
            __syncthreads();
            if (fresh) nb_id[__popcll(fm & ((1ull << lane) - 1ull))] = nid;  // keeps link order
            __syncthreads();
            for (uint32_t f = 0; f < nfresh; f += U * NG) score_nodes(nb_id, f, nfresh);
            n_dists += nfresh;
            __syncthreads();
            for (uint32_t i = 0; i < nfresh; i++) {
                const dist_t d = nb_d[i];
                const uint32_t id = nb_id[i];
                if (lower > d || top_n < P.ef) {
                    // candidate_set.emplace(-d, id).  The live window [chead, ctail) holds at most ccap
                    // entries; when full its worst entry is dropped (or the newcomer, if it is the
                    // worst): with more than 2*ef better visited nodes around, lowerBound is already
                    // below that distance, so the reference would never expand it either.
                    bool keep = true;
                    if (ctail - chead >= P.ccap) {
                        if (cand_less(d, id, cand_d[ctail - 1], cand_i[ctail - 1])) ctail--;
                        else keep = false;
                    }
                    if (keep) {
                        if (ctail >= 2 * P.ccap) {  // physical end reached: slide the live window to the front
                            const uint32_t live = ctail - chead;
                            for (uint32_t b = 0; b < live; b += 64) {
                                const uint32_t j = b + lane;
                                dist_t vd = (dist_t)0;
                                uint32_t vi = 0;
                                if (j < live) { vd = cand_d[chead + j]; vi = cand_i[chead + j]; }
                                __syncthreads();
                                if (j < live) { cand_d[j] = vd; cand_i[j] = vi; }
                                __syncthreads();
                            }
                            chead = 0;
                            ctail = live;
                        }
                        ctail = cand_insert(cand_d, cand_i, chead, ctail, d, id, lane);
                    }
                    if (!P.deleted[id]) {
                        __syncthreads();
                        top_n = P.multi ? top_emplace_label(top_d, top_l, top_n, d, P.labels[id], lane)
                                        : top_insert(top_d, top_l, top_n, d, P.labels[id], lane);
                    }
                    if (top_n > P.ef) top_n--;  // pop the largest (dist, label)
                    __syncthreads();
                    if (top_n > 0) lower = top_d[top_n - 1];
                }
            }
            __syncthreads();
        }

        // ---- results: the k smallest, ascending (dist, label) ----
        const uint32_t nres = min(top_n, P.k);
        for (uint32_t i = lane; i < P.k; i += 64) {
            if (i < nres) {
                P.out_labels[(size_t)q * P.k + i] = top_l[i];
                reinterpret_cast<dist_t *>(P.out_scores)[(size_t)q * P.k + i] = top_d[i];
            }
        }
        if (lane == 0) P.out_counts[q] = nres;
    }
    if (lane == 0) {
        P.slot_epoch[slot] = epoch;
        if (P.stat_dists) atomicAdd((unsigned long long *)P.stat_dists, (unsigned long long)n_dists);
    }
}

}  // namespace vsg
