// hnsw_ref_build.cpp -- the reference's HNSW INSERT path, step for step, for indexes that are to equal the reference's after the
// same sequence of VecSimIndex_AddVector calls (round-5 review: the graph is a deterministic function of insert order, the level
// generator and the index's own distance kernel).
//
//   storeNewElement / indexVector                    hnsw.h:1857-1946   entry point and max level move when the element is STORED
//   insertElementToGraph                             hnsw.h:1567-1610
//   greedySearchLevel<false>                         hnsw.h:1210-1258   (walks the original node's list to its end)
//   searchLayer + processCandidate                   hnsw.h:680-720, 530-613
//   getNeighborsByHeuristic2 (+ _internal)           hnsw.h:720-797     std::sort by distance only; first list uses M, not M0
//   mutuallyConnectNewElement                        hnsw.h:889-963     a list shorter than M is linked in the heap's CONTAINER order
//   revisitNeighborConnections                       hnsw.h:800-886     survivors keep their order, the new node goes last
//
// Serial (one insert at a time), distances in the index's tier order from host_lane_eval.h (the lane program the GPU kernels walk).
// The two heaps are std::vector + std::push_heap / std::pop_heap under std::less<pair<dist, id>> -- what std::priority_queue
// is defined as -- and the heuristic's sort is std::sort with the reference's comparator, so even the library-defined parts
// (heap layout, the order of exactly tied distances) are those of a reference built with the same libstdc++.
// Checked edge for edge against the graphs the reference itself built (tests/golden/ref_hnsw_graphs.npz, its unit tests' own
// serialized indexes) and against the oracle's independent restatement (oracle/vso_hnsw.c) on 3 K - 30 K-node cases:
// tests/test_gpu_hnsw_refbuild.py.  Deletes are NOT the reference's (marks + batch compaction, hnsw_index.h): equality holds for
// insert-only histories.
#include <algorithm>
#include <limits>

#include "hnsw_index.h"

namespace vsa {

namespace {
using Pair = std::pair<double, uint32_t>;   // pair<DistType, idType>; float distances widen exactly
struct RefHeap {                             // std::priority_queue<Pair>, container visible (vecsim_stl.h:63-83 exposes begin / end)
    std::vector<Pair> c;
    void push(double d, uint32_t id) {
        c.emplace_back(d, id);
        std::push_heap(c.begin(), c.end());
    }
    void pop() {
        std::pop_heap(c.begin(), c.end());
        c.pop_back();
    }
    const Pair &top() const { return c.front(); }
    size_t size() const { return c.size(); }
    bool empty() const { return c.empty(); }
    void clear() { c.clear(); }
};
constexpr uint32_t NONE = 0xFFFFFFFFu;
}  // namespace

double HnswIndex::refDistance(uint32_t a, uint32_t b) const {
    return ref_eval_.score(raw_.data() + (size_t)a * blob_bytes_, raw_.data() + (size_t)b * blob_bytes_, vec(a), vec(b));
}

// getNeighborsByHeuristic2_internal<record_removed> (hnsw.h:743-797)
static void ref_heuristic(const HnswIndex *ix, double (HnswIndex::*dist)(uint32_t, uint32_t) const, std::vector<Pair> &cands, size_t M,
                          std::vector<uint32_t> *removed) {
    if (cands.size() < M) return;
    std::vector<Pair> kept;
    kept.reserve(M);
    std::sort(cands.begin(), cands.end(), [](const Pair &a, const Pair &b) { return a.first < b.first; });
    auto cur = cands.begin();
    for (; cur != cands.end() && kept.size() < M; ++cur) {
        bool good = true;
        for (const Pair &s : kept)
            if ((ix->*dist)(s.second, cur->second) < cur->first) {
                if (removed) removed->push_back(cur->second);
                good = false;
                break;
            }
        if (good) kept.push_back(*cur);
    }
    if (removed)
        for (; cur != cands.end(); ++cur) removed->push_back(cur->second);
    cands.swap(kept);
}

void HnswIndex::insertNodeRef(uint32_t id) {
    const int level = level_[id];
    if (entry_ == NONE) {
        entry_ = id;
        max_level_ = level;
        return;
    }
    const uint32_t prev_entry = entry_;
    const int prev_max = max_level_;
    if (level > prev_max) {   // storeNewElement: the new element becomes the entry point BEFORE it is linked
        entry_ = id;
        max_level_ = level;
    }
    BuildCtx &bc = main_ctx_;
    if (bc.tag.size() < n_) bc.tag.resize(n_, 0u);
    uint32_t links[64];
    uint32_t cur = prev_entry;
    int common;
    if (level < prev_max) {
        double cd = refDistance(cur, id);
        uint32_t best_live = cur;
        for (int l = prev_max; l > level; l--) {   // greedySearchLevel<false>
            bool changed = true;
            while (changed) {
                changed = false;
                const uint32_t cnt = copyLinks(cur, l, links, false);
                for (uint32_t i = 0; i < cnt; i++) {
                    const double d = refDistance(links[i], id);
                    if (d < cd) {
                        cd = d;
                        cur = links[i];
                        changed = true;
                        if (!deleted_[links[i]]) best_live = cur;
                    }
                }
            }
            cur = best_live;
        }
        common = level;
    } else {
        common = prev_max;
    }
    RefHeap top, cand;
    std::vector<Pair> list, revisit;
    std::vector<uint32_t> removed;
    for (int l = common; l >= 0; l--) {
        // ---- searchLayer(cur, id, l, efConstruction)
        top.clear();
        cand.clear();
        if (++bc.epoch == 0) {
            std::fill(bc.tag.begin(), bc.tag.end(), 0u);
            bc.epoch = 1;
        }
        const uint32_t tg = bc.epoch;
        double lower;
        if (!deleted_[cur]) {
            const double d = refDistance(cur, id);
            lower = d;
            top.push(d, cur);
            cand.push(-d, cur);
        } else {
            lower = type_ == VecSimType_FLOAT64 ? std::numeric_limits<double>::max() : (double)std::numeric_limits<float>::max();
            cand.push(-lower, cur);
        }
        bc.tag[cur] = tg;
        while (!cand.empty()) {
            const Pair c = cand.top();
            if (-c.first > lower && top.size() >= ef_c_) break;
            cand.pop();
            const uint32_t cnt = copyLinks(c.second, l, links, false);
            for (uint32_t j = 0; j < cnt; j++) {
                const uint32_t nb = links[j];
                if (bc.tag[nb] == tg) continue;
                bc.tag[nb] = tg;
                const double d = refDistance(nb, id);
                if (lower > d || top.size() < ef_c_) {
                    cand.push(-d, nb);
                    if (!deleted_[nb]) top.push(d, nb);
                    if (top.size() > ef_c_) top.pop();
                    if (!top.empty()) lower = top.top().first;
                }
            }
        }
        if (top.empty()) continue;   // (the entry point was deleted and nothing else is reachable)
        // ---- mutuallyConnectNewElement(id, top, l)
        const size_t max_m = l ? M_ : M0_;
        list.assign(top.c.begin(), top.c.end());
        uint32_t next;
        if (list.size() < M_) {
            next = std::min_element(list.begin(), list.end(), [](const Pair &a, const Pair &b) { return a.first < b.first; })->second;
        } else {
            ref_heuristic(this, &HnswIndex::refDistance, list, M_, nullptr);
            next = list.front().second;
        }
        uint32_t *mine_cw = nullptr;
        uint32_t *mine = l == 0 ? links0_.data() + (size_t)id * M0_ : linksAt(id, l, &mine_cw);
        auto my_count = [&]() -> uint32_t { return l == 0 ? (uint32_t)cnt0_[id] : *mine_cw; };
        auto set_my_count = [&](uint32_t c) {
            if (l == 0) cnt0_[id] = (uint16_t)c;
            else *mine_cw = c;
        };
        for (const Pair &sel : list) {
            const uint32_t nb = sel.second;
            if (my_count() == max_m) break;
            if (deleted_[id] || deleted_[nb]) continue;
            uint32_t *nb_cw = nullptr;
            uint32_t *nl = l == 0 ? links0_.data() + (size_t)nb * M0_ : linksAt(nb, l, &nb_cw);
            uint32_t ncnt = l == 0 ? (uint32_t)cnt0_[nb] : *nb_cw;
            auto set_nb_count = [&](uint32_t c) {
                if (l == 0) cnt0_[nb] = (uint16_t)c;
                else *nb_cw = c;
            };
            if (ncnt < max_m) {
                mine[my_count()] = nb;
                set_my_count(my_count() + 1);
                nl[ncnt] = id;
                set_nb_count(ncnt + 1);
                continue;
            }
            // revisitNeighborConnections: the neighbour's list is full -- re-select among its neighbours and the new node
            revisit.clear();
            revisit.emplace_back(sel.first, id);
            for (uint32_t j = 0; j < ncnt; j++) revisit.emplace_back(refDistance(nl[j], nb), nl[j]);
            removed.clear();
            ref_heuristic(this, &HnswIndex::refDistance, revisit, max_m, &removed);
            const bool chosen = std::find(removed.begin(), removed.end(), id) == removed.end();
            uint32_t w = 0;
            for (uint32_t j = 0; j < ncnt; j++)
                if (std::find(removed.begin(), removed.end(), nl[j]) == removed.end()) nl[w++] = nl[j];
            if (my_count() < max_m) {
                mine[my_count()] = nb;
                set_my_count(my_count() + 1);
                if (chosen && w < max_m) nl[w++] = id;   // mutual; otherwise the edge id -> nb stays one-way
            }
            set_nb_count(w);
        }
        cur = next;
    }
}

}  // namespace vsa
