// hnsw_iter.cpp -- the HNSW index's batch iterator: the reference's incremental graph walk
// (algorithms/hnsw/hnsw_batch_iterator.h:96-230, hnsw_single_batch_iterator.h:36-80, hnsw_multi_batch_iterator.h:38-96).
//
// The walk's control state -- the candidates min-heap, the "extras" min-heap carried from batch to batch, the visited marks,
// lower_bound, depleted -- lives on the host, in the reference's own containers' order (ref_heap.h); every DISTANCE the walk needs
// is computed on the GPU in the index's reference-order arithmetic (vsgpu_scores_of: the exact kernels over the HBM-resident rows),
// so a batch holds the labels and the scores the reference's iterator returns on the same graph, bit for bit
// (tests/test_gpu_hnsw.py against oracle/vso_hnsw.c's twin of the same walk, and the closed forms of tests/unit/test_hnsw.cpp:912-1118).
//
// One expansion needs the distances of one node's unvisited neighbours (<= M0 rows): a kernel launch and a download per
// expansion would make the walk launch-bound (15-25 us each).  Distances are pure functions of (query, row), so the walk asks for
// them AHEAD: whenever an expansion finds a neighbour whose distance is not known yet, the request also carries the unvisited
// neighbours of the next few entries of the candidates heap (VECSIM_HNSW_ITER_AHEAD, default 8).  What the walk does with a
// distance -- the order of pushes and pops, the visited marks -- is untouched by when the number was computed.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <unordered_set>

#include "hnsw_index.h"
#include "ref_heap.h"

namespace vsa {

class HnswWalk final : public IterWalker {
public:
    HnswWalk(HnswIndex *ix, std::vector<char> query, VecSimQueryParams *qp)
        : ix_(ix), query_(std::move(query)), tctx_(qp ? qp->timeoutCtx : nullptr) {
        ef_ = (qp && qp->hnswRuntimeParams.efRuntime > 0) ? qp->hnswRuntimeParams.efRuntime : ix_->ef_;
        if (const char *e = std::getenv("VECSIM_HNSW_ITER_AHEAD")) ahead_ = (size_t)std::max(0, std::atoi(e));
        visited_.assign(ix_->n_, 0);
        ix_->walkers_++;
    }
    ~HnswWalk() override { ix_->walkers_--; }

    VecSimQueryReply *next(size_t n_res, VecSimQueryReply_Order order) override {
        std::lock_guard<std::recursive_mutex> gpu_lock(ix_->gpu_mu_);
        auto *batch = new VecSimQueryReply();
        const size_t orig_ef = ef_;
        if (orig_ef < n_res) ef_ = n_res;
        if (results_count_ == 0) {
            entry_ = searchBottomLayerEP(&batch->code);
            if (batch->code != VecSim_QueryReply_OK) return batch;   // (as upstream: a raised ef stays raised after a timeout)
        }
        if (ix_->multi_) {
            RefUpdatableMaxHeap top;
            scanGraph(top, &batch->code);
            if (batch->code == VecSim_QueryReply_OK) prepareResults(batch, top, n_res);
        } else {
            RefMaxHeap<> top;
            scanGraph(top, &batch->code);
            if (batch->code == VecSim_QueryReply_OK) prepareResults(batch, top, n_res);
        }
        if (batch->code != VecSim_QueryReply_OK) return batch;
        results_count_ += batch->results.size();
        if (results_count_ == ix_->indexLabelCount()) depleted_ = true;
        if (order == BY_ID) sort_reply(batch, BY_ID);
        ef_ = orig_ef;
        return batch;
    }
    bool depleted() const override { return depleted_ && extras_.empty(); }
    void reset() override {
        results_count_ = 0;
        depleted_ = false;
        std::fill(visited_.begin(), visited_.end(), 0);
        visited_.resize(ix_->n_, 0);
        lower_bound_ = std::numeric_limits<double>::infinity();
        candidates_.clear();
        extras_ = decltype(extras_)();
        returned_.clear();
        known_.clear();
    }

private:
    static constexpr uint32_t INVALID = 0xFFFFFFFFu;
    using Cand = std::pair<double, uint32_t>;

    // ---- distances: the GPU's, fetched in groups ----
    bool fetch(std::vector<uint32_t> &want) {
        if (want.empty()) return true;
        if (ix_->syncDevice()) return false;
        std::vector<double> s(want.size());
        if (vsgpu_scores_of(ix_->table_, query_.data(), want.data(), want.size(), s.data())) return false;
        for (size_t i = 0; i < want.size(); i++) known_[want[i]] = s[i];
        n_fetches_++;
        return true;
    }
    // unvisited neighbours of `node` at level 0 whose distance is not known yet
    void wanted(uint32_t node, std::vector<uint32_t> &want, std::unordered_set<uint32_t> &seen) const {
        const uint32_t *l = ix_->links0_.data() + (size_t)node * ix_->M0_;
        for (uint32_t j = 0; j < ix_->cnt0_[node]; j++) {
            const uint32_t c = l[j];
            if (c < visited_.size() && visited_[c]) continue;
            if (known_.count(c) || !seen.insert(c).second) continue;
            want.push_back(c);
        }
    }
    double take(uint32_t id) {   // a distance is used once (a node is visited once)
        auto f = known_.find(id);
        const double d = f->second;
        known_.erase(f);
        return d;
    }
    double maxDist() const {   // std::numeric_limits<DistType>::max()
        return ix_->type_ == VecSimType_FLOAT64 ? std::numeric_limits<double>::max() : (double)std::numeric_limits<float>::max();
    }

    // hnsw.h:1967-1981 + greedySearchLevel<true> (hnsw.h:1210-1258): the walk down to level 1
    uint32_t searchBottomLayerEP(VecSimQueryReply_Code *rc) {
        *rc = VecSim_QueryReply_OK;
        uint32_t cur = ix_->entry_;
        if (cur == INVALID || ix_->n_ == 0) return INVALID;
        std::vector<uint32_t> one{cur};
        if (!fetch(one)) return failed(rc);
        double cur_dist = take(cur);
        for (int level = ix_->max_level_; level > 0 && cur != INVALID; level--) {
            bool changed;
            do {
                if (timed_out(tctx_)) {
                    *rc = VecSim_QueryReply_TimedOut;
                    return INVALID;
                }
                changed = false;
                uint32_t *cw;
                const uint32_t *l = ix_->linksAt(cur, level, &cw);
                const uint32_t cnt = *cw;
                std::vector<uint32_t> want(l, l + cnt);   // (every link of the ORIGINAL node, the best moving meanwhile)
                std::vector<double> d(cnt);
                if (cnt) {
                    if (ix_->syncDevice() || vsgpu_scores_of(ix_->table_, query_.data(), want.data(), cnt, d.data())) return failed(rc);
                    n_fetches_++;
                }
                for (uint32_t i = 0; i < cnt; i++)
                    if (d[i] < cur_dist) {
                        cur_dist = d[i];
                        cur = want[i];
                        changed = true;
                    }
            } while (changed);
        }
        return cur;
    }
    uint32_t failed(VecSimQueryReply_Code *rc) {   // a GPU failure: reported the only way a batch can carry it
        std::fprintf(stderr, "vecsim_amd: HNSW batch iterator: GPU distance pass failed: %s\n", vsgpu_last_error());
        *rc = VecSim_QueryReply_TimedOut;
        return INVALID;
    }

    void candPush(double d, uint32_t id) {
        candidates_.emplace_back(d, id);
        std::push_heap(candidates_.begin(), candidates_.end(), RefPairGreater());
    }
    void candPop() {
        std::pop_heap(candidates_.begin(), candidates_.end(), RefPairGreater());
        candidates_.pop_back();
    }

    // hnsw_single_batch_iterator.h:62-80 / hnsw_multi_batch_iterator.h:70-88
    void updateHeaps(RefMaxHeap<> &top, double dist, uint32_t id) {
        if (top.size() < ef_) {
            top.emplace(dist, (size_t)ix_->labels_[id]);
            lower_bound_ = top.top().first;
        } else if (lower_bound_ > dist) {
            top.emplace(dist, (size_t)ix_->labels_[id]);
            extras_.emplace(top.top().first, top.top().second);
            top.pop();
            lower_bound_ = top.top().first;
        }
    }
    void updateHeaps(RefUpdatableMaxHeap &top, double dist, uint32_t id) {
        if (lower_bound_ > dist || top.size() < ef_) {
            const size_t label = (size_t)ix_->labels_[id];
            if (returned_.find(label) == returned_.end()) {
                top.emplace(dist, label);
                if (top.size() > ef_) {
                    extras_.emplace(top.top().first, top.top().second);
                    top.pop();
                }
                lower_bound_ = top.top().first;
            }
        }
    }
    void fillFromExtras(RefMaxHeap<> &top) {
        while (top.size() < ef_ && !extras_.empty()) {
            top.emplace(extras_.top().first, extras_.top().second);
            extras_.pop();
        }
    }
    void fillFromExtras(RefUpdatableMaxHeap &top) {
        while (top.size() < ef_ && !extras_.empty()) {
            if (returned_.find(extras_.top().second) == returned_.end()) top.emplace(extras_.top().first, extras_.top().second);
            extras_.pop();
        }
    }
    template <typename Heap> void prepareResults(VecSimQueryReply *rep, Heap &top, size_t n_res) {
        while (top.size() > n_res) {
            extras_.emplace(top.top().first, top.top().second);
            top.pop();
        }
        rep->results.resize(top.size());
        for (size_t i = rep->results.size(); i-- > 0;) {
            rep->results[i].score = top.top().first;
            rep->results[i].id = top.top().second;
            if (ix_->multi_) returned_.insert(top.top().second);
            top.pop();
        }
    }

    // hnsw_batch_iterator.h:96-162
    template <typename Heap> VecSimQueryReply_Code scanGraphInternal(Heap &top) {
        std::vector<uint32_t> want;
        std::unordered_set<uint32_t> seen;
        while (!candidates_.empty()) {
            const double cur_dist = candidates_.front().first;
            const uint32_t cur = candidates_.front().second;
            if (cur_dist > lower_bound_ && top.size() >= ef_) break;
            if (timed_out(tctx_)) return VecSim_QueryReply_TimedOut;
            if (!ix_->deleted_[cur]) updateHeaps(top, cur_dist, cur);
            candPop();
            // distances this expansion needs and does not have: ask for them together with what the next expansions will want
            want.clear();
            seen.clear();
            wanted(cur, want, seen);
            if (!want.empty()) {
                for (size_t a = 0; a < std::min(ahead_, candidates_.size()); a++) wanted(candidates_[a].second, want, seen);
                if (!fetch(want)) {
                    std::fprintf(stderr, "vecsim_amd: HNSW batch iterator: GPU distance pass failed: %s\n", vsgpu_last_error());
                    return VecSim_QueryReply_TimedOut;
                }
            }
            const uint32_t *l = ix_->links0_.data() + (size_t)cur * ix_->M0_;
            for (uint32_t j = 0; j < ix_->cnt0_[cur]; j++) {
                const uint32_t c = l[j];
                if (visited_[c]) continue;
                visited_[c] = 1;
                candPush(take(c), c);
            }
        }
        return VecSim_QueryReply_OK;
    }

    // hnsw_batch_iterator.h:164-204
    template <typename Heap> void scanGraph(Heap &top, VecSimQueryReply_Code *rc) {
        if (entry_ == INVALID) {
            depleted_ = true;
            return;
        }
        if (visited_.size() < ix_->n_) visited_.resize(ix_->n_, 0);
        if (results_count_ == 0 && extras_.empty() && candidates_.empty()) {
            if (!ix_->deleted_[entry_]) {
                std::vector<uint32_t> one{entry_};
                if (!fetch(one)) {
                    failed(rc);
                    return;
                }
                lower_bound_ = take(entry_);
            } else {
                lower_bound_ = maxDist();
            }
            visited_[entry_] = 1;
            candPush(lower_bound_, entry_);
        }
        if (timed_out(tctx_)) {
            *rc = VecSim_QueryReply_TimedOut;
            return;
        }
        fillFromExtras(top);
        if (top.size() == ef_) return;
        *rc = scanGraphInternal(top);
        if (top.size() < ef_) depleted_ = true;
    }

    HnswIndex *ix_;
    std::vector<char> query_;
    void *tctx_;
    size_t ef_ = 10, ahead_ = 8;
    size_t results_count_ = 0;
    bool depleted_ = false;
    uint32_t entry_ = INVALID;
    double lower_bound_ = std::numeric_limits<double>::infinity();
    std::vector<uint8_t> visited_;
    std::vector<Cand> candidates_;   // std::priority_queue's own algorithm over an open vector (the look-ahead reads its front)
    std::priority_queue<std::pair<double, size_t>, std::vector<std::pair<double, size_t>>, RefPairGreater> extras_;
    std::unordered_set<size_t> returned_;   // multi: labels handed out so far
    std::unordered_map<uint32_t, double> known_;
    size_t n_fetches_ = 0;
};

IterWalker *HnswIndex::newWalker(std::vector<char> query, VecSimQueryParams *qp) { return new HnswWalk(this, std::move(query), qp); }

}  // namespace vsa
