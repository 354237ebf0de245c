// ref_heap.h -- the order of the reference's top-k containers, spelled out.
//
// vecsim_stl::max_priority_queue (utils/vecsim_stl.h:63-83) is std::priority_queue over std::less<std::pair<DistType,
// labelType>>, and the reference is built as gnu++20 (src/VecSim/CMakeLists.txt:15): pair's `<` is synthesised from
// operator<=> there.  On ordinary scores that is the familiar lexicographic order.  On a NaN score it is not what C++17's
// operator< did: `a.first <=> b.first` is partial_ordering::unordered, the pair comparison stops there, and NEITHER pair
// is less than the other -- C++17 fell through to the labels.  Which rows a reply holds once NaN scores sit in the heap
// depends on exactly this (oracle/_ref runs the reference's own container: tests/golden/ref_scalar_random.json, section
// `topk`).  This library is built as C++17, so the rule is written out instead of inherited from the standard in force.
#pragma once
#include <cstddef>
#include <functional>
#include <map>
#include <queue>
#include <unordered_map>
#include <utility>
#include <vector>

namespace vsa {

// -1 less, 0 equivalent, +1 greater, 2 unordered (a NaN on either side)
inline int score_cmp3(double a, double b) { return a < b ? -1 : (b < a ? 1 : (a == b ? 0 : 2)); }

struct RefPairLess {
    bool operator()(const std::pair<double, size_t> &a, const std::pair<double, size_t> &b) const {
        const int c = score_cmp3(a.first, b.first);
        return c != 0 ? c < 0 : a.second < b.second;
    }
    // ((score, label), tag): the batch iterator keeps an index beside the reference's pair
    bool operator()(const std::pair<std::pair<double, size_t>, size_t> &a,
                    const std::pair<std::pair<double, size_t>, size_t> &b) const {
        const int c = score_cmp3(a.first.first, b.first.first);
        if (c != 0) return c < 0;
        if (a.first.second != b.first.second) return a.first.second < b.first.second;
        return a.second < b.second;
    }
};

template <typename Item = std::pair<double, size_t>>
using RefMaxHeap = std::priority_queue<Item, std::vector<Item>, RefPairLess>;

// vecsim_stl::min_priority_queue (utils/vecsim_stl.h:85-105): std::priority_queue over std::greater<pair>, i.e. `b < a` in the
// order above -- the smallest (score, id) on top
struct RefPairGreater {
    template <typename P> bool operator()(const P &a, const P &b) const {
        const int c = score_cmp3(b.first, a.first);
        return c != 0 ? c < 0 : b.second < a.second;
    }
};

// vecsim_stl::updatable_max_heap (utils/updatable_heap.h:20-113): one entry per label, a label keeps the LOWEST score it was
// offered; top() = the largest score, among equal scores the largest label.  Scores in a multimap ordered by std::greater as
// upstream (a NaN key breaks that container's strict weak order there as here; multi-value indexes never fill their heap with NaN
// in the reference's tests, and none of ours depends on it).
class RefUpdatableMaxHeap {
public:
    using Item = std::pair<double, size_t>;
    bool empty() const { return by_score_.empty(); }
    size_t size() const { return by_score_.size(); }
    Item top() const {
        auto t = topIt();
        return {t->first, t->second};
    }
    void pop() {
        auto t = topIt();
        node_of_.erase(t->second);
        by_score_.erase(t);
    }
    void emplace(double score, size_t label) {
        auto f = node_of_.find(label);
        if (f == node_of_.end()) node_of_.emplace(label, by_score_.emplace(score, label));
        else if (f->second->first > score) {
            by_score_.erase(f->second);
            f->second = by_score_.emplace(score, label);
        }
    }

private:
    using Map = std::multimap<double, size_t, std::greater<double>>;
    Map::const_iterator topIt() const {
        auto rng = by_score_.equal_range(by_score_.begin()->first);
        auto best = rng.first;
        for (auto i = rng.first; i != rng.second; ++i)
            if (best->second < i->second) best = i;
        return best;
    }
    Map by_score_;
    std::unordered_map<size_t, Map::iterator> node_of_;
};

}  // namespace vsa
