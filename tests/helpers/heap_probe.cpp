// heap_probe.cpp -- test helper: the sequential top-k loop of brute_force.h:257-288 over a score array, on the REAL
// std::priority_queue<std::pair<score, label>> of this toolchain's libstdc++ (what utils/vecsim_stl.h:66-72 wraps).
// Compiled as gnu++20 LIKE THE REFERENCE (src/VecSim/CMakeLists.txt:15): pair's `<` comes from operator<=> there and treats NaN
// scores as unordered, where C++17 compared the labels.  tests/test_oracle_kats.py uses it to pin oracle/vso.c's restated heap moves, which matter once scores hold NaNs.
#include <cstddef>
#include <limits>
#include <queue>
#include <utility>

extern "C" size_t heap_probe_topk(const double *scores, const size_t *labels, size_t n, size_t k, size_t *out_labels,
                                  double *out_scores) {
    if (k == 0) return 0;
    std::priority_queue<std::pair<double, size_t>> heap;
    double upper = std::numeric_limits<double>::lowest();
    for (size_t i = 0; i < n; i++) {
        const double s = scores[i];
        if (s < upper || heap.size() < k) {
            heap.emplace(s, labels ? labels[i] : i);
            if (heap.size() > k) heap.pop();
            upper = heap.top().first;
        }
    }
    const size_t cnt = heap.size();
    for (size_t i = cnt; i-- > 0;) {
        out_scores[i] = heap.top().first;
        out_labels[i] = heap.top().second;
        heap.pop();
    }
    return cnt;
}
