// ref_heap_probe.cpp -- the product's heap comparators / containers (vectorsimilarity_amd/csrc/host/ref_heap.h: what the Flat replay,
// the sharded merge and the HNSW batch iterator's walk keep their state in) driven by a script of operations, for
// tests/test_ref_fixture.py: the same script ran through the REFERENCE's containers (oracle/ref_driver.cpp:vsref_heap_script) when the
// fixture tests/golden/ref_heap_scripts.json was made.  Built as C++17, as the product is.
#include <cstddef>
#include <limits>
#include <queue>

#include "../../vectorsimilarity_amd/csrc/host/ref_heap.h"

extern "C" void heap_script(int kind, const int *op, const double *score, const size_t *label, size_t n, size_t *out_size,
                            double *out_top_score, size_t *out_top_label) {
    using Item = std::pair<double, size_t>;
    std::priority_queue<Item, std::vector<Item>, vsa::RefPairGreater> mn;
    vsa::RefMaxHeap<> mx;
    vsa::RefUpdatableMaxHeap up;
    for (size_t i = 0; i < n; i++) {
        if (op[i] == 0) {
            if (kind == 0) mn.emplace(score[i], label[i]);
            else if (kind == 1) mx.emplace(score[i], label[i]);
            else up.emplace(score[i], label[i]);
        } else {
            if (kind == 0) { if (!mn.empty()) mn.pop(); }
            else if (kind == 1) { if (!mx.empty()) mx.pop(); }
            else { if (!up.empty()) up.pop(); }
        }
        const size_t sz = kind == 0 ? mn.size() : (kind == 1 ? mx.size() : up.size());
        out_size[i] = sz;
        if (sz == 0) {
            out_top_score[i] = std::numeric_limits<double>::quiet_NaN();
            out_top_label[i] = ~(size_t)0;
        } else if (kind == 0) {
            out_top_score[i] = mn.top().first; out_top_label[i] = mn.top().second;
        } else if (kind == 1) {
            out_top_score[i] = mx.top().first; out_top_label[i] = mx.top().second;
        } else {
            out_top_score[i] = up.top().first; out_top_label[i] = up.top().second;
        }
    }
}
