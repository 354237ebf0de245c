// vso_stdsort.cpp -- TEST INFRASTRUCTURE (oracle).  The one library-defined step of the reference's HNSW insert path:
// getNeighborsByHeuristic2_internal sorts its candidates with
//     std::sort(begin, end, [](const auto &a, const auto &b) { return a.first < b.first; })        (hnsw.h:763-764)
// -- by distance only, so the order of EXACTLY tied distances is whatever libstdc++'s introsort leaves.  oracle/vso_hnsw.c calls
// this shim for that step, i.e. the same library routine on the same sequence with the same comparator, instead of restating
// introsort in C: a reference built with this toolchain orders ties exactly like this.
#include <algorithm>
#include <cstddef>
#include <cstdint>

struct vso_pair {   // layout of pr_t in vso_hnsw.c
    double d;
    uint32_t id;
};
extern "C" void vso_std_sort_by_distance(vso_pair *v, size_t n) {
    std::sort(v, v + n, [](const vso_pair &a, const vso_pair &b) { return a.d < b.d; });
}
