// hnsw_index.h -- host side of the HNSW index: the graph (levels, link lists, entry point), its
// construction, labels and deletion marks.  Queries run on the GPU (vsgpu_graph_search) over a device
// snapshot of this graph; every distance a *query* needs is evaluated by the gfx950 kernels.
//
// Graph construction is ingest-side host work, as in the reference ("graph build /
// delete / repair must exist host-side to produce a graph but are not GPU work"): insertion follows
// the reference's algorithm -- random level from the same generator (hnsw.h:418-422, seed 100),
// greedy descent, ef_construction-bounded layer search, neighbour selection by the diversity heuristic
// (hnsw.h:743-797), mutual linking with re-selection on overflow -- with its own host distance
// routine (hnsw_index.cpp:build_distance).  That routine is never reached from a query entry point.
#pragma once
#include <cstring>
#include <atomic>
#include <cstdint>
#include <memory>
#include <mutex>
#include <random>
#include <unordered_map>
#include <vector>

#include "flat_index.h"
#include "host_lane_eval.h"

namespace vsa {

class HnswIndex final : public VecSimIndexInterface {
public:
    static HnswIndex *create(const HNSWParams &p, void *logCtx);
    ~HnswIndex() override;

    int addVector(const void *blob, size_t label) override;
    int deleteVector(size_t label) override;
    size_t indexSize() const override { return n_ - n_deleted_; }
    size_t indexLabelCount() const override { return multi_ ? label_to_ids_.size() : label_to_id_.size(); }
    VecSimQueryReply *topKQuery(const void *query, size_t k, VecSimQueryParams *qp) override;
    int topKQueryBatch(const void *queries, size_t nq, size_t stride, size_t k, VecSimQueryParams *qp,
                       VecSimQueryReply_Order order, VecSimQueryReply **out) override;
    int topKCandidates(const void *, size_t, size_t, size_t, size_t, uint32_t *, size_t *, double *, uint32_t *) override {
        return -1;  // HNSW is replicas-only across GPUs
    }
    VecSimQueryReply *rangeQuery(const void *query, double radius, VecSimQueryParams *qp,
                                 VecSimQueryReply_Order order) override;
    double getDistanceFrom(size_t label, const void *blob) override;
    VecSimBatchIterator *newBatchIterator(const void *query, VecSimQueryParams *qp) override;
    int iteratorScores(const void *processed_query, std::vector<std::pair<double, size_t>> &out) override;
    vsgpu_scorebuf *iteratorDeviceBegin(const void *processed_query) override;
    int iteratorDeviceNext(vsgpu_scorebuf *b, size_t k, size_t cap, uint32_t *ids, double *scores, uint32_t *count) override;
    int iteratorDeviceRetire(vsgpu_scorebuf *b, const uint32_t *rows, size_t m) override;
    int iteratorDeviceRead(vsgpu_scorebuf *b, double *all) override;
    void iteratorDeviceEnd(vsgpu_scorebuf *b) override;
    size_t rowLabel(size_t id) const override { return (size_t)labels_[id]; }
    bool preferAdHocSearch(size_t subsetSize, size_t k, bool initial_check) override;
    VecSimIndexBasicInfo basicInfo() const override;
    VecSimIndexStatsInfo statsInfo() const override;
    VecSimIndexDebugInfo debugInfo() const override;
    long addBulk(const void *blobs, const size_t *labels, size_t n) override;
    long addSynthetic(size_t, uint64_t) override { return -1; }
    long storedVectors(size_t label, void *out, size_t cap_bytes) override;
    size_t storedBlobBytes() const override { return blob_bytes_; }
    // stored blobs of internal ids [first, first + n) (tests / tools: the graph's own row order, which compaction changes)
    int readRows(size_t first, size_t n, void *out) const {
        if (first + n > n_) return -1;
        std::memcpy(out, raw_.data() + first * blob_bytes_, n * blob_bytes_);
        return 0;
    }
    vsgpu_ctx *gpu() override { return ctx_; }
    int distanceTier() const override { return tier_; }
    void setLastMode(VecSearchMode m) override { last_mode_ = m; }

    // test / tooling access to the built graph (VecSimGpu_HnswExport)
    struct Export {
        uint32_t n, M, M0, entry;
        int max_level;
        const uint32_t *links0;
        const uint16_t *cnt0;
        const uint32_t *upper_off;
        const uint32_t *upper;
        size_t upper_words;
        const uint8_t *deleted;
        const uint64_t *labels;
    };
    Export exportGraph();
    uint64_t lastDistanceEvals() const { return last_dist_evals_.load(); }
    const uint8_t *levels() const { return level_.data(); }
    bool referenceOrderBuild() const { return ref_add_; }
    // labels of `label`'s neighbours, level by level (VecSimDebug_GetElementNeighborsInHNSWGraph); -1: unknown label, -2: multi-value
    int neighborLabels(size_t label, std::vector<std::vector<size_t>> &out);
    std::vector<vsgpu_ctx *> gpus() override;

    // the batch iterator's graph walk (hnsw_iter.cpp)
    IterWalker *newWalker(std::vector<char> processed_query, VecSimQueryParams *qp);

private:
    friend class HnswWalk;
    HnswIndex() = default;
    float buildDistance(const float *a, const float *b) const;
    const float *vec(uint32_t id) const { return host_vecs_.data() + (size_t)id * dim_; }
    // per-thread construction state (visited marks); `locked` = other threads are inserting too
    struct BuildCtx {
        std::vector<uint32_t> tag;
        uint32_t epoch = 0;
        bool locked = false;
        std::vector<uint32_t> scratch;
    };
    uint32_t *linksAt(uint32_t id, int level, uint32_t **count_word);
    uint32_t copyLinks(uint32_t id, int level, uint32_t *dst, bool locked);
    uint32_t allocNode(const char *stored_blob, size_t label, int level);
    void widen(const char *stored_blob, float *out) const;
    int drawLevel();
    // the reference's insert path with the tier's own distances, serial (hnsw_ref_build.cpp)
    void insertNodeRef(uint32_t id);
    double refDistance(uint32_t a, uint32_t b) const;
    void insertNode(uint32_t id, const float *v, BuildCtx &bc);
    void searchLayer(const float *q, uint32_t ep, float ep_dist, int level, size_t ef,
                     std::vector<std::pair<float, uint32_t>> &out, BuildCtx &bc);
    void selectNeighbors(std::vector<std::pair<float, uint32_t>> &cands, size_t M);
    void connect(uint32_t id, int level, const std::vector<std::pair<float, uint32_t>> &selected, bool locked);
    void lockNode(uint32_t id) {
        while (node_lock_[id].test_and_set(std::memory_order_acquire)) {
        }
    }
    void unlockNode(uint32_t id) { node_lock_[id].clear(std::memory_order_release); }
    int syncDevice();
    // Deleted nodes are marked at once (never returned, still traversed: hnsw.h:572-573) and REMOVED in batches -- the reference
    // repairs in place on every delete (hnsw.h:1796-1852) with per-node incoming-edge sets; this index keeps none, so finding who
    // points at a dead node is one pass over the graph, worth making once per batch of deletes: when a sixteenth of the nodes is
    // dead (and no batch iterator holds node ids), every live node that points at a dead one gets its list rebuilt from its live
    // neighbours and the dead ones' live neighbours (the reference's candidate set, repairConnectionsForDeletion hnsw.h:965-1060;
    // the build's own neighbour heuristic when they exceed the list), the entry point is replaced if it died, the last live nodes
    // move into the holes (swapWithLast, hnsw.h:1766-1786: rows, labels, links renumbered, the GPU table's rows moved), and the
    // arrays shrink.  So dead nodes cost searches at most ~6 % of their evaluations, not an ever-growing share.
    void maybeCompact();
    int compactDeleted();
    std::vector<char> preprocess(const void *blob) const;
    friend class HnswWalk;
    std::atomic<int> walkers_{0};   // live batch iterators (they hold node ids across calls: no compaction under them)

    VecSimType type_ = VecSimType_FLOAT32;
    VecSimMetric metric_ = VecSimMetric_L2;
    size_t dim_ = 0, block_size_ = DEFAULT_BLOCK_SIZE;
    size_t M_ = 16, M0_ = 32, ef_c_ = 200, ef_ = 10;
    double epsilon_ = 0.01, mult_ = 0;
    void *log_ctx_ = nullptr;
    std::default_random_engine level_gen_{100};

    // vectors: host copy for construction + device table for queries
    std::vector<float> host_vecs_;  // widened copy of every vector: construction-time distances only
    std::vector<char> raw_;         // stored (preprocessed) blobs in the index's own type: what goes to HBM
    size_t blob_bytes_ = 0, elem_bytes_ = 4;
    vsgpu_ctx *ctx_ = nullptr;
    int tier_ = 0;
    vsgpu_table *table_ = nullptr;
    vsgpu_graph *graph_ = nullptr;
    std::atomic<size_t> uploaded_rows_{0};   // (atomics: a reader looking for a lane peeks at these without the lock)
    std::atomic<bool> graph_dirty_{true};

    // graph
    size_t n_ = 0, n_deleted_ = 0;
    std::vector<uint32_t> links0_;     // [n][M0]
    std::vector<uint16_t> cnt0_;       // [n]
    std::vector<uint8_t> level_;       // [n]
    std::vector<uint32_t> upper_off_;  // [n] block index or 0xFFFFFFFF
    std::vector<uint32_t> upper_;      // blocks of 1+M words
    std::vector<uint8_t> deleted_;
    std::vector<uint64_t> labels_;
    std::unordered_map<size_t, uint32_t> label_to_id_;
    // multi-value index (hnsw_multi.h:16-247): a label owns any number of nodes; adds never overwrite, a delete marks them all
    bool multi_ = false;
    std::unordered_map<size_t, std::vector<uint32_t>> label_to_ids_;
    uint32_t entry_ = 0xFFFFFFFFu;
    int max_level_ = -1;
    // Build modes ($VECSIM_GPU_HNSW_BUILD).  VecSimIndex_AddVector follows the REFERENCE's insert path -- same level generator,
    // same candidate order, distances in the tier's own summation order (host_lane_eval.h) -- whenever the tier has a host walker
    // ("reference"; the default), so that N AddVector calls leave the graph the reference would hold; "fast" = the round 1-5
    // builder (any-order AVX-512 distances on widened rows).  The bulk entry point builds in parallel with the fast routine
    // unless the variable says "reference" (then serial, reference order).
    HostLaneEval ref_eval_;
    bool ref_add_ = false, ref_bulk_ = false;
    BuildCtx main_ctx_;                               // single-threaded inserts
    std::unique_ptr<std::atomic_flag[]> node_lock_;  // per-node link-list locks (parallel bulk build only)
    size_t node_lock_n_ = 0;
    std::mutex entry_mu_;

    // the index's own GPU context (staging buffers, stream) plus reader lanes: a reader that finds it busy searches through a
    // view of the same snapshot -- own stream, query staging, visited tags and result buffers (vsgpu_graph_view_create) -- so one
    // reader's query staging, download and reply construction overlap with the other's search kernel (the reference lets any
    // number of readers in: bindings.cpp:250-283)
    struct Lane {
        std::mutex mu;
        vsgpu_ctx *ctx = nullptr;
        vsgpu_table *view = nullptr;
        vsgpu_graph *graph = nullptr;
    };
    std::vector<std::unique_ptr<Lane>> lanes_;
    Lane *tryLane();
    mutable std::recursive_mutex gpu_mu_;
    std::atomic<uint64_t> last_dist_evals_{0};
    mutable VecSearchMode last_mode_ = EMPTY_MODE;
};

}  // namespace vsa
