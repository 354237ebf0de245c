// flat_index.h -- host side of the Flat (brute-force) index: labels, block bookkeeping, blob
// preprocessing, reply construction.  Vector bytes live in HBM behind vsgpu_table; every distance
// comes back from the gfx950 kernels through include/vsgpu.h.
//
// Mirrors the behaviour of the reference's BruteForceIndex / BruteForceIndex_Single
// (algorithms/brute_force/brute_force.h:175-326, brute_force_single.h:135-212) for the calls on the
// hot path and the calls either side of it.
#pragma once
#include <cstddef>
#include <atomic>
#include <cstdint>
#include <limits>
#include <memory>
#include <unordered_map>
#include <utility>
#include <mutex>
#include <set>
#include <vector>

#include "VecSim/vec_sim.h"
#include "VecSim/vec_sim_gpu.h"
#include "vsgpu.h"

// Reply objects (reference: query_result_definitions.h:25-39)
struct VecSimQueryResult {
    size_t id;
    double score;
};
struct VecSimQueryReply {
    std::vector<VecSimQueryResult> results;
    VecSimQueryReply_Code code = VecSim_QueryReply_OK;
};
struct VecSimQueryReply_Iterator {
    VecSimQueryReply *reply;
    size_t pos;
};

namespace vsa {
struct Globals {
    timeoutCallbackFunction timeout_cb = nullptr;  // default: never times out (vec_sim_interface.cpp:78)
    logCallbackFunction log_cb = nullptr;
    VecSimWriteMode write_mode = VecSim_WriteAsync;
    int device = -1;  // -1: $VECSIM_GPU_DEVICE or 0
};
Globals &globals();
inline bool timed_out(void *ctx) {
    timeoutCallbackFunction cb = globals().timeout_cb;
    return cb && cb(ctx) != 0;
}
void sort_reply(VecSimQueryReply *rep, VecSimQueryReply_Order order);
}  // namespace vsa

// The C API's opaque index type
struct VecSimIndexInterface {
    virtual ~VecSimIndexInterface() = default;
    virtual int addVector(const void *blob, size_t label) = 0;
    virtual int deleteVector(size_t label) = 0;
    virtual size_t indexSize() const = 0;
    virtual size_t indexLabelCount() const = 0;
    virtual VecSimQueryReply *topKQuery(const void *query, size_t k, VecSimQueryParams *qp) = 0;
    virtual int topKQueryBatch(const void *queries, size_t nq, size_t stride, size_t k, VecSimQueryParams *qp,
                               VecSimQueryReply_Order order, VecSimQueryReply **out) = 0;
    virtual int topKCandidates(const void *queries, size_t nq, size_t stride, size_t k, size_t cap, uint32_t *ids,
                               size_t *labels, double *scores, uint32_t *counts) = 0;
    virtual VecSimQueryReply *rangeQuery(const void *query, double radius, VecSimQueryParams *qp,
                                         VecSimQueryReply_Order order) = 0;
    virtual double getDistanceFrom(size_t label, const void *blob) = 0;
    virtual VecSimBatchIterator *newBatchIterator(const void *query, VecSimQueryParams *qp) = 0;
    virtual bool preferAdHocSearch(size_t subsetSize, size_t k, bool initial_check) = 0;
    virtual VecSimIndexBasicInfo basicInfo() const = 0;
    virtual VecSimIndexStatsInfo statsInfo() const = 0;
    virtual VecSimIndexDebugInfo debugInfo() const = 0;
    virtual long addBulk(const void *blobs, const size_t *labels, size_t n) = 0;
    virtual long addSynthetic(size_t n, uint64_t seed) = 0;
    // stored blobs of a label, in internal-id order (bindings.cpp get_vector); returns the vector count, -1 on error
    virtual long storedVectors(size_t label, void *out, size_t cap_bytes) = 0;
    virtual size_t storedBlobBytes() const = 0;
    // batch iterator: (score, label) of every live label against a processed query, one GPU score pass
    virtual int iteratorScores(const void *processed_query, std::vector<std::pair<double, size_t>> &out) = 0;
    // device-resident iterator state (Flat single-value only; others return nullptr and use iteratorScores)
    virtual vsgpu_scorebuf *iteratorDeviceBegin(const void *) { return nullptr; }
    virtual int iteratorDeviceNext(vsgpu_scorebuf *, size_t, size_t, uint32_t *, double *, uint32_t *) { return -1; }
    virtual int iteratorDeviceRetire(vsgpu_scorebuf *, const uint32_t *, size_t) { return -1; }
    virtual int iteratorDeviceRead(vsgpu_scorebuf *, double *) { return -1; }
    virtual void iteratorDeviceEnd(vsgpu_scorebuf *) {}
    virtual size_t rowLabel(size_t) const { return 0; }
    virtual vsgpu_ctx *gpu() = 0;
    // every GPU context of the index (reader lanes included): options go to all, statistics are summed over them
    virtual std::vector<vsgpu_ctx *> gpus() { return {gpu()}; }
    virtual void setLastMode(VecSearchMode m) = 0;
    virtual int distanceTier() const { return VSGPU_TIER_AVX512; }   // which reference ISA tier's order the scores follow (host_tier.h)
    // Lifetime (c_api.cpp): the reference lets a batch iterator outlive its index (it keeps the allocator alive: query_results.cpp:77-82;
    // only freeing it is legal then).  Here an iterator holds device state and node ids of its index, so VecSimIndex_Free of an index
    // with live iterators only marks it; the last VecSimBatchIterator_Free destroys it.
    int live_iterators_ = 0;
    bool orphaned_ = false;
};

namespace vsa {

class FlatIndex final : public VecSimIndexInterface {
public:
    // returns nullptr (with VecSimGpu_LastError set) when the GPU context cannot be created
    static FlatIndex *create(const BFParams &p, void *logCtx);
    // SQ8 storage (types/sq8.h, QuantPreprocessor): callers add and query fp32 vectors, rows are uint8 codes + FP32
    // metadata, every distance is the reference's asymmetric SQ8 x FP32 kernel.  p.type must be FLOAT32.
    // mean != nullptr: mean-centred blobs (QuantPreprocessor<..., WithNorm = true>, L2 and IP only); mean_sum_squares is the
    // constant DistanceCalculatorWithNorm takes for its symmetric IP correction (calculator.h:204-214)
    static FlatIndex *createSQ8(const BFParams &p, void *logCtx, const float *mean = nullptr, float mean_sum_squares = 0.0f);
    bool isSQ8() const { return sq8_; }
    // symmetric SQ8 x SQ8 distance between the stored vectors of two labels (NaN for an unknown label)
    double storedDistance(size_t label_a, size_t label_b);
    ~FlatIndex() override;

    int addVector(const void *blob, size_t label) override;
    int deleteVector(size_t label) override;
    size_t indexSize() const override { return count_; }
    size_t indexLabelCount() const override { return multi_ ? label_to_ids_.size() : count_; }
    VecSimQueryReply *topKQuery(const void *query, size_t k, VecSimQueryParams *qp) override;
    int topKQueryBatch(const void *queries, size_t nq, size_t stride, size_t k, VecSimQueryParams *qp,
                       VecSimQueryReply_Order order, VecSimQueryReply **out) override;
    int topKCandidates(const void *queries, size_t nq, size_t stride, size_t k, size_t cap, uint32_t *ids,
                       size_t *labels, double *scores, uint32_t *counts) override;
    VecSimQueryReply *rangeQuery(const void *query, double radius, VecSimQueryParams *qp,
                                 VecSimQueryReply_Order order) override;
    double getDistanceFrom(size_t label, const void *blob) override;
    VecSimBatchIterator *newBatchIterator(const void *query, VecSimQueryParams *qp) override;
    bool preferAdHocSearch(size_t subsetSize, size_t k, bool initial_check) override;
    VecSimIndexBasicInfo basicInfo() const override;
    VecSimIndexStatsInfo statsInfo() const override;
    VecSimIndexDebugInfo debugInfo() const override;
    long addBulk(const void *blobs, const size_t *labels, size_t n) override;
    long addSynthetic(size_t n, uint64_t seed) override;
    long storedVectors(size_t label, void *out, size_t cap_bytes) override;
    size_t storedBlobBytes() const override { return stored_bytes_; }
    vsgpu_ctx *gpu() override { return ctx_; }
    int distanceTier() const override { return tier_; }
    std::vector<vsgpu_ctx *> gpus() override;
    void setLastMode(VecSearchMode m) override { last_mode_ = m; }

    int iteratorScores(const void *processed_query, std::vector<std::pair<double, size_t>> &out) override;
    vsgpu_scorebuf *iteratorDeviceBegin(const void *processed_query) override;
    int iteratorDeviceNext(vsgpu_scorebuf *b, size_t k, size_t cap, uint32_t *ids, double *scores, uint32_t *count) override;
    int iteratorDeviceRetire(vsgpu_scorebuf *b, const uint32_t *rows, size_t m) override;
    int iteratorDeviceRead(vsgpu_scorebuf *b, double *all) override;
    void iteratorDeviceEnd(vsgpu_scorebuf *b) override;
    size_t rowLabel(size_t id) const override { return id_to_label_[id]; }
    int allScores(const void *processed_query, std::vector<double> &scores);
    // NaN bookkeeping for the sharded index: the smallest id of a row that can score NaN (SIZE_MAX: none) and whether a raw
    // query can (NaN / Inf elements, a zero vector under Cosine): such replies replay the reference's heap over every row
    size_t firstNanRow() const { return nan_ids_.empty() ? (size_t)-1 : (size_t)*nan_ids_.begin(); }
    bool queryMayScoreNaN(const void *raw_query) const;
    // row-level operations for the sharded index (sharded_index.cpp), which replays the equivalent single index's
    // swap-delete (brute_force.h:196-224) across shards: the global last row moves into the hole
    int readRows(uint32_t first, size_t n, void *stored_blobs);                      // rows [first, first + n) by internal id
    void forgetIdOfLabel(size_t label, uint32_t id);   // multi-value: drop one id from a label's list
    int readRow(uint32_t id, void *stored_blob);                                   // stored (preprocessed) bytes of a row
    int overwriteRow(uint32_t id, const void *stored_blob, size_t new_label);     // raw stored bytes, relabelled
    int dropLastRow();                                                             // forget the shard's last row
    size_t labelOf(size_t id) const { return id_to_label_[id]; }
    bool isMulti() const { return multi_; }
    size_t queryBytes() const { return query_bytes_; }
    std::vector<char> preprocessQuery(const void *query) const;

private:
    FlatIndex() = default;
    int flush();  // push host-staged rows to the device table
    void stageRow(const void *processed);
    int appendStored(const void *stored, size_t label);
    void log(const char *level, const char *fmt, ...) const;
    std::vector<char> packQueries(const void *queries, size_t nq, size_t stride) const;
    void replay(const uint32_t *ids, const double *scores, size_t n, size_t k, VecSimQueryReply *rep) const;
    void replayMulti(const uint32_t *ids, const double *scores, size_t n, size_t k, VecSimQueryReply *rep) const;
    size_t distinctLabels(const uint32_t *ids, size_t n) const;
    void removeRow(uint32_t id);

    VecSimType type_ = VecSimType_FLOAT32;
    int tier_ = 0;
    VecSimMetric metric_ = VecSimMetric_L2;
    size_t dim_ = 0, block_size_ = DEFAULT_BLOCK_SIZE;
    size_t stored_bytes_ = 0, query_bytes_ = 0;
    void *log_ctx_ = nullptr;
    vsgpu_ctx *ctx_ = nullptr;
    vsgpu_table *table_ = nullptr;
    size_t count_ = 0;  // vectors in the index (device rows + staged rows)
    std::vector<size_t> id_to_label_;
    std::unordered_map<size_t, uint32_t> label_to_id_;
    // multi-value index (brute_force_multi.h): a label owns any number of vectors
    // one GPU context (staging buffers, stream) per index: concurrent readers take turns
    mutable std::recursive_mutex gpu_mu_;
    bool multi_ = false;
    bool sq8_ = false;
    std::vector<float> sq8_mean_;   // empty: plain SQ8
    float sq8_mss_ = 0.0f;
    // fp32 vector (dim_ floats) -> stored blob / query blob of this index (Cosine: normalised first; SQ8: quantised)
    void toStored(const void *blob, char *out) const;
    void toQuery(const void *query, char *out) const;
    std::unordered_map<size_t, std::vector<uint32_t>> label_to_ids_;
    std::vector<char> staged_;  // rows appended but not yet uploaded
    // Rows whose score may be NaN for some query (NaN / Inf / huge elements, a Cosine zero vector).  The reference's loop
    // (brute_force.h:272) lets a NaN score in only while its heap is not full, i.e. for rows with id < k; a query that can
    // meet such a row -- or is such a vector itself -- replays the sequential heap over every row's score instead of over
    // the GPU's candidate set, so that even those replies equal the reference's.  Everything else is unaffected.
    std::set<uint32_t> nan_ids_;
    bool mayScoreNaN(const char *stored_or_query) const;
    void noteRow(uint32_t id, const void *stored);
    std::atomic<size_t> staged_rows_{0};
    // Reader lanes: the reference lets any number of readers query one index at a time (vec_sim.h threading contract,
    // bindings.cpp:250-283).  The first reader uses the index's own context; a reader that finds it busy takes a lane --
    // another context (stream + scratch) over a view of the same rows (vsgpu_table_view_create) -- so its query upload,
    // probe, re-rank, download and host replay overlap with the other reader's scan kernel.  VECSIM_GPU_READER_LANES
    // (default 2) counts the contexts; writers are the caller's to keep out, as upstream.
    struct Lane {
        vsgpu_ctx *ctx = nullptr;
        vsgpu_table *view = nullptr;
        std::mutex mu;
    };
    std::vector<std::unique_ptr<Lane>> lanes_;
    Lane *tryLane();
    mutable VecSearchMode last_mode_ = EMPTY_MODE;
};

}  // namespace vsa

namespace vsa {
// an iterator that is not "the next n best of one score pass": the HNSW index's graph walk (hnsw_batch_iterator.h:96-230)
struct IterWalker {
    virtual ~IterWalker() = default;
    virtual VecSimQueryReply *next(size_t n_res, VecSimQueryReply_Order order) = 0;
    virtual bool depleted() const = 0;
    virtual void reset() = 0;
};
}  // namespace vsa

// "next n best" cursor (reference: batch_iterator.h, brute_force/bf_batch_iterator.h:24-199)
struct VecSimBatchIterator {
    VecSimIndexInterface *index;
    std::unique_ptr<vsa::IterWalker> walker;   // set: Next / HasNext / Reset are the walker's
    // sparse mode: scores stay on the device, the host only tracks where the reference's array compaction
    // (bf_batch_iterator.h: returned entries are swapped out of the live range) has moved entries
    vsgpu_scorebuf *dev = nullptr;
    bool dev_tried = false;
    size_t dev_rows = 0;
    std::unordered_map<uint32_t, size_t> moved_to;  // row -> current array position (absent: position == row)
    std::unordered_map<size_t, uint32_t> moved_at;  // array position -> row sitting there (absent: the row == position)
    std::vector<char> query;  // processed query, owned
    void *timeout_ctx;
    std::vector<std::pair<double, size_t>> scores;  // (score, label) of every vector, lazily filled
    bool scored = false;
    size_t valid_start = 0;
    size_t label_count = 0;
    size_t returned = 0;
};
