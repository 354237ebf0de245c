// sharded_index.h -- a Flat index whose vector blocks are dealt over G shards (one GPU each), answering
// exactly what the equivalent single BruteForceIndex would.
//
// Partition: vector number i of the equivalent single index (its internal id there, `gid`) lives in block
// b = i / blockSize; block b belongs to shard b % G at local id (b / G) * blockSize + i % blockSize.  Every
// process sees every ingest call (SPMD) and keeps only its own shards' rows, so ingest moves no data.
// Query: each shard returns EVERY local row with score <= its local k-th score (FlatIndex::topKCandidates,
// GPU scan), the fixed-size records are exchanged once (RCCL all-gather over xGMI between processes; nothing
// when one process drives all shards), and the reference's sequential heap (brute_force.h:257-288) is replayed
// over the union in gid order -- ties included, the reply equals the single index's.
// Delete: the single index moves its LAST row into the hole (brute_force.h:196-224); here the owner of the
// last gid broadcasts that row and the owner of the hole overwrites, so gids keep matching.
#pragma once
#include <cstddef>
#include <cstdint>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "flat_index.h"

namespace vsa {

struct ShardPlan {
    size_t block = 1024;
    size_t world = 1;
    size_t owner(uint64_t gid) const { return (gid / block) % world; }
    uint64_t local(uint64_t gid) const { return ((gid / block) / world) * block + gid % block; }
    uint64_t gid(uint64_t local_id, size_t shard) const { return ((local_id / block) * world + shard) * block + local_id % block; }
};

// transport between the processes that hold the shards
struct Exchange {
    virtual ~Exchange() = default;
    virtual int allgather(const void *send, size_t bytes, void *recv) = 0;  // recv: world * bytes, rank order
    virtual int broadcast(void *buf, size_t bytes, int root) = 0;
    virtual bool canBroadcast() const { return true; }
    // gives up on the peers: collectives in flight on this process come back with an error, later ones are refused (RCCL:
    // ncclCommAbort; a caller-provided transport has its own means)
    virtual void abort() {}
    virtual const char *mode() const { return "transport"; }
};

// what the sharded index needs from one shard; FlatIndex (GPU) in the product, caller-provided for external shards
struct ShardOps {
    virtual ~ShardOps() = default;
    virtual int add(const void *blob, size_t label) = 0;  // new label: append (1); existing label: overwrite (0)
    virtual int candidates(const void *queries, size_t nq, size_t stride, size_t k, size_t cap, uint32_t *ids,
                           size_t *labels, double *scores, uint32_t *counts) = 0;
    virtual size_t size() const = 0;
    virtual size_t storedBytes() const = 0;
    // swap-delete needs to read / overwrite / drop rows (FlatIndex can; an external append-only shard cannot)
    virtual bool supportsRowOps() const { return false; }
    // smallest local id of a row that can score NaN (FlatIndex::nan_ids_), SIZE_MAX if none
    virtual size_t firstNanRow() const { return (size_t)-1; }
    // 1 if this query can score NaN against finite rows (NaN / Inf elements, a zero vector under Cosine ...)
    virtual bool queryMayScoreNaN(const void *) const { return false; }
    // every row's score for ONE query: ids[i] = i, scores[i], i < size() (the NaN-aware replay scans all rows)
    virtual int allScores(const void *, uint32_t *, size_t *, double *) { return -1; }
    virtual int readRow(uint32_t, void *) { return -1; }
    virtual int overwriteRow(uint32_t, const void *, size_t) { return -1; }
    virtual int dropLastRow() { return -1; }
    virtual long addSynthetic(size_t, uint64_t) { return -1; }
    virtual VecSimIndexInterface *index() { return nullptr; }
};

class ShardedIndex {
public:
    // one process per GPU: this process holds shard `rank`; `ex` moves records between the processes
    static ShardedIndex *createDistributed(const BFParams &p, void *logCtx, int rank, int world, int device,
                                           std::unique_ptr<Exchange> ex, std::unique_ptr<ShardOps> external = nullptr);
    // one process drives all shards (devices[i] may repeat: several shards on one GPU)
    static ShardedIndex *createLocal(const BFParams &p, void *logCtx, int n_shards, const int *devices);
    ~ShardedIndex();

    int addVector(const void *blob, size_t label);
    long addBulk(const void *blobs, const size_t *labels, size_t n);
    long addSyntheticLocal(size_t rows_per_shard, uint64_t seed_base);
    int deleteVector(size_t label);
    size_t indexSize() const { return n_global_; }
    // seq: position of this batch in the stream of batches every process answers (0, 1, 2 ...; NO_SEQ = one reader, call
    // order is the order).  Several reader threads may be inside at once: their scans overlap on the shard's reader lanes,
    // their exchanges are issued strictly in seq order on every process, so the collectives pair up whatever the threads'
    // relative speed -- batch i's exchange and merge run under batch i+1's scan.
    static constexpr uint64_t NO_SEQ = ~0ull;
    int topKQueryBatch(const void *queries, size_t nq, size_t stride, size_t k, VecSimQueryParams *qp,
                       VecSimQueryReply_Order order, VecSimQueryReply **out, uint64_t seq = NO_SEQ);
    // accumulated wall time per phase since the last reset, in ms: {scan (shard candidates), wait for the exchange turn,
    // exchange, merge + replies, batches, exchanged bytes per rank}
    void stats(double out[6]);
    void resetStats();
    void resetSeq();   // the next numbered batch is 0 again (no batch may be in flight)
    VecSimIndexInterface *localIndex(int shard);
    void setExchange(std::unique_ptr<Exchange> ex) { ex_ = std::move(ex); }
    int world() const { return (int)plan_.world; }
    int rank() const { return rank_; }
    void abortExchange() {
        if (ex_) ex_->abort();
    }
    const char *exchangeMode() const { return ex_ ? ex_->mode() : "local"; }
    // every process all-gathers `bytes` rank-stamped bytes and checks every rank's slice; also agrees on the verdict (a second
    // 8-byte all-gather), so every process returns the same value: 0 = the exchange moves bytes correctly between all ranks
    int exchangeSelfTest(size_t bytes);
    // the exchange + merge step on its own (also the body of topKQueryBatch): `mine` = this process's records
    static size_t recordBytes(size_t nq, size_t cap) { return 32 + nq * 8 * (1 + 3 * cap); }

private:
    ShardedIndex() = default;
    bool owns(size_t shard) const { return rank_ < 0 || (size_t)rank_ == shard; }
    ShardOps *shard(size_t s) { return shards_[rank_ < 0 ? s : 0].get(); }
    uint64_t gidOf(uint64_t local_id, size_t shard) const {
        return synthetic_rows_ ? shard * synthetic_rows_ + local_id : plan_.gid(local_id, shard);
    }
    struct Pass {   // what one exchange tells every process identically
        bool overflow = false, timed_out = false, nan_rows_at_head = false;
        bool more_follows = false;   // in: the caller will exchange again for this batch whatever this pass finds
        std::vector<char> needs_all;   // per query: a NaN-capable query
    };
    int queryOnce(const void *queries, size_t nq, size_t stride, size_t k, size_t cap, bool local_timeout, bool all_rows,
                  const std::function<bool()> &poll_timeout, const std::function<void(bool)> &turn_hook,
                  std::vector<size_t> &out_labels, std::vector<double> &out_scores, std::vector<uint32_t> &out_counts, Pass *pass);
    bool takeTurn(uint64_t seq);   // false: the number has already passed
    void passTurn(uint64_t seq);

    BFParams params_{};
    ShardPlan plan_;
    int rank_ = -1;  // -1: local mode (all shards here)
    std::vector<std::unique_ptr<ShardOps>> shards_;
    std::unique_ptr<Exchange> ex_;
    size_t n_global_ = 0;
    size_t synthetic_rows_ = 0;  // > 0: filled by addSyntheticLocal (gid = shard * rows + local id, label = gid), append-only
    std::unordered_map<size_t, uint64_t> label_to_gid_;
    std::unordered_map<size_t, std::vector<uint64_t>> label_to_gids_;   // multi-value: a label's rows, in the reference's list order
    int removeGid(uint64_t hole);   // one swap-delete of the equivalent single index, across the shards
    long test_fail_remove_at_ = -1, test_removes_ = 0;   // $VECSIM_GPU_TEST_FAIL_REMOVE_AT (removeGid)
    std::vector<size_t> gid_to_label_;
    // exchange turns (seq order) and phase timers
    std::mutex turn_mu_;
    std::condition_variable turn_cv_;
    uint64_t next_seq_ = 0;
    bool turn_held_ = false;
    std::mutex stats_mu_;
    double st_scan_ = 0, st_wait_ = 0, st_exchange_ = 0, st_merge_ = 0, st_batches_ = 0, st_bytes_ = 0;
};

// Merge side of a sharded query: `parts` candidate lists (layout [part][nq][cap], counts [part][nq], gids = the row's
// internal id in the equivalent single index) -> per query the reply of the reference's sequential heap over the union
// scanned in gid order: keep score <= T (k-th smallest of the union), sort by gid, replay brute_force.h:264-281.
// out_labels/out_scores are [nq][k], out_counts[q] results written.  Returns -1 if any count is the overflow marker.
// every_row: the lists hold EVERY row (the NaN-aware replay: a NaN score enters the reference's heap only while it fills,
// brute_force.h:272): no threshold, the heap loop runs over all of them in gid order.
int merge_topk(size_t nq, size_t parts, size_t cap, const uint64_t *gids, const size_t *labels, const double *scores,
               const uint32_t *counts, size_t k, size_t *out_labels, double *out_scores, uint32_t *out_counts,
               bool every_row = false);
// the same over parts whose arrays lie part_stride 8-byte words apart (the exchange records as the collective wrote them); wide and
// deep batches are merged on a few host threads (queries are independent)
int merge_topk_strided(size_t nq, size_t parts, size_t cap, size_t part_stride, const uint64_t *gids, const size_t *labels,
                       const double *scores, const uint32_t *counts, size_t k, size_t *out_labels, double *out_scores,
                       uint32_t *out_counts, bool every_row = false);
int merge_topk_multi_strided(size_t nq, size_t parts, size_t cap, size_t part_stride, const uint64_t *gids, const size_t *labels,
                             const double *scores, const uint32_t *counts, size_t k, size_t *out_labels, double *out_scores,
                             uint32_t *out_counts);
std::unique_ptr<Exchange> make_rccl_exchange(vsgpu_ctx *ctx, int rank, int world, const void *id128);
}  // namespace vsa

// the C API's opaque type (VecSim/vec_sim_gpu.h)
struct VecSimShardedIndex {
    std::unique_ptr<vsa::ShardedIndex> impl;
};
