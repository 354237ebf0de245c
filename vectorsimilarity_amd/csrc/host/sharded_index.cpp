// sharded_index.cpp -- see sharded_index.h
#include "sharded_index.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <queue>
#include <thread>

#include "blob_prep.h"
#include "ref_heap.h"

namespace vsa {

namespace {
// RCCL over xGMI (include/vsgpu.h comm group)
struct RcclExchange final : Exchange {
    vsgpu_comm *comm = nullptr;
    ~RcclExchange() override { vsgpu_comm_destroy(comm); }
    int allgather(const void *send, size_t bytes, void *recv) override { return vsgpu_comm_allgather(comm, send, bytes, recv); }
    int broadcast(void *buf, size_t bytes, int root) override { return vsgpu_comm_broadcast(comm, buf, bytes, root); }
    void abort() override { (void)vsgpu_comm_abort(comm); }
    const char *mode() const override { return vsgpu_comm_staged(comm) ? "rccl-staged" : "rccl-mapped"; }
};

// a shard held by this process: a GPU Flat index
struct FlatShard final : ShardOps {
    std::unique_ptr<FlatIndex> ix;
    int add(const void *blob, size_t label) override { return ix->addVector(blob, label); }
    int candidates(const void *queries, size_t nq, size_t stride, size_t k, size_t cap, uint32_t *ids, size_t *labels,
                   double *scores, uint32_t *counts) override {
        return ix->topKCandidates(queries, nq, stride, k, cap, ids, labels, scores, counts);
    }
    size_t size() const override { return ix->indexSize(); }
    size_t storedBytes() const override { return ix->storedBlobBytes(); }
    bool supportsRowOps() const override { return true; }
    size_t firstNanRow() const override { return ix->firstNanRow(); }
    bool queryMayScoreNaN(const void *q) const override { return ix->queryMayScoreNaN(q); }
    int allScores(const void *query, uint32_t *ids, size_t *labels, double *scores) override {
        std::vector<char> q = ix->preprocessQuery(query);
        std::vector<double> sc;
        if (ix->allScores(q.data(), sc)) return -1;
        for (size_t i = 0; i < sc.size(); i++) {
            ids[i] = (uint32_t)i;
            labels[i] = ix->labelOf(i);
            scores[i] = sc[i];
        }
        return 0;
    }
    int readRow(uint32_t id, void *out) override { return ix->readRow(id, out); }
    int overwriteRow(uint32_t id, const void *blob, size_t label) override { return ix->overwriteRow(id, blob, label); }
    int dropLastRow() override { return ix->dropLastRow(); }
    long addSynthetic(size_t n, uint64_t seed) override { return ix->addSynthetic(n, seed); }
    VecSimIndexInterface *index() override { return ix.get(); }
};

std::unique_ptr<ShardOps> make_flat_shard(const BFParams &p, void *logCtx, int device) {
    const int saved = globals().device;
    globals().device = device;
    FlatIndex *ix = FlatIndex::create(p, logCtx);
    globals().device = saved;
    if (!ix) return nullptr;
    auto s = std::make_unique<FlatShard>();
    s->ix.reset(ix);
    return s;
}
}  // namespace

// One query of the merge: the parts' candidates with a score at or below the union's k-th smallest, in gid order, through the
// reference's heap loop.  part_stride: distance between two parts' arrays in 8-byte words (nq * cap when they are contiguous; the
// exchange records are merged where the collective wrote them).
namespace {
struct MergeCand {
    uint64_t gid;
    size_t label;
    double score;
};
struct MergeScratch {
    std::vector<MergeCand> c;
    std::vector<double> tmp;
    std::vector<std::pair<double, size_t>> store;
};
void merge_one_query(size_t q, size_t parts, size_t cap, size_t part_stride, const uint64_t *gids, const size_t *labels, const double *scores,
                     const uint32_t *counts, size_t nq, size_t k, size_t *out_labels, double *out_scores, uint32_t *out_counts, bool every_row,
                     MergeScratch &S) {
    auto &c = S.c;
    c.clear();
    for (size_t p = 0; p < parts; p++) {
        const size_t base = p * part_stride + q * cap;
        const uint32_t n = counts[p * nq + q];
        for (uint32_t i = 0; i < n; i++) c.push_back(MergeCand{gids[base + i], labels[base + i], scores[base + i]});
    }
    if (!every_row && c.size() > k) {   // (every_row: scores may be NaN -- no order to select by; the heap loop below is the reference's)
        auto &tmp = S.tmp;
        tmp.resize(c.size());
        for (size_t i = 0; i < c.size(); i++) tmp[i] = c[i].score;
        std::nth_element(tmp.begin(), tmp.begin() + (std::ptrdiff_t)(k - 1), tmp.end());
        const double T = tmp[k - 1];
        size_t w = 0;
        for (size_t i = 0; i < c.size(); i++)
            if (c[i].score <= T) c[w++] = c[i];
        c.resize(w);
    }
    // std::priority_queue<.., RefPairLess> spelled out over the scratch's storage (push_back + push_heap, pop_heap + pop_back: the
    // container's definition), so that a query allocates nothing
    auto &h = S.store;
    h.clear();
    const RefPairLess less{};
    if (!every_row && c.size() <= k) {
        // at most k rows at or below the k-th score (no tie across it): the heap loop ends holding exactly these whatever the order
        // they arrive in -- each finds the heap short of k or topped by a score above the k-th -- and pops them from the largest
        // (score, label) down: the reply is the set in ascending heap order, no replay needed
        bool ordered = true;
        for (const MergeCand &x : c) {
            ordered = ordered && x.score == x.score;
            h.emplace_back(x.score, x.label);
        }
        if (ordered) {
            std::sort(h.begin(), h.end(), less);
            out_counts[q] = (uint32_t)h.size();
            for (size_t i = 0; i < h.size(); i++) out_labels[q * k + i] = h[i].second, out_scores[q * k + i] = h[i].first;
            return;
        }
        h.clear();
    }
    std::sort(c.begin(), c.end(), [](const MergeCand &a, const MergeCand &b) { return a.gid < b.gid; });
    double upper = std::numeric_limits<double>::lowest();
    for (const MergeCand &x : c) {
        if (x.score < upper || h.size() < k) {
            h.emplace_back(x.score, x.label);
            std::push_heap(h.begin(), h.end(), less);
            if (h.size() > k) {
                std::pop_heap(h.begin(), h.end(), less);
                h.pop_back();
            }
            upper = h.front().first;
        }
    }
    out_counts[q] = (uint32_t)h.size();
    for (size_t i = h.size(); i-- > 0;) {
        out_labels[q * k + i] = h.front().second;
        out_scores[q * k + i] = h.front().first;
        std::pop_heap(h.begin(), h.end(), less);
        h.pop_back();
    }
}
}  // namespace

int merge_topk_strided(size_t nq, size_t parts, size_t cap, size_t part_stride, const uint64_t *gids, const size_t *labels, const double *scores,
                       const uint32_t *counts, size_t k, size_t *out_labels, double *out_scores, uint32_t *out_counts, bool every_row) {
    for (size_t i = 0; i < parts * nq; i++)
        if (counts[i] == 0xFFFFFFFFu || counts[i] > cap) return -1;   // (overflow marker; a count beyond the record's room is a corrupt record)
    for (size_t q = 0; q < nq; q++) out_counts[q] = 0;
    if (k == 0) return 0;
    auto range = [&](size_t q0, size_t q1) {
        MergeScratch S;
        for (size_t q = q0; q < q1; q++)
            merge_one_query(q, parts, cap, part_stride, gids, labels, scores, counts, nq, k, out_labels, out_scores, out_counts, every_row, S);
    };
    // queries are independent: wide AND deep batches merge on a few threads, as the single index's replay does (flat_index.cpp).  On the GPU
    // box's host (tools/merge_time.py, profiles/r05_merge_time.txt): config 3 from 8 shards -- 256 queries x 800 candidates, k = 100 -- 2.26 ms
    // on one thread, 0.65 on four, 0.46 on eight; configs 2 / 4 (k = 10) take 0.02-0.035 ms and stay on the caller's thread.  The work is
    // roughly one step per candidate plus the k-deep selection and ordering per query
    size_t total = 0;
    for (size_t i = 0; i < parts * nq; i++) total += counts[i];
    const size_t work = total + nq * std::min(k, total / std::max<size_t>(nq, 1) + 1) * 8;
    static const size_t max_workers = [] {   // VECSIM_GPU_MERGE_THREADS: 1 = merge on the calling thread
        const char *e = std::getenv("VECSIM_GPU_MERGE_THREADS");
        const long v = e ? std::atol(e) : 0;
        return v > 0 ? (size_t)std::min<long>(v, 64) : std::min<size_t>(8, std::max<size_t>(1, std::thread::hardware_concurrency()));
    }();
    const size_t workers = (!every_row && nq >= 16 && work >= 100000) ? max_workers : 1;
    if (workers > 1) {
        std::vector<std::thread> pool;
        const size_t per = (nq + workers - 1) / workers;
        for (size_t w = 1; w < workers; w++)
            if (w * per < nq) pool.emplace_back(range, w * per, std::min(nq, (w + 1) * per));
        range(0, std::min(nq, per));
        for (auto &th : pool) th.join();
    } else {
        range(0, nq);
    }
    return 0;
}

int merge_topk(size_t nq, size_t parts, size_t cap, const uint64_t *gids, const size_t *labels, const double *scores,
               const uint32_t *counts, size_t k, size_t *out_labels, double *out_scores, uint32_t *out_counts, bool every_row) {
    return merge_topk_strided(nq, parts, cap, nq * cap, gids, labels, scores, counts, k, out_labels, out_scores, out_counts, every_row);
}

// Multi-value merge (brute_force_multi.h:108-277, utils/updatable_heap.h:20-113): the union of the shards' rows in gid order
// through the label-keyed updatable heap -- a label keeps its lowest score, the heap keeps the k labels with the lowest,
// evicting the largest (score, label) on overflow; same procedure as FlatIndex::replayMulti on a single index.
int merge_topk_multi(size_t nq, size_t parts, size_t cap, const uint64_t *gids, const size_t *labels, const double *scores,
                     const uint32_t *counts, size_t k, size_t *out_labels, double *out_scores, uint32_t *out_counts) {
    return merge_topk_multi_strided(nq, parts, cap, nq * cap, gids, labels, scores, counts, k, out_labels, out_scores, out_counts);
}

int merge_topk_multi_strided(size_t nq, size_t parts, size_t cap, size_t part_stride, const uint64_t *gids, const size_t *labels,
                             const double *scores, const uint32_t *counts, size_t k, size_t *out_labels, double *out_scores,
                             uint32_t *out_counts) {
    struct Cand {
        uint64_t gid;
        size_t label;
        double score;
    };
    for (size_t i = 0; i < parts * nq; i++)
        if (counts[i] == 0xFFFFFFFFu) return -1;
    std::vector<Cand> c;
    for (size_t q = 0; q < nq; q++) {
        out_counts[q] = 0;
        if (k == 0) continue;
        c.clear();
        for (size_t p = 0; p < parts; p++) {
            const size_t base = p * part_stride + q * cap;
            for (uint32_t i = 0; i < counts[p * nq + q]; i++) c.push_back(Cand{gids[base + i], labels[base + i], scores[base + i]});
        }
        std::sort(c.begin(), c.end(), [](const Cand &a, const Cand &b) { return a.gid < b.gid; });
        std::multimap<double, size_t, std::greater<double>> by_score;
        std::unordered_map<size_t, std::multimap<double, size_t, std::greater<double>>::iterator> node_of;
        auto top_it = [&]() {
            auto rng = by_score.equal_range(by_score.begin()->first);
            auto best = rng.first;
            for (auto i = rng.first; i != rng.second; ++i)
                if (best->second < i->second) best = i;
            return best;
        };
        double upper = std::numeric_limits<double>::lowest();
        for (const Cand &x : c) {
            if (x.score < upper || node_of.size() < k) {
                auto f = node_of.find(x.label);
                if (f == node_of.end()) node_of.emplace(x.label, by_score.emplace(x.score, x.label));
                else if (f->second->first > x.score) {
                    by_score.erase(f->second);
                    f->second = by_score.emplace(x.score, x.label);
                }
                if (node_of.size() > k) {
                    auto t = top_it();
                    node_of.erase(t->second);
                    by_score.erase(t);
                }
                upper = top_it()->first;
            }
        }
        out_counts[q] = (uint32_t)node_of.size();
        for (size_t i = node_of.size(); i-- > 0;) {
            auto t = top_it();
            out_labels[q * k + i] = t->second;
            out_scores[q * k + i] = t->first;
            node_of.erase(t->second);
            by_score.erase(t);
        }
    }
    return 0;
}

std::unique_ptr<Exchange> make_rccl_exchange(vsgpu_ctx *ctx, int rank, int world, const void *id128) {
    vsgpu_comm *c = vsgpu_comm_create(ctx, rank, world, id128);
    if (!c) return nullptr;
    auto ex = std::make_unique<RcclExchange>();
    ex->comm = c;
    return ex;
}

ShardedIndex::~ShardedIndex() {
    ex_.reset();  // the communicator borrows the shard's GPU context: it goes first
    shards_.clear();
}

ShardedIndex *ShardedIndex::createDistributed(const BFParams &p, void *logCtx, int rank, int world, int device,
                                              std::unique_ptr<Exchange> ex, std::unique_ptr<ShardOps> external) {
    if (world < 1 || rank < 0 || rank >= world) return nullptr;
    auto *sx = new ShardedIndex();
    if (const char *e = std::getenv("VECSIM_GPU_TEST_FAIL_REMOVE_AT")) sx->test_fail_remove_at_ = std::atol(e);
    sx->params_ = p;
    sx->plan_.block = p.blockSize ? p.blockSize : DEFAULT_BLOCK_SIZE;
    sx->plan_.world = (size_t)world;
    sx->rank_ = rank;
    std::unique_ptr<ShardOps> s = external ? std::move(external) : make_flat_shard(p, logCtx, device);
    if (!s) {
        delete sx;
        return nullptr;
    }
    sx->shards_.push_back(std::move(s));
    sx->ex_ = std::move(ex);
    return sx;
}

ShardedIndex *ShardedIndex::createLocal(const BFParams &p, void *logCtx, int n_shards, const int *devices) {
    if (n_shards < 1) return nullptr;
    auto *sx = new ShardedIndex();
    if (const char *e = std::getenv("VECSIM_GPU_TEST_FAIL_REMOVE_AT")) sx->test_fail_remove_at_ = std::atol(e);
    sx->params_ = p;
    sx->plan_.block = p.blockSize ? p.blockSize : DEFAULT_BLOCK_SIZE;
    sx->plan_.world = (size_t)n_shards;
    sx->rank_ = -1;
    for (int s = 0; s < n_shards; s++) {
        auto sh = make_flat_shard(p, logCtx, devices ? devices[s] : 0);
        if (!sh) {
            delete sx;
            return nullptr;
        }
        sx->shards_.push_back(std::move(sh));
    }
    return sx;
}

VecSimIndexInterface *ShardedIndex::localIndex(int s) {
    if (s < 0 || (size_t)s >= plan_.world || !owns((size_t)s)) return nullptr;
    return shard((size_t)s)->index();
}

// ---- ingest (SPMD: every process makes the same calls in the same order) ----
int ShardedIndex::addVector(const void *blob, size_t label) {
    if (synthetic_rows_) return -1;  // synthetic fills are append-only through addSyntheticLocal
    auto f = params_.multi ? label_to_gid_.end() : label_to_gid_.find(label);   // (multi-value: a label's vectors just accumulate)
    if (f != label_to_gid_.end()) {
        // overwrite in place (brute_force_single.h:139-143): the row keeps its id, so only its owner acts and the
        // global count does not move
        const size_t s = plan_.owner(f->second);
        if (owns(s)) shard(s)->add(blob, label);
        return 0;
    }
    const uint64_t gid = n_global_;
    const size_t s = plan_.owner(gid);
    if (owns(s)) {
        if (shard(s)->size() != plan_.local(gid)) {
            std::fprintf(stderr, "vecsim_amd: shard %zu holds %zu rows where gid %llu expects local id %llu\n", s,
                         shard(s)->size(), (unsigned long long)gid, (unsigned long long)plan_.local(gid));
            return -1;
        }
        if (shard(s)->add(blob, label) != 1) return -1;
    }
    n_global_++;
    if (!params_.multi) label_to_gid_.emplace(label, gid);
    else label_to_gids_[label].push_back(gid);
    gid_to_label_.push_back(label);
    return 1;
}

long ShardedIndex::addBulk(const void *blobs, const size_t *labels, size_t n) {
    const size_t in_bytes = params_.dim * type_size(params_.type);
    long added = 0;
    for (size_t i = 0; i < n; i++) {
        const int rc = addVector(static_cast<const char *>(blobs) + i * in_bytes, labels[i]);
        if (rc < 0) return -1;
        added += rc;
    }
    return added;
}

long ShardedIndex::addSyntheticLocal(size_t rows_per_shard, uint64_t seed_base) {
    if (n_global_ != 0 || rows_per_shard == 0) return -1;
    for (size_t s = 0; s < plan_.world; s++)
        if (owns(s) && shard(s)->addSynthetic(rows_per_shard, seed_base + 1000 * s) != (long)rows_per_shard) return -1;
    synthetic_rows_ = rows_per_shard;
    n_global_ = rows_per_shard * plan_.world;
    return (long)n_global_;
}

// The swap-delete of ONE row of the equivalent single index (brute_force.h:196-224) across the shards: the last row moves into
// the hole (from its owner to the hole's owner, status travelling with it), the outcome is agreed on with one 8-byte all-gather,
// and only then do the maps move.  Every process makes the same call.
int ShardedIndex::removeGid(uint64_t hole) {
    const uint64_t last = n_global_ - 1;
    const size_t s_hole = plan_.owner(hole), s_last = plan_.owner(last);
    const size_t last_label = gid_to_label_[last];
    const size_t bytes = shards_[0]->storedBytes();
    uint64_t failed = 0;
    if (hole != last) {
        // the last row of the equivalent single index moves into the hole; the owner's status travels with the row
        std::vector<char> row(8 + bytes, 0);
        if (owns(s_last) && shard(s_last)->readRow((uint32_t)plan_.local(last), row.data() + 8)) row[0] = 1;
        // test hook ($VECSIM_GPU_TEST_FAIL_REMOVE_AT = n): the n-th row move reports a failed read on its owner -- the one failure
        // that leaves every shard untouched, so the caller may retry (tests/test_gpu_sharded.py)
        if (owns(s_last) && test_fail_remove_at_ >= 0 && test_removes_++ == test_fail_remove_at_) row[0] = 1;
        if (ex_ && s_hole != s_last && ex_->broadcast(row.data(), 8 + bytes, (int)s_last)) return -1;   // (transport failure: fatal everywhere)
        if (row[0]) failed = 1;
        else if (owns(s_hole) && shard(s_hole)->overwriteRow((uint32_t)plan_.local(hole), row.data() + 8, last_label)) failed = 1;
    }
    if (!failed && owns(s_last) && shard(s_last)->dropLastRow()) failed = 1;
    if (ex_) {   // agree on the outcome before the maps move
        std::vector<uint64_t> all(plan_.world, 0);
        if (ex_->allgather(&failed, 8, all.data())) return -1;
        for (uint64_t v : all) failed |= v;
    }
    if (failed) return -1;
    if (hole != last) {
        gid_to_label_[hole] = last_label;
        if (params_.multi) {   // replaceIdOfLabel (brute_force_multi.h:244-265): the LAST occurrence of the moved id
            auto &v = label_to_gids_.at(last_label);
            for (size_t i = v.size(); i-- > 0;)
                if (v[i] == last) {
                    v[i] = hole;
                    break;
                }
        } else {
            label_to_gid_[last_label] = hole;
        }
    }
    gid_to_label_.pop_back();
    n_global_--;
    return 0;
}

int ShardedIndex::exchangeSelfTest(size_t bytes) {
    if (!ex_ || rank_ < 0) return 0;
    bytes = std::max<size_t>(bytes, 64);
    const size_t world = plan_.world;
    auto stamp = [](size_t r, size_t i) { return (unsigned char)((r * 131 + i * 7 + (i >> 8) * 13 + 5) & 0xFF); };
    std::vector<unsigned char> mine(bytes), all(bytes * world, 0);
    for (size_t i = 0; i < bytes; i++) mine[i] = stamp((size_t)rank_, i);
    if (ex_->allgather(mine.data(), bytes, all.data())) return -1;
    uint64_t bad = 0;
    for (size_t r = 0; r < world && !bad; r++)
        for (size_t i = 0; i < bytes; i++)
            if (all[r * bytes + i] != stamp(r, i)) {
                bad = 1;
                break;
            }
    std::vector<uint64_t> verdicts(world, 0);
    if (ex_->allgather(&bad, 8, verdicts.data())) return -1;
    for (uint64_t v : verdicts) bad |= v;
    return bad ? -1 : 0;
}

int ShardedIndex::deleteVector(size_t label) {
    if (synthetic_rows_) return -1;
    // every process takes the same decisions from the same state (SPMD): shards or transports that cannot move rows are
    // refused before anything changes, and no process leaves between the collectives on a locally evaluated condition
    if (!shards_[0]->supportsRowOps() || (ex_ && !ex_->canBroadcast())) return -1;
    if (params_.multi) {
        // brute_force_multi.h:133-150: every vector of the label goes, one swap-delete at a time, walking the label's id list
        // while the removals rewrite its tail (a row of the same label may be the one that moves into a hole)
        auto f = label_to_gids_.find(label);
        if (f == label_to_gids_.end()) return 0;
        int removed = 0;
        for (size_t i = 0; i < f->second.size(); i++) {
            if (removeGid(f->second[i])) {
                // the ids already removed now name OTHER rows (swapped into the holes): they leave the label's list, so that
                // the maps keep agreeing with gid_to_label_ and a retried delete takes up where this one stopped
                f->second.erase(f->second.begin(), f->second.begin() + removed);
                if (f->second.empty()) label_to_gids_.erase(f);
                return -1;
            }
            removed++;
        }
        label_to_gids_.erase(label);
        return removed;
    }
    auto f = label_to_gid_.find(label);
    if (f == label_to_gid_.end()) return 0;
    const uint64_t hole = f->second;
    label_to_gid_.erase(f);
    if (removeGid(hole)) {
        label_to_gid_.emplace(label, hole);   // nothing moved: the label is still there
        return -1;
    }
    return 1;
}

// ---- query ----
// A numbered batch waits for its number; a batch WITHOUT a number takes the exchange section as a plain critical section, so its
// collectives never interleave with a numbered batch's on this process (mixing the two modes across processes is still the
// caller's to keep consistent, like the order of the calls themselves).  A number that has already passed is an error, not a
// wait for ever: next_seq_ only grows (resetSeq starts a new stream of batches at 0).
bool ShardedIndex::takeTurn(uint64_t seq) {
    std::unique_lock<std::mutex> lk(turn_mu_);
    if (seq == NO_SEQ) {
        turn_cv_.wait(lk, [&] { return !turn_held_; });
        turn_held_ = true;
        return true;
    }
    if (seq < next_seq_) return false;
    turn_cv_.wait(lk, [&] { return (next_seq_ == seq && !turn_held_) || seq < next_seq_; });
    if (seq < next_seq_) return false;   // (a resetSeq / an earlier duplicate overtook this call)
    turn_held_ = true;
    return true;
}
void ShardedIndex::passTurn(uint64_t seq) {
    {
        std::lock_guard<std::mutex> lk(turn_mu_);
        turn_held_ = false;
        if (seq != NO_SEQ) next_seq_ = seq + 1;
    }
    turn_cv_.notify_all();
}
void ShardedIndex::resetSeq() {
    {
        std::lock_guard<std::mutex> lk(turn_mu_);
        next_seq_ = 0;
    }
    turn_cv_.notify_all();
}
void ShardedIndex::stats(double out[6]) {
    std::lock_guard<std::mutex> lk(stats_mu_);
    out[0] = st_scan_, out[1] = st_wait_, out[2] = st_exchange_, out[3] = st_merge_, out[4] = st_batches_, out[5] = st_bytes_;
}
void ShardedIndex::resetStats() {
    std::lock_guard<std::mutex> lk(stats_mu_);
    st_scan_ = st_wait_ = st_exchange_ = st_merge_ = st_batches_ = st_bytes_ = 0;
}
namespace {
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

// One pass: candidates of the local shards at capacity `cap` (all_rows: every row's score, one query), exchange, merge.
// Record of one shard: header u64 [4] = {flags (1 = shard failed, 2 = timeout callback fired), smallest gid of a row that can
// score NaN, 0, 0} | counts u64 [nq] | gids u64 [nq][cap] | labels u64 [nq][cap] | scores f64 [nq][cap].  Whatever a process
// learns locally (a failure, a timeout, NaN-capable rows) travels in the header, so every process takes the same decision
// AFTER the exchange; nobody leaves before it.
int ShardedIndex::queryOnce(const void *queries, size_t nq, size_t stride, size_t k, size_t cap, bool local_timeout, bool all_rows,
                            const std::function<bool()> &poll_timeout,
                            const std::function<void(bool)> &turn_hook, std::vector<size_t> &out_labels,
                            std::vector<double> &out_scores, std::vector<uint32_t> &out_counts, Pass *pass) {
    const size_t G = plan_.world, rec = recordBytes(nq, cap);
    const size_t n_mine = shards_.size();
    std::vector<char> mine(n_mine * rec, 0);
    std::vector<int> rcs(n_mine, 0);
    const double t0 = now_ms();
    auto run = [&](size_t i) {
        const size_t s = rank_ < 0 ? i : (size_t)rank_;
        char *r = mine.data() + i * rec;
        uint64_t *hdr = reinterpret_cast<uint64_t *>(r);
        uint64_t *cnt = hdr + 4;
        uint64_t *gids = cnt + nq;
        size_t *labels = reinterpret_cast<size_t *>(gids + nq * cap);
        double *scores = reinterpret_cast<double *>(labels + nq * cap);
        const size_t fn = shards_[i]->firstNanRow();
        hdr[1] = fn == (size_t)-1 ? ~0ull : gidOf(fn, s);
        if (local_timeout) hdr[0] |= 2;
        std::vector<uint32_t> ids(nq * cap), c32(nq, 0);
        if (all_rows) {   // nq == 1
            const size_t n_local = shards_[i]->size();
            rcs[i] = n_local > cap ? -1 : shards_[i]->allScores(queries, ids.data(), labels, scores);
            c32[0] = (uint32_t)n_local;
        } else if (!local_timeout) {
            rcs[i] = shards_[i]->candidates(queries, nq, stride, k, cap, ids.data(), labels, scores, c32.data());
        }
        if (rcs[i]) {
            hdr[0] |= 1;
            return;
        }
        for (size_t q = 0; q < nq; q++) {
            cnt[q] = c32[q];
            if (c32[q] == 0xFFFFFFFFu) continue;
            for (uint32_t j = 0; j < c32[q]; j++) {
                gids[q * cap + j] = gidOf(ids[q * cap + j], s);
                if (synthetic_rows_) labels[q * cap + j] = (size_t)gids[q * cap + j];  // label := gid (globally unique)
            }
        }
    };
    if (n_mine > 1) {  // one host thread per shard: the GPUs scan concurrently
        std::vector<std::thread> pool;
        for (size_t i = 0; i < n_mine; i++) pool.emplace_back(run, i);
        for (auto &t : pool) t.join();
    } else {
        run(0);
    }
    if (poll_timeout && poll_timeout()) reinterpret_cast<uint64_t *>(mine.data())[0] |= 2;   // (polled again behind the scan)
    const double t1 = now_ms();
    const char *all = mine.data();
    std::vector<char> gathered;
    double t_turn = 0;
    if (ex_) {
        const double tw = now_ms();
        turn_hook(true);   // this batch's turn in the stream of exchanges (seq order on every process)
        t_turn = now_ms() - tw;
        gathered.resize(G * rec);
        if (ex_->allgather(mine.data(), rec, gathered.data())) return -1;
        all = gathered.data();
    }
    const double t2 = now_ms();
    // the merge reads the records where the collective wrote them (parts rec / 8 words apart); only the counts are repacked
    std::vector<uint32_t> counts(G * nq);
    bool failed = false;
    uint64_t min_nan = ~0ull;
    for (size_t p = 0; p < G; p++) {
        const uint64_t *hdr = reinterpret_cast<const uint64_t *>(all + p * rec);
        const uint64_t *cnt = hdr + 4;
        if (hdr[0] & 1) failed = true;
        if (hdr[0] & 2) pass->timed_out = true;
        min_nan = std::min(min_nan, hdr[1]);
        for (size_t q = 0; q < nq; q++) {
            counts[p * nq + q] = (uint32_t)cnt[q];
            if ((uint32_t)cnt[q] == 0xFFFFFFFFu) pass->overflow = true;
        }
    }
    const uint64_t *rec_gids = reinterpret_cast<const uint64_t *>(all) + 4 + nq;
    const size_t *rec_labels = reinterpret_cast<const size_t *>(rec_gids + nq * cap);
    const double *rec_scores = reinterpret_cast<const double *>(rec_gids + 2 * nq * cap);
    pass->nan_rows_at_head = !all_rows && !params_.multi && min_nan < (uint64_t)k;   // (multi-value: as on a single index, no NaN-aware replay)
    // every process knows by now whether this batch exchanges again (ties beyond cap, NaN-aware passes): if not, the turn
    // goes to the next batch before the merge
    if (failed || pass->timed_out || !(pass->overflow || pass->nan_rows_at_head || pass->more_follows)) turn_hook(false);
    int rc = 0;
    if (failed) rc = -1;
    else if (!pass->timed_out && !pass->overflow) {
        out_labels.assign(nq * k, 0);
        out_scores.assign(nq * k, 0.0);
        out_counts.assign(nq, 0);
        rc = params_.multi ? merge_topk_multi_strided(nq, G, cap, rec / 8, rec_gids, rec_labels, rec_scores, counts.data(), k,
                                                      out_labels.data(), out_scores.data(), out_counts.data())
                           : merge_topk_strided(nq, G, cap, rec / 8, rec_gids, rec_labels, rec_scores, counts.data(), k, out_labels.data(),
                                                out_scores.data(), out_counts.data(), all_rows);
    }
    const double t3 = now_ms();
    {
        std::lock_guard<std::mutex> lk(stats_mu_);
        st_scan_ += t1 - t0, st_exchange_ += t2 - t1 - t_turn, st_merge_ += t3 - t2, st_bytes_ += ex_ ? (double)rec : 0.0;
    }
    return rc;
}

int ShardedIndex::topKQueryBatch(const void *queries, size_t nq, size_t stride, size_t k, VecSimQueryParams *qp,
                                 VecSimQueryReply_Order order, VecSimQueryReply **out, uint64_t seq) {
    void *tctx = qp ? qp->timeoutCtx : nullptr;
    // (a batch with a sequence number takes and passes its turn even when there is nothing to exchange)
    struct Turn {
        ShardedIndex *sx;
        uint64_t seq;
        bool taken = false, stale = false;
        void take() {
            if (!taken && !stale) {
                const double t0 = now_ms();
                if (!sx->takeTurn(seq)) {
                    stale = true;
                    return;
                }
                taken = true;
                std::lock_guard<std::mutex> lk(sx->stats_mu_);
                sx->st_wait_ += now_ms() - t0;
            }
        }
        bool passed = false;
        void release() {   // no further exchange belongs to this batch: the next batch's may go
            if (passed) return;
            take();
            if (taken) sx->passTurn(seq);
            passed = true;
        }
        ~Turn() { release(); }
    } turn{this, seq};
    if (seq != NO_SEQ) {   // a number that already passed would wait for ever (and strand the peers in their collective)
        std::lock_guard<std::mutex> lk(turn_mu_);
        if (seq < next_seq_) {
            turn.stale = turn.passed = true;
            std::fprintf(stderr, "vecsim_amd: sharded batch with sequence number %llu after %llu was answered (VecSimGpu_ShardedResetSeq starts a new stream)\n",
                         (unsigned long long)seq, (unsigned long long)next_seq_);
            return -1;
        }
    }
    if (nq == 0) return 0;
    std::vector<VecSimQueryReply *> reps(nq);
    for (auto &r : reps) r = new VecSimQueryReply();
    auto finish = [&]() {
        for (size_t q = 0; q < nq; q++) out[q] = reps[q];
        return 0;
    };
    auto fail_all = [&](int rc) {
        for (auto *r : reps) delete r;
        return rc;
    };
    auto time_out_all = [&]() {
        for (auto *r : reps) {
            r->results.clear();
            r->code = VecSim_QueryReply_TimedOut;
        }
        return finish();
    };
    if (k == 0 || n_global_ == 0) return finish();
    // queries that can score NaN themselves (the same answer on every process: a function of the query alone)
    std::vector<char> needs_all(nq, 0);
    bool any_needs_all = false;
    for (size_t q = 0; q < nq; q++)
        if (!params_.multi && shards_[0]->queryMayScoreNaN(static_cast<const char *>(queries) + q * stride)) needs_all[q] = 1, any_needs_all = true;
    std::vector<size_t> labels;
    std::vector<double> scores;
    std::vector<uint32_t> found;
    size_t cap = std::max<size_t>(2 * k, k + 16);
    Pass pass;
    // the timeout callback is polled locally before the scan; its verdict travels in the exchange (a process that left here
    // on its own would strand the others in the collective)
    bool local_timeout = timed_out(tctx);
    const std::function<bool()> poll = [&]() { return timed_out(tctx); };
    // the scan of a pass runs outside the turn (it overlaps with other batches' exchanges and merges on this process); the turn
    // is taken right before the exchange and held until the reply is complete
    // (after_exchange: the merge of the last pass runs outside the turn again)
    const std::function<void(bool)> turn_hook = [&](bool before) {
        if (before) turn.take();
        else turn.release();
    };
    for (;;) {
        pass = Pass();
        pass.more_follows = any_needs_all;
        int rc = queryOnce(queries, nq, stride, k, cap, local_timeout, false, poll, turn_hook, labels, scores, found, &pass);
        if (rc) return fail_all(rc);
        if (pass.timed_out) return time_out_all();
        if (!pass.overflow) break;
        // more than `cap` rows tie at some shard's k-th score: again with room for every tie (every process saw
        // the same counts, so all of them come back here together)
        const size_t biggest = synthetic_rows_ ? synthetic_rows_ : (n_global_ / plan_.world + plan_.block);
        if (cap >= biggest) return fail_all(-1);
        cap = std::min(biggest, cap * 8);
    }
    // NaN-aware replies (brute_force.h:272: a NaN score enters the heap only while it fills): when a row that can score NaN sits
    // below gid k, or the query itself can, the reference's reply depends on every row in id order -- those queries are answered
    // from every shard's full score vector, one query per exchange
    if (pass.nan_rows_at_head || any_needs_all) {
        const size_t cap_all = synthetic_rows_ ? synthetic_rows_ : (n_global_ / plan_.world + plan_.block);
        size_t last_q = 0;
        for (size_t q = 0; q < nq; q++)
            if (pass.nan_rows_at_head || needs_all[q]) last_q = q;
        for (size_t q = 0; q < nq; q++) {
            if (!pass.nan_rows_at_head && !needs_all[q]) continue;
            std::vector<size_t> l1;
            std::vector<double> s1;
            std::vector<uint32_t> f1;
            Pass p1;
            p1.more_follows = q != last_q;
            int rc = queryOnce(static_cast<const char *>(queries) + q * stride, 1, 0, k, cap_all, false, true, nullptr, turn_hook, l1, s1, f1, &p1);
            if (rc) return fail_all(rc);
            found[q] = f1[0];
            for (size_t j = 0; j < f1[0]; j++) labels[q * k + j] = l1[j], scores[q * k + j] = s1[j];
        }
    }
    const double t0 = now_ms();
    for (size_t q = 0; q < nq; q++) {
        auto &res = reps[q]->results;
        for (size_t j = 0; j < found[q]; j++) res.push_back(VecSimQueryResult{labels[q * k + j], scores[q * k + j]});
        if (order == BY_ID) sort_reply(reps[q], BY_ID);
    }
    {
        std::lock_guard<std::mutex> lk(stats_mu_);
        st_merge_ += now_ms() - t0;
        st_batches_ += 1;
    }
    return finish();
}

}  // namespace vsa
