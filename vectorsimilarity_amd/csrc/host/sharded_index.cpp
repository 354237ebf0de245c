// sharded_index.cpp -- see sharded_index.h
#include "sharded_index.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <limits>
#include <queue>
#include <thread>

#include "blob_prep.h"

namespace vsa {

namespace {
// RCCL over xGMI (include/vsgpu.h comm group)
struct RcclExchange final : Exchange {
    vsgpu_comm *comm = nullptr;
    ~RcclExchange() override { vsgpu_comm_destroy(comm); }
    int allgather(const void *send, size_t bytes, void *recv) override { return vsgpu_comm_allgather(comm, send, bytes, recv); }
    int broadcast(void *buf, size_t bytes, int root) override { return vsgpu_comm_broadcast(comm, buf, bytes, root); }
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

int merge_topk(size_t nq, size_t parts, size_t cap, const uint64_t *gids, const size_t *labels, const double *scores,
               const uint32_t *counts, size_t k, size_t *out_labels, double *out_scores, uint32_t *out_counts) {
    struct Cand {
        uint64_t gid;
        size_t label;
        double score;
    };
    for (size_t i = 0; i < parts * nq; i++)
        if (counts[i] == 0xFFFFFFFFu) return -1;
    for (size_t q = 0; q < nq; q++) out_counts[q] = 0;
    if (k == 0) return 0;
    std::vector<Cand> c;
    std::vector<double> tmp;
    for (size_t q = 0; q < nq; q++) {
        c.clear();
        for (size_t p = 0; p < parts; p++) {
            const size_t base = (p * nq + q) * cap;
            for (uint32_t i = 0; i < counts[p * nq + q]; i++) c.push_back(Cand{gids[base + i], labels[base + i], scores[base + i]});
        }
        if (c.size() > k) {
            tmp.resize(c.size());
            for (size_t i = 0; i < c.size(); i++) tmp[i] = c[i].score;
            std::nth_element(tmp.begin(), tmp.begin() + (std::ptrdiff_t)(k - 1), tmp.end());
            const double T = tmp[k - 1];
            size_t w = 0;
            for (size_t i = 0; i < c.size(); i++)
                if (c[i].score <= T) c[w++] = c[i];
            c.resize(w);
        }
        std::sort(c.begin(), c.end(), [](const Cand &a, const Cand &b) { return a.gid < b.gid; });
        std::priority_queue<std::pair<double, size_t>> heap;
        double upper = std::numeric_limits<double>::lowest();
        for (const Cand &x : c) {
            if (x.score < upper || heap.size() < k) {
                heap.emplace(x.score, x.label);
                if (heap.size() > k) heap.pop();
                upper = heap.top().first;
            }
        }
        out_counts[q] = (uint32_t)heap.size();
        for (size_t i = heap.size(); i-- > 0;) {
            out_labels[q * k + i] = heap.top().second;
            out_scores[q * k + i] = heap.top().first;
            heap.pop();
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
    if (world < 1 || rank < 0 || rank >= world || p.multi) return nullptr;
    auto *sx = new ShardedIndex();
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
    if (n_shards < 1 || p.multi) return nullptr;
    auto *sx = new ShardedIndex();
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
    auto f = label_to_gid_.find(label);
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
    label_to_gid_.emplace(label, gid);
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

int ShardedIndex::deleteVector(size_t label) {
    if (synthetic_rows_) return -1;
    auto f = label_to_gid_.find(label);
    if (f == label_to_gid_.end()) return 0;
    const uint64_t hole = f->second, last = n_global_ - 1;
    const size_t s_hole = plan_.owner(hole), s_last = plan_.owner(last);
    const size_t last_label = gid_to_label_[last];
    const size_t bytes = shards_[0]->storedBytes();
    if (hole != last) {
        // the last row of the equivalent single index moves into the hole
        std::vector<char> row(bytes);
        if (owns(s_last) && shard(s_last)->readRow((uint32_t)plan_.local(last), row.data())) return -1;
        if (ex_ && s_hole != s_last && ex_->broadcast(row.data(), bytes, (int)s_last)) return -1;
        if (owns(s_hole) && shard(s_hole)->overwriteRow((uint32_t)plan_.local(hole), row.data(), last_label)) return -1;
        gid_to_label_[hole] = last_label;
        label_to_gid_[last_label] = hole;
    }
    if (owns(s_last) && shard(s_last)->dropLastRow()) return -1;
    label_to_gid_.erase(label);
    gid_to_label_.pop_back();
    n_global_--;
    return 1;
}

// ---- query ----
// One pass: candidates of the local shards at capacity `cap`, exchange, merge.  *overflow is set (identically on
// every process) when some shard had more than `cap` rows tied at or below its local k-th score.
int ShardedIndex::queryOnce(const void *queries, size_t nq, size_t stride, size_t k, size_t cap,
                            std::vector<size_t> &out_labels, std::vector<double> &out_scores,
                            std::vector<uint32_t> &out_counts, bool *overflow) {
    const size_t G = plan_.world, rec = recordBytes(nq, cap);
    const size_t n_mine = shards_.size();
    // record of one shard: counts u64 [nq] | gids u64 [nq][cap] | labels u64 [nq][cap] | scores f64 [nq][cap]
    std::vector<char> mine(n_mine * rec, 0);
    std::vector<int> rcs(n_mine, 0);
    auto run = [&](size_t i) {
        const size_t s = rank_ < 0 ? i : (size_t)rank_;
        char *r = mine.data() + i * rec;
        uint64_t *cnt = reinterpret_cast<uint64_t *>(r);
        uint64_t *gids = cnt + nq;
        size_t *labels = reinterpret_cast<size_t *>(gids + nq * cap);
        double *scores = reinterpret_cast<double *>(labels + nq * cap);
        std::vector<uint32_t> ids(nq * cap), c32(nq);
        rcs[i] = shards_[i]->candidates(queries, nq, stride, k, cap, ids.data(), labels, scores, c32.data());
        if (rcs[i]) return;
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
    int rc = 0;
    for (int r : rcs) rc = rc ? rc : r;
    const char *all = mine.data();
    std::vector<char> gathered;
    if (ex_) {
        // a failed shard still takes part in the exchange (a rank that skipped it would hang the others): it
        // contributes a record flagged as failed
        if (rc) reinterpret_cast<uint64_t *>(mine.data())[0] = 0xFFFFFFFEull;
        gathered.resize(G * rec);
        if (ex_->allgather(mine.data(), rec, gathered.data())) return -1;
        all = gathered.data();
        for (size_t p = 0; p < G; p++)
            if (reinterpret_cast<const uint64_t *>(all + p * rec)[0] == 0xFFFFFFFEull) return -1;
    } else if (rc) {
        return rc;
    }
    // repack for the merge: [part][nq][cap] arrays
    std::vector<uint32_t> counts(G * nq);
    std::vector<uint64_t> gids(G * nq * cap);
    std::vector<size_t> labels(G * nq * cap);
    std::vector<double> scores(G * nq * cap);
    *overflow = false;
    for (size_t p = 0; p < G; p++) {
        const char *r = all + p * rec;
        const uint64_t *cnt = reinterpret_cast<const uint64_t *>(r);
        for (size_t q = 0; q < nq; q++) {
            counts[p * nq + q] = (uint32_t)cnt[q];
            if ((uint32_t)cnt[q] == 0xFFFFFFFFu) *overflow = true;
        }
        std::memcpy(gids.data() + p * nq * cap, cnt + nq, nq * cap * 8);
        std::memcpy(labels.data() + p * nq * cap, cnt + nq + nq * cap, nq * cap * 8);
        std::memcpy(scores.data() + p * nq * cap, cnt + nq + 2 * nq * cap, nq * cap * 8);
    }
    if (*overflow) return 0;
    out_labels.assign(nq * k, 0);
    out_scores.assign(nq * k, 0.0);
    out_counts.assign(nq, 0);
    return merge_topk(nq, G, cap, gids.data(), labels.data(), scores.data(), counts.data(), k, out_labels.data(),
                      out_scores.data(), out_counts.data());
}

int ShardedIndex::topKQueryBatch(const void *queries, size_t nq, size_t stride, size_t k, VecSimQueryParams *qp,
                                 VecSimQueryReply_Order order, VecSimQueryReply **out) {
    void *tctx = qp ? qp->timeoutCtx : nullptr;
    if (nq == 0) return 0;
    std::vector<VecSimQueryReply *> reps(nq);
    for (auto &r : reps) r = new VecSimQueryReply();
    auto finish = [&]() {
        for (size_t q = 0; q < nq; q++) out[q] = reps[q];
        return 0;
    };
    if (k == 0 || n_global_ == 0) return finish();
    if (timed_out(tctx)) {
        for (auto *r : reps) r->code = VecSim_QueryReply_TimedOut;
        return finish();
    }
    std::vector<size_t> labels;
    std::vector<double> scores;
    std::vector<uint32_t> found;
    size_t cap = std::max<size_t>(2 * k, k + 16);
    for (;;) {
        bool overflow = false;
        int rc = queryOnce(queries, nq, stride, k, cap, labels, scores, found, &overflow);
        if (rc) {
            for (auto *r : reps) delete r;
            return rc;
        }
        if (!overflow) break;
        // more than `cap` rows tie at some shard's k-th score: again with room for every tie (every process saw
        // the same counts, so all of them come back here together)
        const size_t biggest = synthetic_rows_ ? synthetic_rows_ : (n_global_ / plan_.world + plan_.block);
        if (cap >= biggest) {
            for (auto *r : reps) delete r;
            return -1;
        }
        cap = std::min(biggest, cap * 8);
    }
    if (timed_out(tctx)) {
        for (auto *r : reps) r->code = VecSim_QueryReply_TimedOut;
        return finish();
    }
    for (size_t q = 0; q < nq; q++) {
        auto &res = reps[q]->results;
        for (size_t j = 0; j < found[q]; j++) res.push_back(VecSimQueryResult{labels[q * k + j], scores[q * k + j]});
        if (order == BY_ID) sort_reply(reps[q], BY_ID);
    }
    return finish();
}

}  // namespace vsa
