// flat_index.cpp -- see flat_index.h
#include "flat_index.h"
#include "host_tier.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <chrono>
#include <map>
#include <memory>
#include <queue>
#include <thread>
#include <unordered_set>

#include "blob_prep.h"
#include "ref_heap.h"
#include "sq8_prep.h"

namespace vsa {

Globals &globals() {
    static Globals g;
    return g;
}

// utils/vec_utils.cpp:100-130 -- same comparators, same std::sort, same input order => same output
void sort_reply(VecSimQueryReply *rep, VecSimQueryReply_Order order) {
    auto &r = rep->results;
    switch (order) {
    case BY_ID:
        std::sort(r.begin(), r.end(), [](const VecSimQueryResult &a, const VecSimQueryResult &b) { return a.id < b.id; });
        break;
    case BY_SCORE:
        std::sort(r.begin(), r.end(),
                  [](const VecSimQueryResult &a, const VecSimQueryResult &b) { return a.score < b.score; });
        break;
    case BY_SCORE_THEN_ID:
        std::sort(r.begin(), r.end(), [](const VecSimQueryResult &a, const VecSimQueryResult &b) {
            return a.score == b.score ? a.id < b.id : a.score < b.score;
        });
        break;
    }
}

static int resolve_device() {
    int d = globals().device;
    if (d >= 0) return d;
    if (const char *e = std::getenv("VECSIM_GPU_DEVICE")) return std::atoi(e);
    if (const char *e = std::getenv("LOCAL_RANK")) {
        int n = vsgpu_device_count();
        if (n > 0) return std::atoi(e) % n;
    }
    return 0;
}


static size_t reader_lanes() {
    if (const char *e = std::getenv("VECSIM_GPU_READER_LANES")) return (size_t)std::max(1, std::min(8, std::atoi(e)));
    return 2;
}

FlatIndex *FlatIndex::create(const BFParams &p, void *logCtx) {
    if (p.dim == 0 || p.type > VecSimType_UINT8 || p.metric > VecSimMetric_Cosine) return nullptr;
    vsgpu_ctx *ctx = vsgpu_ctx_create(resolve_device());
    if (!ctx) return nullptr;
    FlatIndex *ix = new FlatIndex();
    ix->type_ = p.type;
    ix->metric_ = p.metric;
    ix->dim_ = p.dim;
    ix->block_size_ = p.blockSize ? p.blockSize : DEFAULT_BLOCK_SIZE;
    ix->stored_bytes_ = blob_bytes(p.type, p.dim, p.metric);
    ix->query_bytes_ = ix->stored_bytes_;
    ix->multi_ = p.multi;
    ix->log_ctx_ = logCtx;
    ix->ctx_ = ctx;
    ix->tier_ = resolve_tier((int)p.type);   // from the host's CPUID, as the reference's choosers do (host_tier.h)
    ix->table_ = vsgpu_table_create(ctx, (int)p.type, (int)p.metric, ix->tier_, p.dim, ix->stored_bytes_);
    if (!ix->table_) {
        vsgpu_ctx_destroy(ctx);
        ix->ctx_ = nullptr;
        delete ix;
        return nullptr;
    }
    for (size_t i = 1; i < reader_lanes(); i++) {   // extra reader lanes (best effort: without them readers take turns)
        auto lane = std::make_unique<Lane>();
        lane->ctx = vsgpu_ctx_create(vsgpu_ctx_device(ctx));
        if (!lane->ctx) break;
        lane->view = vsgpu_table_view_create(ix->table_, lane->ctx);
        if (!lane->view) {
            vsgpu_ctx_destroy(lane->ctx);
            break;
        }
        ix->lanes_.push_back(std::move(lane));
    }
    return ix;
}

FlatIndex::Lane *FlatIndex::tryLane() {
    for (auto &l : lanes_)
        if (l->mu.try_lock()) return l.get();
    return nullptr;
}
std::vector<vsgpu_ctx *> FlatIndex::gpus() {
    std::vector<vsgpu_ctx *> v{ctx_};
    for (auto &l : lanes_) v.push_back(l->ctx);
    return v;
}

FlatIndex *FlatIndex::createSQ8(const BFParams &p, void *logCtx, const float *mean, float mean_sum_squares) {
    if (p.dim == 0 || (p.type != VecSimType_FLOAT32 && p.type != VecSimType_FLOAT16) || p.metric > VecSimMetric_Cosine) return nullptr;
    if (mean && p.metric == VecSimMetric_Cosine) return nullptr;   // "WithNorm does not support Cosine metric" (preprocessors.h:267)
    vsgpu_ctx *ctx = vsgpu_ctx_create(resolve_device());
    if (!ctx) return nullptr;
    const bool f16 = p.type == VecSimType_FLOAT16;
    FlatIndex *ix = new FlatIndex();
    ix->type_ = p.type;
    ix->sq8_ = true;
    ix->metric_ = p.metric;
    ix->dim_ = p.dim;
    ix->block_size_ = p.blockSize ? p.blockSize : DEFAULT_BLOCK_SIZE;
    if (mean) ix->sq8_mean_.assign(mean, mean + p.dim);
    ix->sq8_mss_ = mean_sum_squares;
    ix->stored_bytes_ = sq8_storage_bytes(p.dim, p.metric, mean != nullptr);
    ix->query_bytes_ = sq8_query_bytes(p.dim, p.metric, mean != nullptr, f16);
    ix->multi_ = p.multi;
    ix->log_ctx_ = logCtx;
    ix->ctx_ = ctx;
    ix->tier_ = resolve_tier();
    ix->table_ = vsgpu_table_create(ctx, f16 ? VSGPU_SQ8H : VSGPU_SQ8, (int)p.metric, ix->tier_, p.dim, ix->stored_bytes_);
    if (ix->table_ && mean) vsgpu_table_set_sq8_mean_sum_squares(ix->table_, mean_sum_squares);
    if (!ix->table_) {
        vsgpu_ctx_destroy(ctx);
        ix->ctx_ = nullptr;
        delete ix;
        return nullptr;
    }
    for (size_t i = 1; i < reader_lanes(); i++) {   // extra reader lanes (best effort: without them readers take turns)
        auto lane = std::make_unique<Lane>();
        lane->ctx = vsgpu_ctx_create(vsgpu_ctx_device(ctx));
        if (!lane->ctx) break;
        lane->view = vsgpu_table_view_create(ix->table_, lane->ctx);
        if (!lane->view) {
            vsgpu_ctx_destroy(lane->ctx);
            break;
        }
        ix->lanes_.push_back(std::move(lane));
    }
    return ix;
}

// caller's vector -> stored blob.  Cosine: normalise (vec_sim_index.h:397-402); SQ8: then quantise (preprocessors.h:270-390)
void FlatIndex::toStored(const void *blob, char *out) const {
    if (sq8_) {
        std::vector<float> tmp(dim_);
        if (type_ == VecSimType_FLOAT16) {   // normalise in fp16 as a Cosine fp16 index does, then widen exactly
            std::vector<uint16_t> h(dim_);
            std::memcpy(h.data(), blob, dim_ * 2);
            if (metric_ == VecSimMetric_Cosine) normalize_blob(h.data(), dim_, type_);
            for (size_t i = 0; i < dim_; i++) tmp[i] = fp16_widen(h[i]);
        } else {
            std::memcpy(tmp.data(), blob, dim_ * sizeof(float));
            if (metric_ == VecSimMetric_Cosine) normalize_blob(tmp.data(), dim_, type_);
        }
        if (!sq8_mean_.empty()) {
            std::vector<float> scratch(dim_);
            sq8_quantize_centred(tmp.data(), sq8_mean_.data(), dim_, metric_, reinterpret_cast<uint8_t *>(out), scratch.data());
        } else {
            sq8_quantize(tmp.data(), dim_, metric_, reinterpret_cast<uint8_t *>(out));
        }
        return;
    }
    std::memcpy(out, blob, dim_ * type_size(type_));
    if (metric_ == VecSimMetric_Cosine) normalize_blob(out, dim_, type_);
}
void FlatIndex::toQuery(const void *query, char *out) const {
    std::memcpy(out, query, dim_ * type_size(type_));
    if (metric_ == VecSimMetric_Cosine) normalize_blob(out, dim_, type_);
    if (sq8_ && !sq8_mean_.empty()) {
        // preprocessQuery, WithNorm (preprocessors.h:574-598): L2 stores the centred query (fp16: re-rounded), IP the raw
        // one; metadata over the stored body, y_mean_ip over the original input
        const bool f16 = type_ == VecSimType_FLOAT16;
        std::vector<float> orig(dim_), body(dim_);
        for (size_t i = 0; i < dim_; i++) {
            if (f16) {
                uint16_t h;
                std::memcpy(&h, out + 2 * i, 2);
                orig[i] = fp16_widen(h);
                if (metric_ == VecSimMetric_L2) {
                    h = fp16_round(orig[i] - sq8_mean_[i]);
                    std::memcpy(out + 2 * i, &h, 2);
                }
                body[i] = fp16_widen(h);
            } else {
                std::memcpy(&orig[i], out + 4 * i, 4);
                body[i] = metric_ == VecSimMetric_L2 ? orig[i] - sq8_mean_[i] : orig[i];
                std::memcpy(out + 4 * i, &body[i], 4);
            }
        }
        float meta[2];
        sq8_query_meta_centred(body.data(), orig.data(), sq8_mean_.data(), dim_, metric_, meta);
        std::memcpy(out + dim_ * (f16 ? 2 : 4), meta, sizeof meta);
        return;
    }
    if (sq8_ && type_ == VecSimType_FLOAT16) {
        std::vector<float> wide(dim_);
        for (size_t i = 0; i < dim_; i++) {
            uint16_t h;
            std::memcpy(&h, out + 2 * i, 2);
            wide[i] = fp16_widen(h);
        }
        sq8_query_meta_f16(wide.data(), dim_, metric_, out);
    } else if (sq8_) {
        sq8_query_blob(reinterpret_cast<const float *>(out), dim_, metric_, reinterpret_cast<float *>(out));
    }
}

// stored blobs and query blobs start with dim_ elements of the same kind (SQ8 storage: codes, then FP32 metadata)
bool FlatIndex::mayScoreNaN(const char *b) const {
    if (is_int_type(type_)) {
        if (metric_ != VecSimMetric_Cosine) return false;
        float norm;  // 1 - dot / (norm_x * norm_q): a zero vector makes 0 / 0
        std::memcpy(&norm, b + dim_, 4);
        return !(norm > 0.0f && norm <= 1e15f);
    }
    return values_may_nan(b, type_, dim_);
}
void FlatIndex::noteRow(uint32_t id, const void *stored) {
    const char *b = static_cast<const char *>(stored);
    bool w;
    if (sq8_) {
        w = values_may_nan(b + dim_, VecSimType_FLOAT32, metric_ == VecSimMetric_L2 ? 4 : 3);
        // (the filter's table-wide maxima are kept on the device: k_row_aux_sq8)
    } else {
        w = mayScoreNaN(b);
    }
    if (w) nan_ids_.insert(id);
    else if (!nan_ids_.empty()) nan_ids_.erase(id);
}

double FlatIndex::storedDistance(size_t label_a, size_t label_b) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);
    if (!sq8_ || multi_) return std::numeric_limits<double>::quiet_NaN();
    auto ia = label_to_id_.find(label_a), ib = label_to_id_.find(label_b);
    if (ia == label_to_id_.end() || ib == label_to_id_.end() || flush()) return std::numeric_limits<double>::quiet_NaN();
    const uint32_t a = ia->second, b = ib->second;
    double sc = 0;
    if (vsgpu_sq8_pair_scores(table_, &a, &b, 1, &sc)) return std::numeric_limits<double>::quiet_NaN();
    return sc;
}

FlatIndex::~FlatIndex() {
    for (auto &l : lanes_) {
        if (l->view) vsgpu_table_destroy(l->view);
        if (l->ctx) vsgpu_ctx_destroy(l->ctx);
    }
    if (table_) vsgpu_table_destroy(table_);
    if (ctx_) vsgpu_ctx_destroy(ctx_);
}

void FlatIndex::log(const char *level, const char *fmt, ...) const {
    logCallbackFunction cb = globals().log_cb;
    if (!cb) return;
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    cb(log_ctx_, level, buf);
}

// ---- ingest ----
void FlatIndex::stageRow(const void *processed) {
    const char *p = static_cast<const char *>(processed);
    staged_.insert(staged_.end(), p, p + stored_bytes_);
    staged_rows_++;
}

int FlatIndex::flush() {
    if (staged_rows_ == 0) return 0;
    int rc = vsgpu_table_append(table_, staged_.data(), staged_rows_.load());
    if (rc) {
        log("warning", "device append failed: %s", vsgpu_last_error());
        return rc;
    }
    staged_.clear();
    staged_rows_.store(0, std::memory_order_release);
    return 0;
}

int FlatIndex::addVector(const void *blob, size_t label) {
    auto it = multi_ ? label_to_id_.end() : label_to_id_.find(label);  // multi: every call appends (brute_force_multi.h:128-132)
    if (it != label_to_id_.end()) {
        // Overwrite.  The reference copies the caller's blob as-is here, without the storage
        // preprocessing (brute_force_single.h:139-143 -> updateElement).  fp blobs have exactly the
        // stored size, so we do the same; for int8/uint8 Cosine the reference would read 4 bytes
        // past the caller's blob for the norm -- we recompute the norm instead.
        if (flush()) return 0;
        if (sq8_) {   // (no reference behaviour to follow here: SQ8 rows are only ever written by the preprocessor)
            std::vector<char> tmp(stored_bytes_);
            toStored(blob, tmp.data());
            noteRow(it->second, tmp.data());
            vsgpu_table_write(table_, it->second, tmp.data());
        } else if (metric_ == VecSimMetric_Cosine && is_int_type(type_)) {
            std::vector<char> tmp(stored_bytes_);
            std::memcpy(tmp.data(), blob, dim_);
            normalize_blob(tmp.data(), dim_, type_);
            noteRow(it->second, tmp.data());
            vsgpu_table_write(table_, it->second, tmp.data());
        } else {
            noteRow(it->second, blob);
            vsgpu_table_write(table_, it->second, blob);
        }
        return 0;
    }
    // appendVector (brute_force.h:175-193): preprocess for storage, next id, maps
    if (metric_ == VecSimMetric_Cosine || sq8_) {
        std::vector<char> tmp(stored_bytes_);
        toStored(blob, tmp.data());
        return appendStored(tmp.data(), label);
    }
    return appendStored(blob, label);
}

// the append half of addVector, for a blob that already went through the storage preprocessing
int FlatIndex::appendStored(const void *stored, size_t label) {
    stageRow(stored);
    noteRow((uint32_t)count_, stored);
    const uint32_t id = (uint32_t)count_++;
    if (id_to_label_.size() < count_) {
        // grow metadata by whole blocks, like growByBlock() (brute_force.h:109-117)
        id_to_label_.resize(id_to_label_.size() + block_size_);
    }
    id_to_label_[id] = label;
    if (multi_) label_to_ids_[label].push_back(id);
    else label_to_id_[label] = id;
    if (staged_.size() >= ((size_t)8 << 20)) flush();
    return 1;
}

long FlatIndex::addBulk(const void *blobs, const size_t *labels, size_t n) {
    const size_t in_bytes = dim_ * type_size(type_);
    if (!multi_)
        for (size_t i = 0; i < n; i++)
            if (label_to_id_.count(labels[i])) return -1;
    if ((metric_ == VecSimMetric_Cosine || sq8_) && n >= 2048) {
        // the storage preprocessing (normalise / quantise) is per row and pure: a few host threads do it, the maps and the
        // staging buffer are then filled in order.  (SQ8 10 M x 768: 82 s of single-thread quantiser otherwise.)
        std::vector<char> pre(n * stored_bytes_);
        const size_t workers = std::min<size_t>(16, std::max<size_t>(1, std::thread::hardware_concurrency()));
        const size_t per = (n + workers - 1) / workers;
        std::vector<std::thread> pool;
        for (size_t w = 0; w < workers && w * per < n; w++)
            pool.emplace_back([&, w]() {
                for (size_t i = w * per; i < std::min(n, (w + 1) * per); i++)
                    toStored(static_cast<const char *>(blobs) + i * in_bytes, pre.data() + i * stored_bytes_);
            });
        for (auto &th : pool) th.join();
        for (size_t i = 0; i < n; i++) {
            if (!multi_ && label_to_id_.count(labels[i])) addVector(static_cast<const char *>(blobs) + i * in_bytes, labels[i]);   // a label repeated inside the batch: the overwrite path
            else appendStored(pre.data() + i * stored_bytes_, labels[i]);
        }
        return flush() ? -1 : (long)n;
    }
    for (size_t i = 0; i < n; i++) addVector(static_cast<const char *>(blobs) + i * in_bytes, labels[i]);
    return flush() ? -1 : (long)n;
}

long FlatIndex::addSynthetic(size_t n, uint64_t seed) {
    // device-generated rows are stored as generated: fp Cosine would need normalisation, so only int8 Cosine
    // (norm appended by the fill kernel) is accepted among the Cosine indexes
    if (type_ == VecSimType_FLOAT64 || type_ == VecSimType_UINT8 || multi_ || sq8_) return -1;
    if (metric_ == VecSimMetric_Cosine && type_ != VecSimType_INT8) return -1;
    if (flush()) return -1;
    const size_t first = count_;
    for (size_t i = 0; i < n; i++)
        if (label_to_id_.count(first + i)) return -1;
    if (vsgpu_table_append_synthetic(table_, n, seed)) return -1;
    count_ += n;
    size_t cap = ((count_ + block_size_ - 1) / block_size_) * block_size_;
    id_to_label_.resize(cap);
    label_to_id_.reserve(count_);
    for (size_t i = 0; i < n; i++) {
        id_to_label_[first + i] = first + i;
        label_to_id_[first + i] = (uint32_t)(first + i);
    }
    return (long)n;
}

// removeVector (brute_force.h:196-224): move the last row into the hole, shrink by one
void FlatIndex::removeRow(uint32_t id) {
    const uint32_t last = (uint32_t)(--count_);
    if (!nan_ids_.empty()) {  // the last row's flag travels with it into the hole
        const bool last_flag = nan_ids_.erase(last) > 0;
        nan_ids_.erase(id);
        if (last_flag && id != last) nan_ids_.insert(id);
    }
    if (id != last) {
        const size_t last_label = id_to_label_[last];
        id_to_label_[id] = last_label;
        if (multi_) {
            // replaceIdOfLabel (brute_force_multi.h:241-263): the LAST occurrence of the moved id
            auto &v = label_to_ids_.at(last_label);
            for (size_t i = v.size(); i-- > 0;)
                if (v[i] == last) {
                    v[i] = id;
                    break;
                }
        } else {
            label_to_id_[last_label] = id;
        }
        vsgpu_table_move(table_, id, last);
    }
    vsgpu_table_truncate(table_, count_);
    if (count_ % block_size_ == 0) {
        // shrinkByBlock (brute_force.h:119-137): keep at most one spare block of metadata
        if (count_ == 0) id_to_label_.clear();
        else if (id_to_label_.size() >= count_ + 2 * block_size_) id_to_label_.resize(id_to_label_.size() - block_size_);
        id_to_label_.shrink_to_fit();
    }
}

int FlatIndex::deleteVector(size_t label) {
    if (multi_) {
        auto it = label_to_ids_.find(label);
        if (it == label_to_ids_.end()) return 0;
        if (flush()) return 0;
        int removed = 0;
        // brute_force_multi.h:135-152: walk the label's id list while removals may rewrite its tail
        for (size_t i = 0; i < it->second.size(); i++) {
            removeRow(it->second[i]);
            removed++;
        }
        label_to_ids_.erase(label);
        return removed;
    }
    auto it = label_to_id_.find(label);
    if (it == label_to_id_.end()) return 0;
    if (flush()) return 0;
    const uint32_t id = it->second;
    label_to_id_.erase(it);
    removeRow(id);
    return 1;
}

int FlatIndex::readRow(uint32_t id, void *stored_blob) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);
    if (id >= count_ || flush()) return -1;
    return vsgpu_table_read(table_, id, stored_blob);
}
int FlatIndex::readRows(uint32_t first, size_t n, void *stored_blobs) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);
    if ((size_t)first + n > count_ || flush()) return -1;
    return vsgpu_table_read_range(table_, first, n, stored_blobs);
}
// (multi-value shards of a sharded index: a label's id list loses one entry / gains one; its order only matters to a local
// deleteVector, which a sharded index never calls -- it deletes by global id)
void FlatIndex::forgetIdOfLabel(size_t label, uint32_t id) {
    auto f = label_to_ids_.find(label);
    if (f == label_to_ids_.end()) return;
    auto &v = f->second;
    for (size_t i = v.size(); i-- > 0;)
        if (v[i] == id) {
            v.erase(v.begin() + (std::ptrdiff_t)i);
            break;
        }
    if (v.empty()) label_to_ids_.erase(f);
}
int FlatIndex::overwriteRow(uint32_t id, const void *stored_blob, size_t new_label) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);
    if (id >= count_ || flush()) return -1;
    if (multi_) {
        forgetIdOfLabel(id_to_label_[id], id);
        id_to_label_[id] = new_label;
        label_to_ids_[new_label].push_back(id);
        noteRow(id, stored_blob);
        return vsgpu_table_write(table_, id, stored_blob);
    }
    label_to_id_.erase(id_to_label_[id]);
    id_to_label_[id] = new_label;
    label_to_id_[new_label] = id;
    noteRow(id, stored_blob);
    return vsgpu_table_write(table_, id, stored_blob);
}
int FlatIndex::dropLastRow() {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);
    if (count_ == 0 || flush()) return -1;
    const uint32_t last = (uint32_t)(count_ - 1);
    if (multi_) {
        forgetIdOfLabel(id_to_label_[last], last);
        removeRow(last);
        return 0;
    }
    auto f = label_to_id_.find(id_to_label_[last]);
    if (f != label_to_id_.end() && f->second == last) label_to_id_.erase(f);  // (relabelled elsewhere: keep that entry)
    removeRow(last);
    return 0;
}
long FlatIndex::storedVectors(size_t label, void *out, size_t cap_bytes) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);  // readers may call concurrently (vec_sim.h contract)
    std::vector<uint32_t> ids;
    if (multi_) {
        auto f = label_to_ids_.find(label);
        if (f != label_to_ids_.end()) ids = f->second;
    } else {
        auto f = label_to_id_.find(label);
        if (f != label_to_id_.end()) ids.push_back(f->second);
    }
    if (ids.size() * stored_bytes_ > cap_bytes || flush()) return -1;
    for (size_t i = 0; i < ids.size(); i++)
        if (vsgpu_table_read(table_, ids[i], (char *)out + i * stored_bytes_)) return -1;
    return (long)ids.size();
}

// ---- queries ----
std::vector<char> FlatIndex::preprocessQuery(const void *query) const {
    std::vector<char> q(query_bytes_);
    toQuery(query, q.data());
    return q;
}

// The sequential heap of brute_force.h:257-288 replayed over the GPU's candidate rows (all rows
// with score <= T_k, ascending internal id): insert iff score < heap top or heap not full; evict the
// largest (score,label).  SURVEY.md §8a row A10 proves this equals the full scan.
void FlatIndex::replay(const uint32_t *ids, const double *scores, size_t n, size_t k, VecSimQueryReply *rep) const {
    using Item = std::pair<double, size_t>;
    RefMaxHeap<Item> heap;  // the reference's max-heap on (score, label), gnu++20 pair order (ref_heap.h)
    double upper = std::numeric_limits<double>::lowest();
    for (size_t i = 0; i < n; i++) {
        const double s = scores[i];
        if (s < upper || heap.size() < k) {
            heap.emplace(s, id_to_label_[ids[i]]);
            if (heap.size() > k) heap.pop();
            upper = heap.top().first;
        }
    }
    rep->results.resize(heap.size());
    for (size_t i = rep->results.size(); i-- > 0;) {
        rep->results[i].score = heap.top().first;
        rep->results[i].id = heap.top().second;
        heap.pop();
    }
}

// Multi-value replay: the same scan with an updatable max-heap keyed by label (utils/updatable_heap.h:
// 20-113): a label keeps its lowest score; eviction removes the largest score, the largest label among
// equal scores.
void FlatIndex::replayMulti(const uint32_t *ids, const double *scores, size_t n, size_t k, VecSimQueryReply *rep) const {
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
    for (size_t i = 0; i < n; i++) {
        const double s = scores[i];
        if (s < upper || node_of.size() < k) {
            const size_t label = id_to_label_[ids[i]];
            auto f = node_of.find(label);
            if (f == node_of.end()) {
                node_of.emplace(label, by_score.emplace(s, label));
            } else if (f->second->first > s) {
                by_score.erase(f->second);
                f->second = by_score.emplace(s, label);
            }
            if (node_of.size() > k) {
                auto t = top_it();
                node_of.erase(t->second);
                by_score.erase(t);
            }
            upper = top_it()->first;
        }
    }
    rep->results.resize(node_of.size());
    for (size_t i = rep->results.size(); i-- > 0;) {
        auto t = top_it();
        rep->results[i].score = t->first;
        rep->results[i].id = t->second;
        node_of.erase(t->second);
        by_score.erase(t);
    }
}

size_t FlatIndex::distinctLabels(const uint32_t *ids, size_t n) const {
    std::unordered_set<size_t> seen;
    for (size_t i = 0; i < n; i++) seen.insert(id_to_label_[ids[i]]);
    return seen.size();
}

bool FlatIndex::queryMayScoreNaN(const void *raw_query) const {
    std::vector<char> q = preprocessQuery(raw_query);
    return mayScoreNaN(q.data());
}

int FlatIndex::allScores(const void *processed_query, std::vector<double> &scores) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);  // readers may call concurrently (vec_sim.h contract)
    if (flush()) return -1;
    scores.resize(count_);
    if (count_ == 0) return 0;
    return vsgpu_scores(table_, processed_query, 0, count_, scores.data());
}

int FlatIndex::iteratorScores(const void *processed_query, std::vector<std::pair<double, size_t>> &out) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);
    if (flush()) return -1;
    out.clear();
    if (count_ == 0) return 0;
    std::unique_ptr<double[]> s(new double[count_]);  // (not value-initialised: tens of MB at 10 M rows)
    if (vsgpu_scores(table_, processed_query, 0, count_, s.get())) return -1;
    if (multi_) {
        // bfm_batch_iterator.h:24-53: lowest score per label, emitted in the hash map's iteration order
        std::unordered_map<size_t, double> best;
        for (size_t i = 0; i < count_; i++) {
            const size_t label = id_to_label_[i];
            auto f = best.find(label);
            if (f == best.end()) best.emplace(label, s[i]);
            else if (f->second > s[i]) f->second = s[i];
        }
        out.reserve(best.size());
        for (auto &p : best) out.emplace_back(p.second, p.first);
    } else {
        // bfs_batch_iterator.h:24-41
        out.resize(count_);
        const size_t workers = count_ >= ((size_t)1 << 20) ? std::min<size_t>(8, std::max<size_t>(1, std::thread::hardware_concurrency())) : 1;
        auto fill = [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) out[i] = std::make_pair(s[i], id_to_label_[i]);
        };
        if (workers > 1) {
            std::vector<std::thread> pool;
            const size_t per = (count_ + workers - 1) / workers;
            for (size_t w = 0; w < workers; w++)
                if (w * per < count_) pool.emplace_back(fill, w * per, std::min(count_, (w + 1) * per));
            for (auto &th : pool) th.join();
        } else {
            fill(0, count_);
        }
    }
    return 0;
}

vsgpu_scorebuf *FlatIndex::iteratorDeviceBegin(const void *processed_query) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);
    if (multi_ || flush() || count_ == 0) return nullptr;  // multi-value needs the per-label minimum: host path
    return vsgpu_scorebuf_create(table_, processed_query);
}
int FlatIndex::iteratorDeviceNext(vsgpu_scorebuf *b, size_t k, size_t cap, uint32_t *ids, double *scores, uint32_t *count) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);
    return vsgpu_scorebuf_next(b, k, cap, ids, scores, count);
}
int FlatIndex::iteratorDeviceRetire(vsgpu_scorebuf *b, const uint32_t *rows, size_t m) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);
    return vsgpu_scorebuf_retire(b, rows, m);
}
int FlatIndex::iteratorDeviceRead(vsgpu_scorebuf *b, double *all) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);
    return vsgpu_scorebuf_read(b, all);
}
void FlatIndex::iteratorDeviceEnd(vsgpu_scorebuf *b) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);
    vsgpu_scorebuf_destroy(b);
}

int FlatIndex::topKQueryBatch(const void *queries, size_t nq, size_t stride, size_t k, VecSimQueryParams *qp,
                              VecSimQueryReply_Order order, VecSimQueryReply **out) {
    // readers may call concurrently (vec_sim.h contract): the first one runs on the index's own context, one that finds
    // it busy on a reader lane (flat_index.h), the rest take turns
    std::unique_lock<std::recursive_mutex> gpu_lock(gpu_mu_, std::defer_lock);
    Lane *lane = nullptr;
    if (!gpu_lock.try_lock()) {
        if (!multi_ && staged_rows_.load(std::memory_order_acquire) == 0) lane = tryLane();
        if (!lane) gpu_lock.lock();
    }
    struct LaneRelease {
        Lane *l;
        ~LaneRelease() {
            if (l) l->mu.unlock();
        }
    } lane_release{lane};
    vsgpu_table *tbl = table_;
    if (lane) {
        vsgpu_table_view_sync(lane->view);
        tbl = lane->view;
    }
    void *tctx = qp ? qp->timeoutCtx : nullptr;
    if (!lane) last_mode_ = STANDARD_KNN;   // (a lane reader holds no lock: the plain member is the lock holder's to write)
    if (nq == 0) return 0;
    std::vector<VecSimQueryReply *> reps(nq);
    for (auto &r : reps) r = new VecSimQueryReply();
    auto finish = [&]() {
        for (size_t q = 0; q < nq; q++) out[q] = reps[q];
        return 0;
    };
    if (k == 0) return finish();
    // the reference polls the timeout callback once per scanned vector (brute_force.h:265); the GPU
    // path polls at launch granularity: before the scan and after it
    if (timed_out(tctx)) {
        for (auto *r : reps) r->code = VecSim_QueryReply_TimedOut;
        return finish();
    }
    if (!lane && flush()) {
        for (auto *r : reps) delete r;
        return -1;
    }
    if (count_ == 0) return finish();
    std::vector<char> qbuf = packQueries(queries, nq, stride);
    if (multi_) {
        // Rows with score <= the k'-th smallest ROW score, k' grown until they cover k distinct labels (or all
        // rows): a superset of the rows at or below the k-th smallest per-label minimum, which is what the
        // label-keyed replay needs.
        const size_t n_labels = label_to_ids_.size();
        for (size_t q = 0; q < nq; q++) {
            size_t kr = std::min(count_, std::max<size_t>(k, 1));
            std::vector<uint32_t> ids, cnt(1);
            std::vector<double> sc;
            for (;;) {
                const size_t cap = std::max<size_t>(2 * kr, kr + 64);
                ids.resize(cap);
                sc.resize(cap);
                int rc = vsgpu_topk(table_, qbuf.data() + q * query_bytes_, 1, query_bytes_, kr, cap, ids.data(), sc.data(),
                                    cnt.data());
                if (rc) {
                    for (auto *r : reps) delete r;
                    return rc;
                }
                if (cnt[0] == VSGPU_COUNT_OVERFLOW) {  // heavy ties: take every row's score
                    ids.resize(count_);
                    sc.resize(count_);
                    rc = vsgpu_scores(table_, qbuf.data() + q * query_bytes_, 0, count_, sc.data());
                    if (rc) {
                        for (auto *r : reps) delete r;
                        return rc;
                    }
                    for (size_t i = 0; i < count_; i++) ids[i] = (uint32_t)i;
                    cnt[0] = (uint32_t)count_;
                    break;
                }
                if (kr >= count_ || distinctLabels(ids.data(), cnt[0]) >= std::min(k, n_labels)) break;
                kr = std::min(count_, kr * 4);
            }
            replayMulti(ids.data(), sc.data(), cnt[0], k, reps[q]);
            if (order == BY_ID) sort_reply(reps[q], BY_ID);
        }
        if (timed_out(tctx)) {
            for (auto *r : reps) {
                r->results.clear();
                r->code = VecSim_QueryReply_TimedOut;
            }
        }
        return finish();
    }
    const size_t kk = std::min(k, count_);
    const size_t cap = std::max<size_t>(2 * kk, kk + 64);
    std::vector<uint32_t> ids(nq * cap), counts(nq);
    std::vector<double> sc(nq * cap);
    // queries that can meet a NaN score while the reference's heap is still filling (see nan_ids_) take the every-row replay
    const bool nan_head = !nan_ids_.empty() && *nan_ids_.begin() < k;
    std::vector<char> every_row(nq, (char)nan_head);
    size_t n_every = nan_head ? nq : 0;
    if (!nan_head)
        for (size_t q = 0; q < nq; q++)
            if (mayScoreNaN(qbuf.data() + q * query_bytes_)) every_row[q] = 1, n_every++;
    // a registered timeout callback is also polled between the launches of the GPU pass (behind the probe and behind the scan)
    struct PollScope {
        vsgpu_ctx *c;
        PollScope(vsgpu_ctx *cc, void *user) : c(cc) {
            // (whenever a callback is registered: the reference calls it with a NULL context as well, brute_force.h:265)
            if (globals().timeout_cb) vsgpu_set_poll(c, [](void *u) { return timed_out(u) ? 1 : 0; }, user);
        }
        ~PollScope() { vsgpu_set_poll(c, nullptr, nullptr); }
    } poll_scope(lane ? lane->ctx : ctx_, tctx);
    int rc = n_every == nq ? 0 : vsgpu_topk(tbl, qbuf.data(), nq, query_bytes_, k, cap, ids.data(), sc.data(), counts.data());
    if (rc == VSGPU_ERR_TIMEOUT) {
        for (auto *r : reps) r->code = VecSim_QueryReply_TimedOut;
        return finish();
    }
    if (n_every)
        for (size_t q = 0; q < nq; q++)
            if (every_row[q]) counts[q] = VSGPU_COUNT_OVERFLOW;
    if (rc) {
        log("warning", "GPU top-k failed: %s", vsgpu_last_error());
        for (auto *r : reps) delete r;
        return rc;
    }
    if (timed_out(tctx)) {
        for (auto *r : reps) r->code = VecSim_QueryReply_TimedOut;
        return finish();
    }
    // replies are independent: wide batches replay on a few host threads (heap replay + label lookups, ~4 us each)
    auto replay_range = [&](size_t q0, size_t q1) {
        for (size_t q = q0; q < q1; q++) {
            if (counts[q] == VSGPU_COUNT_OVERFLOW) continue;
            replay(ids.data() + q * cap, sc.data() + q * cap, counts[q], k, reps[q]);
            if (order == BY_ID) sort_reply(reps[q], BY_ID);
        }
    };
    // (spawning threads costs ~0.3 ms: only worth it for wide AND deep batches, e.g. 256 queries x top-100)
    const size_t workers = nq * k >= 16384 ? std::min<size_t>(8, std::max<size_t>(1, std::thread::hardware_concurrency())) : 1;
    if (workers > 1) {
        std::vector<std::thread> pool;
        const size_t per = (nq + workers - 1) / workers;
        for (size_t w = 0; w < workers; w++)
            if (w * per < nq) pool.emplace_back(replay_range, w * per, std::min(nq, (w + 1) * per));
        for (auto &th : pool) th.join();
    } else {
        replay_range(0, nq);
    }
    std::vector<double> all;
    std::vector<uint32_t> all_ids;
    for (size_t q = 0; q < nq; q++) {
        if (counts[q] != VSGPU_COUNT_OVERFLOW) continue;
        // more than `cap` rows tie at the k-th score: replay over every row's GPU score
        rc = vsgpu_scores(tbl, qbuf.data() + q * query_bytes_, 0, count_, (all.resize(count_), all.data()));
        if (rc) {
            for (auto *r : reps) delete r;
            return rc;
        }
        if (all_ids.size() != count_) {
            all_ids.resize(count_);
            for (size_t i = 0; i < count_; i++) all_ids[i] = (uint32_t)i;
        }
        replay(all_ids.data(), all.data(), count_, k, reps[q]);
        if (order == BY_ID) sort_reply(reps[q], BY_ID);
    }
    return finish();
}

// preprocess queries into one contiguous buffer (Cosine: copy + normalise, vec_sim_index.h:397-402)
std::vector<char> FlatIndex::packQueries(const void *queries, size_t nq, size_t stride) const {
    std::vector<char> qbuf(nq * query_bytes_);
    const size_t in_bytes = dim_ * type_size(type_);
    for (size_t q = 0; q < nq; q++) {
        char *dst = qbuf.data() + q * query_bytes_;
        (void)in_bytes;
        toQuery(static_cast<const char *>(queries) + q * stride, dst);
    }
    return qbuf;
}

int FlatIndex::topKCandidates(const void *queries, size_t nq, size_t stride, size_t k, size_t cap, uint32_t *ids,
                              size_t *labels, double *scores, uint32_t *counts) {
    // concurrent readers (the sharded index's reader threads): the first runs on the index's own context, one that finds it
    // busy on a reader lane -- as in topKQueryBatch
    std::unique_lock<std::recursive_mutex> gpu_lock(gpu_mu_, std::defer_lock);
    Lane *lane = nullptr;
    if (!gpu_lock.try_lock()) {
        if (!multi_ && staged_rows_.load(std::memory_order_acquire) == 0) lane = tryLane();
        if (!lane) gpu_lock.lock();
    }
    struct LaneRelease {
        Lane *l;
        ~LaneRelease() {
            if (l) l->mu.unlock();
        }
    } lane_release{lane};
    vsgpu_table *tbl = table_;
    if (lane) {
        vsgpu_table_view_sync(lane->view);
        tbl = lane->view;
    }
    if (!lane) last_mode_ = STANDARD_KNN;
    if (nq == 0) return 0;
    if (!lane && flush()) return -1;
    if (k == 0 || count_ == 0) {
        for (size_t q = 0; q < nq; q++) counts[q] = 0;
        return 0;
    }
    std::vector<char> qbuf = packQueries(queries, nq, stride);
    if (multi_) {
        // multi-value shard: the rows at or below the k'-th smallest local ROW score, k' grown until they cover k distinct local
        // labels (or every row).  The local k-th smallest per-label minimum bounds the global one from above (a label's global
        // minimum is at most its local one), so this is a superset of what the label-keyed replay over the union needs.
        const size_t n_labels = label_to_ids_.size();
        std::vector<uint32_t> lid, cnt(1);
        std::vector<double> lsc;
        for (size_t q = 0; q < nq; q++) {
            size_t kr = std::min(count_, std::max<size_t>(k, 1));
            size_t c1 = std::max<size_t>(2 * kr, kr + 64);
            for (;;) {
                c1 = std::min(count_, std::max(c1, std::max<size_t>(2 * kr, kr + 64)));
                lid.resize(c1);
                lsc.resize(c1);
                int rc = vsgpu_topk(tbl, qbuf.data() + q * query_bytes_, 1, query_bytes_, kr, c1, lid.data(), lsc.data(), cnt.data());
                if (rc) return rc;
                if (cnt[0] == VSGPU_COUNT_OVERFLOW) {
                    // more rows tie at the kr-th score than c1 holds (duplicated vectors under distinct labels, low-dim int8): the
                    // room has to grow HERE -- it does not depend on the caller's `cap`, so a caller retrying with a larger cap
                    // would meet the same overflow for ever (round-3 advisor finding)
                    if (c1 >= count_) break;
                    c1 = std::min(count_, c1 * 8);
                    continue;
                }
                if (kr >= count_ || distinctLabels(lid.data(), cnt[0]) >= std::min(k, n_labels)) break;
                kr = std::min(count_, kr * 4);
            }
            if (cnt[0] == VSGPU_COUNT_OVERFLOW || cnt[0] > cap) {
                counts[q] = VSGPU_COUNT_OVERFLOW;
                continue;
            }
            counts[q] = cnt[0];
            for (uint32_t i = 0; i < cnt[0]; i++) {
                ids[q * cap + i] = lid[i];
                scores[q * cap + i] = lsc[i];
                labels[q * cap + i] = id_to_label_[lid[i]];
            }
        }
        return 0;
    }
    int rc = vsgpu_topk(tbl, qbuf.data(), nq, query_bytes_, k, cap, ids, scores, counts);
    if (rc) return rc;
    for (size_t q = 0; q < nq; q++) {
        if (counts[q] == VSGPU_COUNT_OVERFLOW) continue;
        for (uint32_t i = 0; i < counts[q]; i++) labels[q * cap + i] = id_to_label_[ids[q * cap + i]];
    }
    return 0;
}

VecSimQueryReply *FlatIndex::topKQuery(const void *query, size_t k, VecSimQueryParams *qp) {
    VecSimQueryReply *rep = nullptr;
    int rc = topKQueryBatch(query, 1, 0, k, qp, BY_SCORE, &rep);
    if (rc) {
        // no CPU fallback: report loudly and return an empty reply flagged as failed
        std::fprintf(stderr, "vecsim_amd: GPU top-k query failed: %s\n", vsgpu_last_error());
        rep = new VecSimQueryReply();
        rep->code = VecSim_QueryReply_TimedOut;
    }
    return rep;
}

VecSimQueryReply *FlatIndex::rangeQuery(const void *query, double radius, VecSimQueryParams *qp,
                                        VecSimQueryReply_Order order) {
    // readers may call concurrently (vec_sim.h contract): a reader that finds the index's own context busy runs on a reader lane
    // (a view of the same rows with its own stream and scratch), like a top-k batch does
    std::unique_lock<std::recursive_mutex> gpu_lock(gpu_mu_, std::defer_lock);
    Lane *lane = nullptr;
    if (!gpu_lock.try_lock()) {
        if (staged_rows_.load(std::memory_order_acquire) == 0) lane = tryLane();
        if (!lane) gpu_lock.lock();
    }
    struct LaneRelease {
        Lane *l;
        ~LaneRelease() {
            if (l) l->mu.unlock();
        }
    } lane_release{lane};
    vsgpu_table *tbl = table_;
    if (lane) {
        vsgpu_table_view_sync(lane->view);
        tbl = lane->view;
    }
    auto *rep = new VecSimQueryReply();
    void *tctx = qp ? qp->timeoutCtx : nullptr;
    if (!lane) last_mode_ = RANGE_QUERY;   // (a lane reader holds no lock: the plain member is the lock holder's to write)
    if (timed_out(tctx)) {
        rep->code = VecSim_QueryReply_TimedOut;
        return rep;
    }
    if ((!lane && flush()) || count_ == 0) return rep;
    std::vector<char> q = preprocessQuery(query);
    size_t cap = 1024;
    std::vector<uint32_t> ids;
    std::vector<double> sc;
    uint32_t cnt = 0;
    for (;;) {
        ids.resize(cap);
        sc.resize(cap);
        int rc = vsgpu_range(tbl, q.data(), radius, cap, ids.data(), sc.data(), &cnt);
        if (rc) {
            std::fprintf(stderr, "vecsim_amd: GPU range query failed: %s\n", vsgpu_last_error());
            rep->code = VecSim_QueryReply_TimedOut;
            return rep;
        }
        if (cnt != VSGPU_COUNT_OVERFLOW) break;
        if (cap >= count_) {  // cannot overflow at full capacity; defensive
            cnt = 0;
            break;
        }
        cap = std::min(count_, cap * 16);
    }
    if (timed_out(tctx)) {
        rep->code = VecSim_QueryReply_TimedOut;
        return rep;
    }
    if (multi_) {
        // unique results container: one entry per label, its lowest score (brute_force_multi.h:100-104)
        std::unordered_map<size_t, size_t> slot;
        for (uint32_t i = 0; i < cnt; i++) {
            const size_t label = id_to_label_[ids[i]];
            auto f = slot.find(label);
            if (f == slot.end()) {
                slot.emplace(label, rep->results.size());
                rep->results.push_back(VecSimQueryResult{label, sc[i]});
            } else if (sc[i] < rep->results[f->second].score) {
                rep->results[f->second].score = sc[i];
            }
        }
    } else {
        rep->results.resize(cnt);
        for (uint32_t i = 0; i < cnt; i++) {
            rep->results[i].id = id_to_label_[ids[i]];
            rep->results[i].score = sc[i];
        }
    }
    sort_reply(rep, order);
    return rep;
}

double FlatIndex::getDistanceFrom(size_t label, const void *blob) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);  // readers may call concurrently (vec_sim.h contract)
    // "Unsafe": the blob is used as given (caller normalises for Cosine), brute_force_single.h:202-212
    std::vector<char> q(query_bytes_);
    if (sq8_ && metric_ != VecSimMetric_Cosine) {   // no normalisation involved: the query preprocessing of the index
        toQuery(blob, q.data());
    } else if (sq8_ && type_ == VecSimType_FLOAT16) {
        std::memcpy(q.data(), blob, dim_ * 2);
        std::vector<float> wide(dim_);
        for (size_t i = 0; i < dim_; i++) {
            uint16_t h;
            std::memcpy(&h, q.data() + 2 * i, 2);
            wide[i] = fp16_widen(h);
        }
        sq8_query_meta_f16(wide.data(), dim_, metric_, q.data());
    } else if (sq8_) {   // an fp32 vector as given; only the query metadata (y_sum) is appended
        std::memcpy(q.data(), blob, dim_ * sizeof(float));
        sq8_query_blob(reinterpret_cast<const float *>(q.data()), dim_, metric_, reinterpret_cast<float *>(q.data()));
    } else {
        std::memcpy(q.data(), blob, query_bytes_);
    }
    if (multi_) {  // lowest distance over the label's vectors (brute_force_multi.h:224-239)
        auto f = label_to_ids_.find(label);
        if (f == label_to_ids_.end() || flush()) return std::numeric_limits<double>::quiet_NaN();
        std::vector<double> s(f->second.size());
        if (vsgpu_scores_of(table_, q.data(), f->second.data(), f->second.size(), s.data()))
            return std::numeric_limits<double>::quiet_NaN();
        double best = std::numeric_limits<double>::infinity();
        for (double d : s) best = (best < d) ? best : d;
        return best;
    }
    auto it = label_to_id_.find(label);
    if (it == label_to_id_.end()) return std::numeric_limits<double>::quiet_NaN();
    if (flush()) return std::numeric_limits<double>::quiet_NaN();
    uint32_t id = it->second;
    double s = std::numeric_limits<double>::quiet_NaN();
    if (vsgpu_scores_of(table_, q.data(), &id, 1, &s)) return std::numeric_limits<double>::quiet_NaN();
    return s;
}

// scripts/BF_batches_clf.py decision tree as evaluated in brute_force.h:380-451
bool FlatIndex::preferAdHocSearch(size_t subsetSize, size_t k, bool initial_check) {
    (void)k;
    const size_t n = count_;
    subsetSize = std::min(subsetSize, n);
    const size_t d = dim_;
    // the ratio is taken over LABELS, not vectors (brute_force.h:390): they differ on multi-value indexes
    const float r = (n == 0) ? 0.0f : (float)subsetSize / (float)indexLabelCount();
    bool adhoc;
    if (n <= 5500) adhoc = true;
    else if (d <= 300) adhoc = (r <= 0.15) || (r <= 0.35 && d > 75 && n <= 550000);
    else adhoc = (r <= 0.55) || (d > 750 && r <= 0.75);
    last_mode_ = adhoc ? (initial_check ? HYBRID_ADHOC_BF : HYBRID_BATCHES_TO_ADHOC_BF) : HYBRID_BATCHES;
    return adhoc;
}

VecSimIndexBasicInfo FlatIndex::basicInfo() const {
    VecSimIndexBasicInfo b{};
    b.algo = VecSimAlgo_BF;
    b.metric = metric_;
    b.type = type_;
    b.isMulti = multi_;
    b.isTiered = false;
    b.isDisk = false;
    b.blockSize = block_size_;
    b.dim = dim_;
    return b;
}
VecSimIndexStatsInfo FlatIndex::statsInfo() const {
    VecSimIndexStatsInfo s{};
    s.memory = id_to_label_.capacity() * sizeof(size_t) + staged_.capacity() + (table_ ? vsgpu_table_bytes(table_) : 0);
    return s;
}
VecSimIndexDebugInfo FlatIndex::debugInfo() const {
    VecSimIndexDebugInfo d{};
    d.commonInfo.basicInfo = basicInfo();
    d.commonInfo.indexSize = count_;
    d.commonInfo.indexLabelCount = indexLabelCount();
    d.commonInfo.memory = statsInfo().memory;
    d.commonInfo.lastMode = last_mode_;
    return d;
}

VecSimBatchIterator *FlatIndex::newBatchIterator(const void *query, VecSimQueryParams *qp) {
    auto *it = new VecSimBatchIterator();
    it->index = this;
    it->query = preprocessQuery(query);
    it->timeout_ctx = qp ? qp->timeoutCtx : nullptr;
    it->label_count = indexLabelCount();
    return it;
}

}  // namespace vsa
