// hnsw_index.cpp -- see hnsw_index.h
#include "hnsw_index.h"
#include "host_tier.h"
#include "ref_heap.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <queue>
#include <thread>

#include "blob_prep.h"

namespace vsa {

static constexpr uint32_t NONE = 0xFFFFFFFFu;

HnswIndex *HnswIndex::create(const HNSWParams &p, void *logCtx) {
    if (p.dim == 0 || p.metric > VecSimMetric_Cosine) return nullptr;
    if ((unsigned)p.type > (unsigned)VecSimType_UINT8) {
        std::fprintf(stderr, "vecsim_amd: HNSW supports fp32/fp64/bf16/fp16/int8/uint8 (L2, IP, Cosine)\n");
        return nullptr;
    }
    const size_t M = p.M ? p.M : HNSW_DEFAULT_M;
    if (M <= 1 || M > 32) {
        std::fprintf(stderr, "vecsim_amd: HNSW M must be in [2, 32]\n");
        return nullptr;
    }
    int dev = globals().device;
    if (dev < 0) {
        const char *e = std::getenv("VECSIM_GPU_DEVICE");
        dev = e ? std::atoi(e) : 0;
    }
    vsgpu_ctx *ctx = vsgpu_ctx_create(dev);
    if (!ctx) return nullptr;
    HnswIndex *ix = new HnswIndex();
    ix->type_ = p.type;
    ix->metric_ = p.metric;
    ix->dim_ = p.dim;
    ix->blob_bytes_ = blob_bytes(p.type, p.dim, p.metric);
    ix->elem_bytes_ = type_size(p.type);
    ix->block_size_ = p.blockSize ? p.blockSize : DEFAULT_BLOCK_SIZE;
    ix->M_ = M;
    ix->M0_ = 2 * M;
    ix->ef_c_ = std::max<size_t>(p.efConstruction ? p.efConstruction : HNSW_DEFAULT_EF_C, M);  // hnsw.h:1633-1634
    ix->ef_ = p.efRuntime ? p.efRuntime : HNSW_DEFAULT_EF_RT;
    ix->epsilon_ = p.epsilon > 0.0 ? p.epsilon : HNSW_DEFAULT_EPSILON;
    ix->mult_ = 1.0 / std::log(1.0 * (double)M);                                               // hnsw.h:1646
    ix->log_ctx_ = logCtx;
    ix->ctx_ = ctx;
    // (the search kernel scores rows with the lane program of the host's tier as well; the scalar tier is not wired into it)
    ix->tier_ = resolve_tier((int)p.type);
    if (ix->tier_ == VSGPU_TIER_SCALAR) ix->tier_ = VSGPU_TIER_AVX512;
    ix->table_ = vsgpu_table_create(ctx, (int)p.type, (int)p.metric, ix->tier_, p.dim, ix->blob_bytes_);
    ix->graph_ = ix->table_ ? vsgpu_graph_create(ix->table_, M) : nullptr;
    if (!ix->table_ || !ix->graph_) {
        delete ix;
        return nullptr;
    }
    ix->multi_ = p.multi;
    vsgpu_graph_set_multi(ix->graph_, p.multi ? 1 : 0);
    {   // build mode (hnsw_index.h)
        const char *e = std::getenv("VECSIM_GPU_HNSW_BUILD");
        const bool want_fast = e && !std::strcmp(e, "fast");
        const bool have = ix->ref_eval_.init((int)p.type, (int)p.metric, ix->tier_, p.dim);
        ix->ref_add_ = have && !want_fast;
        ix->ref_bulk_ = have && e && !std::strcmp(e, "reference");
        if (e && !std::strcmp(e, "reference") && !have)
            std::fprintf(stderr, "vecsim_amd: VECSIM_GPU_HNSW_BUILD=reference: no host walker for this type / tier (AVX512-FP16 half "
                                 "accumulators); building with the fast routine\n");
    }
    size_t n_lanes = 2;   // the index's own context + one reader lane; VECSIM_GPU_READER_LANES as for the Flat index
    if (const char *e = std::getenv("VECSIM_GPU_READER_LANES")) n_lanes = (size_t)std::max(1, std::min(8, std::atoi(e)));
    for (size_t i = 1; i < n_lanes; i++) {   // best effort: without lanes readers take turns
        auto lane = std::make_unique<Lane>();
        lane->ctx = vsgpu_ctx_create(dev);
        if (!lane->ctx) break;
        lane->view = vsgpu_table_view_create(ix->table_, lane->ctx);
        lane->graph = lane->view ? vsgpu_graph_view_create(ix->graph_, lane->view) : nullptr;
        if (!lane->graph) {
            if (lane->view) vsgpu_table_destroy(lane->view);
            vsgpu_ctx_destroy(lane->ctx);
            break;
        }
        ix->lanes_.push_back(std::move(lane));
    }
    return ix;
}

HnswIndex::~HnswIndex() {
    for (auto &l : lanes_) {
        if (l->graph) vsgpu_graph_destroy(l->graph);
        if (l->view) vsgpu_table_destroy(l->view);
        if (l->ctx) vsgpu_ctx_destroy(l->ctx);
    }
    if (graph_) vsgpu_graph_destroy(graph_);
    if (table_) vsgpu_table_destroy(table_);
    if (ctx_) vsgpu_ctx_destroy(ctx_);
}
HnswIndex::Lane *HnswIndex::tryLane() {
    for (auto &l : lanes_)
        if (l->mu.try_lock()) return l.get();
    return nullptr;
}
std::vector<vsgpu_ctx *> HnswIndex::gpus() {
    std::vector<vsgpu_ctx *> v{ctx_};
    for (auto &l : lanes_) v.push_back(l->ctx);
    return v;
}

// ---- construction-time distance (ingest only; queries never come here) ----
// Explicit intrinsics: at -O2 gcc 11 does not vectorise, and the "avx512f" target clone of a plain loop came out as
// scalar vsubss / vmulss / vaddss with a store and a reload per element -- 1.9 us per 768-dim distance, which is what bound
// the graph build (15 K distances per insert).  Any summation order will do here: the builder only ranks candidates.
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx512f"))) static float l2_build_avx512(const float *a, const float *b, size_t d) {
    __m512 s0 = _mm512_setzero_ps(), s1 = _mm512_setzero_ps(), s2 = _mm512_setzero_ps(), s3 = _mm512_setzero_ps();
    size_t i = 0;
    for (; i + 64 <= d; i += 64) {
        __m512 t0 = _mm512_sub_ps(_mm512_loadu_ps(a + i), _mm512_loadu_ps(b + i));
        __m512 t1 = _mm512_sub_ps(_mm512_loadu_ps(a + i + 16), _mm512_loadu_ps(b + i + 16));
        __m512 t2 = _mm512_sub_ps(_mm512_loadu_ps(a + i + 32), _mm512_loadu_ps(b + i + 32));
        __m512 t3 = _mm512_sub_ps(_mm512_loadu_ps(a + i + 48), _mm512_loadu_ps(b + i + 48));
        s0 = _mm512_fmadd_ps(t0, t0, s0);
        s1 = _mm512_fmadd_ps(t1, t1, s1);
        s2 = _mm512_fmadd_ps(t2, t2, s2);
        s3 = _mm512_fmadd_ps(t3, t3, s3);
    }
    for (; i + 16 <= d; i += 16) {
        __m512 t = _mm512_sub_ps(_mm512_loadu_ps(a + i), _mm512_loadu_ps(b + i));
        s0 = _mm512_fmadd_ps(t, t, s0);
    }
    if (i < d) {
        const __mmask16 m = (__mmask16)((1u << (d - i)) - 1u);
        __m512 t = _mm512_sub_ps(_mm512_maskz_loadu_ps(m, a + i), _mm512_maskz_loadu_ps(m, b + i));
        s1 = _mm512_fmadd_ps(t, t, s1);
    }
    return _mm512_reduce_add_ps(_mm512_add_ps(_mm512_add_ps(s0, s1), _mm512_add_ps(s2, s3)));
}
__attribute__((target("avx512f"))) static float dot_build_avx512(const float *a, const float *b, size_t d) {
    __m512 s0 = _mm512_setzero_ps(), s1 = _mm512_setzero_ps(), s2 = _mm512_setzero_ps(), s3 = _mm512_setzero_ps();
    size_t i = 0;
    for (; i + 64 <= d; i += 64) {
        s0 = _mm512_fmadd_ps(_mm512_loadu_ps(a + i), _mm512_loadu_ps(b + i), s0);
        s1 = _mm512_fmadd_ps(_mm512_loadu_ps(a + i + 16), _mm512_loadu_ps(b + i + 16), s1);
        s2 = _mm512_fmadd_ps(_mm512_loadu_ps(a + i + 32), _mm512_loadu_ps(b + i + 32), s2);
        s3 = _mm512_fmadd_ps(_mm512_loadu_ps(a + i + 48), _mm512_loadu_ps(b + i + 48), s3);
    }
    for (; i + 16 <= d; i += 16) s0 = _mm512_fmadd_ps(_mm512_loadu_ps(a + i), _mm512_loadu_ps(b + i), s0);
    if (i < d) {
        const __mmask16 m = (__mmask16)((1u << (d - i)) - 1u);
        s1 = _mm512_fmadd_ps(_mm512_maskz_loadu_ps(m, a + i), _mm512_maskz_loadu_ps(m, b + i), s1);
    }
    return _mm512_reduce_add_ps(_mm512_add_ps(_mm512_add_ps(s0, s1), _mm512_add_ps(s2, s3)));
}
__attribute__((target("avx2,fma"))) static float hsum256(__m256 v) {
    __m128 lo = _mm_add_ps(_mm256_castps256_ps128(v), _mm256_extractf128_ps(v, 1));
    lo = _mm_add_ps(lo, _mm_movehl_ps(lo, lo));
    lo = _mm_add_ss(lo, _mm_shuffle_ps(lo, lo, 1));
    return _mm_cvtss_f32(lo);
}
__attribute__((target("avx2,fma"))) static float l2_build_avx2(const float *a, const float *b, size_t d) {
    __m256 s0 = _mm256_setzero_ps(), s1 = _mm256_setzero_ps();
    size_t i = 0;
    for (; i + 16 <= d; i += 16) {
        __m256 t0 = _mm256_sub_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i));
        __m256 t1 = _mm256_sub_ps(_mm256_loadu_ps(a + i + 8), _mm256_loadu_ps(b + i + 8));
        s0 = _mm256_fmadd_ps(t0, t0, s0);
        s1 = _mm256_fmadd_ps(t1, t1, s1);
    }
    float s = hsum256(_mm256_add_ps(s0, s1));
    for (; i < d; i++) {
        const float t = a[i] - b[i];
        s += t * t;
    }
    return s;
}
__attribute__((target("avx2,fma"))) static float dot_build_avx2(const float *a, const float *b, size_t d) {
    __m256 s0 = _mm256_setzero_ps(), s1 = _mm256_setzero_ps();
    size_t i = 0;
    for (; i + 16 <= d; i += 16) {
        s0 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i), s0);
        s1 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i + 8), _mm256_loadu_ps(b + i + 8), s1);
    }
    float s = hsum256(_mm256_add_ps(s0, s1));
    for (; i < d; i++) s += a[i] * b[i];
    return s;
}
#endif
static float l2_build_plain(const float *a, const float *b, size_t d) {
    float s = 0;
    for (size_t i = 0; i < d; i++) {
        const float t = a[i] - b[i];
        s += t * t;
    }
    return s;
}
static float dot_build_plain(const float *a, const float *b, size_t d) {
    float s = 0;
    for (size_t i = 0; i < d; i++) s += a[i] * b[i];
    return s;
}
typedef float (*build_fn)(const float *, const float *, size_t);
static build_fn pick_build(bool l2) {
#if defined(__x86_64__)
    __builtin_cpu_init();   // (this runs from a static initialiser)
    if (__builtin_cpu_supports("avx512f")) return l2 ? l2_build_avx512 : dot_build_avx512;
    if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) return l2 ? l2_build_avx2 : dot_build_avx2;
#endif
    return l2 ? l2_build_plain : dot_build_plain;
}
static const build_fn g_l2_build = pick_build(true), g_dot_build = pick_build(false);
static float l2_build(const float *a, const float *b, size_t d) { return g_l2_build(a, b, d); }
static float ip_build(const float *a, const float *b, size_t d) { return 1.0f - g_dot_build(a, b, d); }
float HnswIndex::buildDistance(const float *a, const float *b) const {
    return metric_ == VecSimMetric_L2 ? l2_build(a, b, dim_) : ip_build(a, b, dim_);
}

uint32_t *HnswIndex::linksAt(uint32_t id, int level, uint32_t **count_word) {
    // level >= 1: block {count, links[M]}
    uint32_t *blk = upper_.data() + ((size_t)upper_off_[id] + (size_t)(level - 1)) * (M_ + 1);
    *count_word = blk;
    return blk + 1;
}

std::vector<char> HnswIndex::preprocess(const void *blob) const {
    std::vector<char> v(blob_bytes_);
    std::memcpy(v.data(), blob, dim_ * elem_bytes_);
    if (metric_ == VecSimMetric_Cosine) normalize_blob(v.data(), dim_, type_);
    return v;
}

// fp32 image of a stored blob for the builder's own distance routine
void HnswIndex::widen(const char *b, float *out) const {
    switch (type_) {
    case VecSimType_FLOAT32: std::memcpy(out, b, dim_ * 4); break;
    case VecSimType_FLOAT64:
        for (size_t i = 0; i < dim_; i++) { double d; std::memcpy(&d, b + 8 * i, 8); out[i] = (float)d; }
        break;
    case VecSimType_BFLOAT16:
    case VecSimType_FLOAT16:
        for (size_t i = 0; i < dim_; i++) {
            uint16_t h;
            std::memcpy(&h, b + 2 * i, 2);
            out[i] = type_ == VecSimType_BFLOAT16 ? bf16_widen(h) : fp16_widen(h);
        }
        break;
    case VecSimType_INT8:
        for (size_t i = 0; i < dim_; i++) out[i] = (float)(int)*(const int8_t *)(b + i);
        break;
    default:
        for (size_t i = 0; i < dim_; i++) out[i] = (float)(unsigned)*(const uint8_t *)(b + i);
        break;
    }
    if (metric_ == VecSimMetric_Cosine && (type_ == VecSimType_INT8 || type_ == VecSimType_UINT8)) {
        // the stored blob carries its norm after the elements: the builder links over unit vectors
        float nrm;
        std::memcpy(&nrm, b + dim_, 4);
        if (nrm > 0)
            for (size_t i = 0; i < dim_; i++) out[i] /= nrm;
    }
}

// snapshot of a node's link list (under the node's lock when other threads are linking)
uint32_t HnswIndex::copyLinks(uint32_t id, int level, uint32_t *dst, bool locked) {
    if (locked) lockNode(id);
    const uint32_t *links;
    uint32_t cnt;
    if (level == 0) {
        links = links0_.data() + (size_t)id * M0_;
        cnt = cnt0_[id];
    } else {
        uint32_t *cw;
        links = linksAt(id, level, &cw);
        cnt = *cw;
    }
    std::memcpy(dst, links, cnt * 4);
    if (locked) unlockNode(id);
    return cnt;
}

// ef-bounded best-first search of one layer during construction (hnswlib's searchBaseLayer shape)
void HnswIndex::searchLayer(const float *q, uint32_t ep, float ep_dist, int level, size_t ef,
                            std::vector<std::pair<float, uint32_t>> &out, BuildCtx &bc) {
    using Item = std::pair<float, uint32_t>;
    std::priority_queue<Item> top;                                         // max-heap: worst on top
    std::priority_queue<Item, std::vector<Item>, std::greater<Item>> cand;  // min-heap
    if (bc.tag.size() < n_) bc.tag.resize(n_, 0u);
    if (++bc.epoch == 0) {
        std::fill(bc.tag.begin(), bc.tag.end(), 0u);
        bc.epoch = 1;
    }
    const uint32_t ep_tag = bc.epoch;
    float lower;
    if (!deleted_[ep]) {
        top.emplace(ep_dist, ep);
        lower = ep_dist;
    } else {
        lower = std::numeric_limits<float>::max();
    }
    cand.emplace(ep_dist, ep);
    bc.tag[ep] = ep_tag;
    uint32_t links[64];
    while (!cand.empty()) {
        Item c = cand.top();
        if (c.first > lower && top.size() >= ef) break;
        cand.pop();
        const uint32_t cnt = copyLinks(c.second, level, links, bc.locked);
        for (uint32_t i = 0; i < cnt; i++) {
            const uint32_t nb = links[i];
            if (bc.tag[nb] == ep_tag) continue;
            bc.tag[nb] = ep_tag;
            const float d = buildDistance(vec(nb), q);
            if (lower > d || top.size() < ef) {
                cand.emplace(d, nb);
                if (!deleted_[nb]) top.emplace(d, nb);
                if (top.size() > ef) top.pop();
                if (!top.empty()) lower = top.top().first;
            }
        }
    }
    out.clear();
    out.reserve(top.size());
    while (!top.empty()) {
        out.push_back(top.top());
        top.pop();
    }
}

// diversity heuristic (hnsw.h:743-797): walk candidates by increasing distance, keep one only if no
// already-kept neighbour is closer to it than the query is
void HnswIndex::selectNeighbors(std::vector<std::pair<float, uint32_t>> &cands, size_t M) {
    if (cands.size() < M) return;
    std::sort(cands.begin(), cands.end(),
              [](const std::pair<float, uint32_t> &a, const std::pair<float, uint32_t> &b) { return a.first < b.first; });
    std::vector<std::pair<float, uint32_t>> kept;
    kept.reserve(M);
    for (const auto &c : cands) {
        if (kept.size() >= M) break;
        bool good = true;
        for (const auto &s : kept) {
            if (buildDistance(vec(s.second), vec(c.second)) < c.first) {
                good = false;
                break;
            }
        }
        if (good) kept.push_back(c);
    }
    cands.swap(kept);
}

void HnswIndex::connect(uint32_t id, int level, const std::vector<std::pair<float, uint32_t>> &selected, bool locked) {
    const size_t max_links = level == 0 ? M0_ : M_;
    // the new node's own list
    if (locked) lockNode(id);
    {
        uint32_t *mine;
        uint32_t dummy = 0, *mine_cnt = &dummy;
        if (level == 0) mine = links0_.data() + (size_t)id * M0_;
        else mine = linksAt(id, level, &mine_cnt);
        // Parallel build: another thread may already have appended a back link to this node (it met the node through
        // the level above, or through a neighbour written a moment ago).  Those links stay: the selected neighbours
        // first, then what was there, up to the list's capacity -- writing from index 0 would leave one-way edges
        // (hnswlib holds the element lock for the whole insert instead).
        uint32_t had[64];
        const uint32_t had_n = level == 0 ? cnt0_[id] : *mine_cnt;
        for (uint32_t i = 0; i < had_n; i++) had[i] = mine[i];
        uint32_t c = 0;
        for (const auto &s : selected) mine[c++] = s.second;
        for (uint32_t i = 0; i < had_n && c < max_links; i++) {
            bool dup = false;
            for (uint32_t j = 0; j < c; j++) dup |= (mine[j] == had[i]);
            if (!dup) mine[c++] = had[i];
        }
        if (level == 0) cnt0_[id] = (uint16_t)c;
        else *mine_cnt = c;
    }
    if (locked) unlockNode(id);
    // back links, re-selected when a neighbour's list is full
    for (const auto &s : selected) {
        const uint32_t nb = s.second;
        if (locked) lockNode(nb);
        uint32_t *nl;
        uint32_t ncnt;
        uint32_t *ncw = nullptr;
        if (level == 0) {
            nl = links0_.data() + (size_t)nb * M0_;
            ncnt = cnt0_[nb];
        } else {
            nl = linksAt(nb, level, &ncw);
            ncnt = *ncw;
        }
        bool present = false;
        for (uint32_t i = 0; i < ncnt; i++) present |= (nl[i] == id);
        if (!present) {
            if (ncnt < max_links) {
                nl[ncnt++] = id;
            } else {
                std::vector<std::pair<float, uint32_t>> cand;
                cand.reserve(ncnt + 1);
                cand.emplace_back(s.first, id);
                for (uint32_t i = 0; i < ncnt; i++) cand.emplace_back(buildDistance(vec(nl[i]), vec(nb)), nl[i]);
                selectNeighbors(cand, max_links);
                ncnt = 0;
                for (const auto &k : cand) nl[ncnt++] = k.second;
            }
            if (level == 0) cnt0_[nb] = (uint16_t)ncnt;
            else *ncw = ncnt;
        }
        if (locked) unlockNode(nb);
    }
}

int HnswIndex::drawLevel() {
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    return std::min((int)(size_t)(-std::log(uni(level_gen_)) * mult_), 255);  // hnsw.h:418-422
}

// storage for one new node (sequential: vectors may reallocate here, never during linking)
uint32_t HnswIndex::allocNode(const char *stored_blob, size_t label, int level) {
    const uint32_t id = (uint32_t)n_++;
    raw_.insert(raw_.end(), stored_blob, stored_blob + blob_bytes_);
    host_vecs_.resize(n_ * dim_);
    widen(stored_blob, host_vecs_.data() + (size_t)id * dim_);
    links0_.resize(n_ * M0_, 0u);
    cnt0_.push_back(0);
    level_.push_back((uint8_t)level);
    upper_off_.push_back(NONE);
    deleted_.push_back(0);
    labels_.push_back((uint64_t)label);
    if (multi_) label_to_ids_[label].push_back(id);
    else label_to_id_[label] = id;
    if (level > 0) {
        upper_off_[id] = (uint32_t)(upper_.size() / (M_ + 1));
        upper_.resize(upper_.size() + (size_t)level * (M_ + 1), 0u);
    }
    return id;
}

void HnswIndex::insertNode(uint32_t id, const float *v, BuildCtx &bc) {
    const int level = level_[id];
    std::unique_lock<std::mutex> entry_lock(entry_mu_, std::defer_lock);
    if (bc.locked) entry_lock.lock();
    uint32_t cur = entry_;
    const int top_level = max_level_;
    if (cur == NONE) {
        entry_ = id;
        max_level_ = level;
        return;
    }
    // a node that raises the top level keeps the entry lock for its whole insertion (hnswlib does the same)
    if (bc.locked && level <= top_level) entry_lock.unlock();
    float curd = buildDistance(vec(cur), v);
    uint32_t links[64];
    for (int l = top_level; l > level; l--) {
        bool changed = true;
        while (changed) {
            changed = false;
            const uint32_t cnt = copyLinks(cur, l, links, bc.locked);
            for (uint32_t i = 0; i < cnt; i++) {
                const float d = buildDistance(vec(links[i]), v);
                if (d < curd) {
                    curd = d;
                    cur = links[i];
                    changed = true;
                }
            }
        }
    }
    std::vector<std::pair<float, uint32_t>> W;
    for (int l = std::min(level, top_level); l >= 0; l--) {
        searchLayer(v, cur, curd, l, ef_c_, W, bc);
        if (W.empty()) continue;  // everything reachable is deleted
        auto best = std::min_element(W.begin(), W.end());
        cur = best->second;
        curd = best->first;
        std::vector<std::pair<float, uint32_t>> sel;
        sel.reserve(W.size());
        for (const auto &w : W)
            if (w.second != id) sel.push_back(w);  // (a concurrent insert may already have linked to us)
        selectNeighbors(sel, M_);
        if (sel.size() > M_) sel.resize(M_);
        connect(id, l, sel, bc.locked);
    }
    if (level > top_level) {
        entry_ = id;
        max_level_ = level;
    }
}

int HnswIndex::addVector(const void *blob, size_t label) {
    int is_new = 1;
    auto it = multi_ ? label_to_id_.end() : label_to_id_.find(label);   // multi: "we always add the vector, no overrides" (hnsw_multi.h:213-218)
    if (it != label_to_id_.end()) {  // overwrite = mark the old vector deleted + insert (hnsw_single.h)
        deleted_[it->second] = 1;
        n_deleted_++;
        label_to_id_.erase(it);
        is_new = 0;
    }
    std::vector<char> pv = preprocess(blob);
    const uint32_t id = allocNode(pv.data(), label, drawLevel());
    main_ctx_.locked = false;
    if (ref_add_) insertNodeRef(id);
    else insertNode(id, vec(id), main_ctx_);
    graph_dirty_ = true;
    if (!is_new) maybeCompact();   // (an overwrite leaves a dead node behind)
    return is_new;
}

// Bulk ingest: storage and levels are laid out sequentially (deterministic ids and levels), then the
// linking runs on VECSIM_HNSW_BUILD_THREADS host threads (default: all cores, at most 64) with per-node
// link-list locks, the way the reference's parallel insert path does (hnsw.h:436-445, bindings.cpp:383-426).
long HnswIndex::storedVectors(size_t label, void *out, size_t cap_bytes) {
    if (multi_) {
        auto f = label_to_ids_.find(label);
        if (f == label_to_ids_.end()) return 0;
        if (cap_bytes < f->second.size() * blob_bytes_) return -1;
        for (size_t i = 0; i < f->second.size(); i++)
            std::memcpy((char *)out + i * blob_bytes_, raw_.data() + (size_t)f->second[i] * blob_bytes_, blob_bytes_);
        return (long)f->second.size();
    }
    auto f = label_to_id_.find(label);
    if (f == label_to_id_.end()) return 0;
    if (cap_bytes < blob_bytes_) return -1;
    std::memcpy(out, raw_.data() + (size_t)f->second * blob_bytes_, blob_bytes_);
    return 1;
}

long HnswIndex::addBulk(const void *blobs, const size_t *labels, size_t n) {
    for (size_t i = 0; i < n && !multi_; i++)
        if (label_to_id_.count(labels[i])) return -1;
    if (!multi_) {   // a label may appear once per batch (the sequential path would overwrite; the parallel one would keep both alive)
        std::unordered_map<size_t, char> seen;
        seen.reserve(n * 2);
        for (size_t i = 0; i < n; i++)
            if (!seen.emplace(labels[i], 1).second) return -1;
    }
    size_t threads = std::thread::hardware_concurrency();
    if (const char *e = std::getenv("VECSIM_HNSW_BUILD_THREADS")) threads = (size_t)std::max(1, std::atoi(e));
    // (more threads link faster but see less of each other's nodes: at 256 threads recall on a 20 K-row graph fell
    // from 0.93 to 0.88 and the 1 M-row build was slower, 636 s vs 379 s; software prefetch of the next row: 108 s vs 91 s
    // at 300 K rows -- measured, not kept)
    // default: at most 64 linking threads; VECSIM_HNSW_BUILD_THREADS may ask for up to 256
    threads = std::max<size_t>(1, std::min<size_t>(threads, std::getenv("VECSIM_HNSW_BUILD_THREADS") ? 256 : 64));
    if (n < 2048 || threads == 1 || ref_bulk_) {
        host_vecs_.reserve(host_vecs_.size() + n * dim_);
        for (size_t i = 0; i < n; i++) addVector((const char *)blobs + i * dim_ * elem_bytes_, labels[i]);
        return (long)n;
    }
    const uint32_t first = (uint32_t)n_;
    host_vecs_.reserve(host_vecs_.size() + n * dim_);
    raw_.reserve(raw_.size() + n * blob_bytes_);
    for (size_t i = 0; i < n; i++) {
        std::vector<char> pv = preprocess((const char *)blobs + i * dim_ * elem_bytes_);
        allocNode(pv.data(), labels[i], drawLevel());
    }
    node_lock_.reset(new std::atomic_flag[n_]);
    for (size_t i = 0; i < n_; i++) node_lock_[i].clear();
    node_lock_n_ = n_;
    // the first few nodes go in sequentially so every thread starts from a connected graph
    size_t seq = std::min<size_t>(n, 256);
    main_ctx_.locked = false;
    for (size_t i = 0; i < seq; i++) insertNode(first + (uint32_t)i, vec(first + (uint32_t)i), main_ctx_);
    std::atomic<size_t> next{seq};
    std::vector<std::thread> pool;
    for (size_t t = 0; t < threads; t++) {
        pool.emplace_back([&]() {
            BuildCtx bc;
            bc.locked = true;
            bc.tag.assign(n_, 0u);
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= n) break;
                insertNode(first + (uint32_t)i, vec(first + (uint32_t)i), bc);
            }
        });
    }
    for (auto &th : pool) th.join();
    node_lock_.reset();
    graph_dirty_ = true;
    return (long)n;
}

int HnswIndex::deleteVector(size_t label) {
    if (multi_) {   // hnsw_multi.h:197-211: every node of the label, the number removed is returned
        auto f = label_to_ids_.find(label);
        if (f == label_to_ids_.end()) return 0;
        const int removed = (int)f->second.size();
        for (uint32_t id : f->second) {
            deleted_[id] = 1;
            n_deleted_++;
        }
        label_to_ids_.erase(f);
        graph_dirty_ = true;
        maybeCompact();
        return removed;
    }
    auto it = label_to_id_.find(label);
    if (it == label_to_id_.end()) return 0;
    deleted_[it->second] = 1;  // marked: never returned, still traversable (hnsw.h:572-573) until the next compaction
    n_deleted_++;
    label_to_id_.erase(it);
    graph_dirty_ = true;
    maybeCompact();
    return 1;
}

void HnswIndex::maybeCompact() {
    if (n_deleted_ == 0 || walkers_.load() != 0) return;
    if (n_deleted_ < n_ && (n_deleted_ < 32 || n_deleted_ * 16 < n_)) return;
    if (compactDeleted()) std::fprintf(stderr, "vecsim_amd: HNSW compaction failed: %s\n", vsgpu_last_error());
}

// see hnsw_index.h.  Writers are the caller's to serialise (vec_sim.h); readers on lanes are kept out here.
int HnswIndex::compactDeleted() {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);
    std::vector<std::unique_lock<std::mutex>> lane_locks;
    for (auto &l : lanes_) lane_locks.emplace_back(l->mu);
    if (walkers_.load() != 0) return 0;   // (an iterator created between maybeCompact's look and the locks: it holds node ids)
    const size_t n = n_, n_new = n - n_deleted_;
    // 1. every live node that points at a dead one: a new list, level by level
    std::vector<uint32_t> mark(n, 0xFFFFFFFFu);
    uint32_t stamp = 0;
    std::vector<std::pair<float, uint32_t>> cands;
    for (size_t p = 0; p < n; p++) {
        if (deleted_[p]) continue;
        for (int level = 0; level <= (int)level_[p]; level++) {
            uint32_t *cntw = nullptr;
            uint32_t *links = level == 0 ? links0_.data() + p * M0_ : linksAt((uint32_t)p, level, &cntw);
            const uint32_t cnt = level == 0 ? (uint32_t)cnt0_[p] : *cntw;
            bool touched = false;
            for (uint32_t i = 0; i < cnt; i++) touched |= deleted_[links[i]] != 0;
            if (!touched) continue;
            stamp++;
            cands.clear();
            auto add = [&](uint32_t w) {
                if (w == p || deleted_[w] || mark[w] == stamp) return;
                mark[w] = stamp;
                cands.emplace_back(0.0f, w);
            };
            for (uint32_t i = 0; i < cnt; i++)
                if (!deleted_[links[i]]) add(links[i]);
            for (uint32_t i = 0; i < cnt; i++) {
                const uint32_t d = links[i];
                if (!deleted_[d] || (int)level_[d] < level) continue;
                uint32_t *dcw = nullptr;
                const uint32_t *dl = level == 0 ? links0_.data() + (size_t)d * M0_ : linksAt(d, level, &dcw);
                const uint32_t dn = level == 0 ? (uint32_t)cnt0_[d] : *dcw;
                for (uint32_t j = 0; j < dn; j++) add(dl[j]);
            }
            const size_t max_links = level == 0 ? M0_ : M_;
            if (cands.size() > max_links) {
                for (auto &c : cands) c.first = buildDistance(vec(c.second), vec((uint32_t)p));
                selectNeighbors(cands, max_links);
            }
            const uint32_t kept = (uint32_t)std::min(cands.size(), max_links);
            for (uint32_t i = 0; i < kept; i++) links[i] = cands[i].second;
            if (level == 0) cnt0_[p] = (uint16_t)kept;
            else *cntw = kept;
        }
    }
    // 2. the entry point (replaceEntryPoint, hnsw.h:1455-1500: a live node of the highest level left)
    if (entry_ != 0xFFFFFFFFu && deleted_[entry_]) {
        entry_ = 0xFFFFFFFFu;
        max_level_ = -1;
        for (size_t p = 0; p < n; p++)
            if (!deleted_[p] && (int)level_[p] > max_level_) {
                max_level_ = (int)level_[p];
                entry_ = (uint32_t)p;
            }
    }
    // 3. the last live nodes move into the holes
    std::vector<uint32_t> new_id(n - n_new, 0xFFFFFFFFu);   // for ids >= n_new
    const bool rows_on_device = uploaded_rows_ == n;
    // (the device's row moves are carried out AFTER the host state is whole again -- round-5 advisor: a move that failed half way
    // used to leave labels, links and rows rewritten for some holes only)
    std::vector<std::pair<uint32_t, uint32_t>> device_moves;
    {
        size_t t = n;
        for (size_t h = 0; h < n_new; h++) {
            if (!deleted_[h]) continue;
            do t--; while (deleted_[t]);   // (t >= n_new: there are exactly as many live nodes up there as holes down here)
            new_id[t - n_new] = (uint32_t)h;
            std::memcpy(raw_.data() + h * blob_bytes_, raw_.data() + t * blob_bytes_, blob_bytes_);
            std::memcpy(host_vecs_.data() + h * dim_, host_vecs_.data() + t * dim_, dim_ * sizeof(float));
            labels_[h] = labels_[t];
            level_[h] = level_[t];
            cnt0_[h] = cnt0_[t];
            std::memcpy(links0_.data() + h * M0_, links0_.data() + t * M0_, M0_ * sizeof(uint32_t));
            upper_off_[h] = upper_off_[t];
            deleted_[h] = 0;
            if (multi_) {
                for (uint32_t &v : label_to_ids_.at((size_t)labels_[h]))
                    if (v == (uint32_t)t) v = (uint32_t)h;
            } else label_to_id_[(size_t)labels_[h]] = (uint32_t)h;
            if (rows_on_device) device_moves.emplace_back((uint32_t)h, (uint32_t)t);
        }
    }
    auto remap = [&](uint32_t v) { return v >= n_new ? new_id[v - n_new] : v; };
    if (entry_ != 0xFFFFFFFFu) entry_ = remap(entry_);
    // 4. links renumbered, the upper-level blocks packed, the arrays shrunk
    std::vector<uint32_t> upper_new;
    upper_new.reserve(upper_.size());
    for (size_t p = 0; p < n_new; p++) {
        uint32_t *l0 = links0_.data() + p * M0_;
        for (uint32_t i = 0; i < cnt0_[p]; i++) l0[i] = remap(l0[i]);
        if (level_[p] == 0) {
            upper_off_[p] = 0xFFFFFFFFu;
            continue;
        }
        const uint32_t *blk = upper_.data() + (size_t)upper_off_[p] * (M_ + 1);
        const uint32_t off_new = (uint32_t)(upper_new.size() / (M_ + 1));
        for (int level = 1; level <= (int)level_[p]; level++, blk += M_ + 1) {
            upper_new.push_back(blk[0]);
            for (size_t i = 0; i < M_; i++) upper_new.push_back(i < blk[0] ? remap(blk[1 + i]) : 0u);
        }
        upper_off_[p] = off_new;
    }
    upper_.swap(upper_new);
    raw_.resize(n_new * blob_bytes_);
    host_vecs_.resize(n_new * dim_);
    labels_.resize(n_new);
    level_.resize(n_new);
    cnt0_.resize(n_new);
    links0_.resize(n_new * M0_);
    upper_off_.resize(n_new);
    deleted_.resize(n_new);
    n_ = n_new;
    n_deleted_ = 0;
    graph_dirty_ = true;
    // the host graph is consistent from here on; now the device table.  Any failure: drop the device rows altogether -- the next
    // query re-appends every row from raw_ (syncDevice) -- and report
    bool dev_ok = rows_on_device;
    for (size_t i = 0; dev_ok && i < device_moves.size(); i++) dev_ok = vsgpu_table_move(table_, device_moves[i].first, device_moves[i].second) == 0;
    if (dev_ok) dev_ok = vsgpu_table_truncate(table_, n_new) == 0;
    if (dev_ok) {
        uploaded_rows_ = n_new;
        return 0;
    }
    const int rc = vsgpu_table_truncate(table_, 0);   // (also the path of rows that were still pending)
    uploaded_rows_ = 0;
    return (rows_on_device || rc) ? -1 : 0;
}

int HnswIndex::syncDevice() {
    if (uploaded_rows_ < n_) {
        int rc = vsgpu_table_append(table_, raw_.data() + uploaded_rows_ * blob_bytes_, n_ - uploaded_rows_);
        if (rc) return rc;
        uploaded_rows_ = n_;
    }
    if (graph_dirty_) {
        int rc = vsgpu_graph_upload(graph_, n_, links0_.data(), cnt0_.data(), upper_off_.data(), upper_.data(), upper_.size(),
                                    deleted_.data(), labels_.data(), entry_, max_level_);
        if (rc) return rc;
        graph_dirty_ = false;
    }
    return 0;
}

int HnswIndex::neighborLabels(size_t label, std::vector<std::vector<size_t>> &out) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);
    out.clear();
    if (multi_) return -2;
    auto f = label_to_id_.find(label);
    if (f == label_to_id_.end()) return -1;
    const uint32_t id = f->second;
    for (int level = 0; level <= (int)level_[id]; level++) {
        uint32_t *cnt = nullptr;
        const uint32_t *links = level == 0 ? links0_.data() + (size_t)id * M0_ : linksAt(id, level, &cnt);
        const uint32_t n = level == 0 ? (uint32_t)cnt0_[id] : *cnt;
        std::vector<size_t> labs(n);
        for (uint32_t i = 0; i < n; i++) labs[i] = (size_t)labels_[links[i]];
        out.push_back(std::move(labs));
    }
    return 0;
}

HnswIndex::Export HnswIndex::exportGraph() {
    Export e{};
    e.n = (uint32_t)n_;
    e.M = (uint32_t)M_;
    e.M0 = (uint32_t)M0_;
    e.entry = entry_;
    e.max_level = max_level_;
    e.links0 = links0_.data();
    e.cnt0 = cnt0_.data();
    e.upper_off = upper_off_.data();
    e.upper = upper_.data();
    e.upper_words = upper_.size();
    e.deleted = deleted_.data();
    e.labels = labels_.data();
    return e;
}

int HnswIndex::topKQueryBatch(const void *queries, size_t nq, size_t stride, size_t k, VecSimQueryParams *qp,
                              VecSimQueryReply_Order order, VecSimQueryReply **out) {
    // readers may call concurrently (vec_sim.h contract): a reader that finds the index's own context busy searches on a reader
    // lane -- unless the device snapshot is stale (rows or links not uploaded yet), which only the lock holder may refresh
    std::unique_lock<std::recursive_mutex> gpu_lock(gpu_mu_, std::defer_lock);
    Lane *lane = nullptr;
    if (!gpu_lock.try_lock()) {
        if (!graph_dirty_ && uploaded_rows_ == n_) lane = tryLane();
        if (!lane) gpu_lock.lock();
    }
    struct LaneRelease {
        Lane *l;
        ~LaneRelease() {
            if (l) l->mu.unlock();
        }
    } lane_release{lane};
    vsgpu_graph *gr = graph_;
    vsgpu_table *tbl = table_;
    if (lane) {
        vsgpu_table_view_sync(lane->view);
        gr = lane->graph;
        tbl = lane->view;
    }
    void *tctx = qp ? qp->timeoutCtx : nullptr;
    if (!lane) last_mode_ = STANDARD_KNN;
    if (nq == 0) return 0;
    std::vector<VecSimQueryReply *> reps(nq);
    for (auto &r : reps) r = new VecSimQueryReply();
    auto finish = [&]() {
        for (size_t q = 0; q < nq; q++) out[q] = reps[q];
        return 0;
    };
    if (k == 0 || n_ == 0) return finish();
    if (timed_out(tctx)) {
        for (auto *r : reps) r->code = VecSim_QueryReply_TimedOut;
        return finish();
    }
    size_t ef = ef_;
    if (qp && qp->hnswRuntimeParams.efRuntime != 0) ef = qp->hnswRuntimeParams.efRuntime;
    ef = std::max(ef, k);  // hnsw.h:2073
    // top_candidates never holds more than the live nodes, so an ef (and k) beyond that behaves exactly like ef = live
    // (the admission test `size < ef` then never fails): clamp before sizing the kernel's LDS heaps
    const size_t live = n_ - n_deleted_;
    if (live == 0) return finish();
    // (multi-value: the heap holds labels, hnsw_multi.h:108-112 -- no more than there are)
    const size_t k_eff = std::min(k, multi_ ? label_to_ids_.size() : live);
    ef = std::min(ef, live);
    // only Cosine needs a private (normalised) copy of the queries
    std::vector<char> qbuf;
    const void *qsrc = queries;
    size_t qstride = stride;
    if (metric_ == VecSimMetric_Cosine) {
        qbuf.resize(nq * blob_bytes_);
        for (size_t q = 0; q < nq; q++) {
            std::memcpy(qbuf.data() + q * blob_bytes_, (const char *)queries + q * stride, dim_ * elem_bytes_);
            normalize_blob(qbuf.data() + q * blob_bytes_, dim_, type_);
        }
        qsrc = qbuf.data();
        qstride = blob_bytes_;
    }
    std::vector<uint64_t> labs(nq * k_eff);
    std::vector<double> sc(nq * k_eff);
    std::vector<uint32_t> cnt(nq);
    int rc = lane ? 0 : syncDevice();
    uint64_t evals = 0;
    if (!rc) rc = vsgpu_graph_search(gr, qsrc, nq, qstride, k_eff, ef, labs.data(), sc.data(), cnt.data(), &evals);
    last_dist_evals_ = evals;
    if (rc == VSGPU_ERR_UNSUPPORTED) {
        // ef beyond what the per-query LDS heaps hold (about 1.3 K at dim 768): the graph walk cannot be replayed, so
        // the batch is answered by the exact GPU scan of the table -- the k best live vectors, a reply at least as good
        // as any graph walk's.  Deleted nodes still sit in the table: ask for that many more rows and drop them.
        const size_t kk = std::min(n_, k_eff + n_deleted_);
        const size_t cap = std::max<size_t>(2 * kk, kk + 64);
        std::vector<uint32_t> ids(cap), c1(1);
        std::vector<double> s1(cap);
        using Item = std::pair<double, size_t>;
        for (size_t q = 0; q < nq; q++) {
            const char *qp1 = (const char *)qsrc + q * qstride;
            std::vector<double> all;
            if (multi_) {   // the k best LABELS need every row's score (a label's best row may rank anywhere): no top-k pass first
                all.resize(n_);
                rc = vsgpu_scores(tbl, qp1, 0, n_, all.data());
            } else {
                rc = vsgpu_topk(tbl, qp1, 1, qstride, kk, cap, ids.data(), s1.data(), c1.data());
                if (!rc && c1[0] == VSGPU_COUNT_OVERFLOW) {  // massive ties at the kk-th score: every row's score
                    all.resize(n_);
                    rc = vsgpu_scores(tbl, qp1, 0, n_, all.data());
                }
            }
            if (rc) break;
            RefMaxHeap<Item> heap;
            std::unordered_map<size_t, double> best_of;   // multi: per-label minimum first, then the plain heap over labels
            if (multi_) {
                for (size_t i = 0; i < n_; i++) {
                    if (deleted_[i]) continue;
                    auto f = best_of.find((size_t)labels_[i]);
                    if (f == best_of.end()) best_of.emplace((size_t)labels_[i], all[i]);
                    else if (all[i] < f->second) f->second = all[i];
                }
                for (auto &e : best_of) {
                    heap.emplace(e.second, e.first);
                    if (heap.size() > k_eff) heap.pop();
                }
                cnt[q] = (uint32_t)heap.size();
                for (size_t i = heap.size(); i-- > 0;) {
                    labs[q * k_eff + i] = heap.top().second;
                    sc[q * k_eff + i] = heap.top().first;
                    heap.pop();
                }
                continue;
            }
            double upper = std::numeric_limits<double>::lowest();
            const size_t m = all.empty() ? c1[0] : n_;
            for (size_t i = 0; i < m; i++) {
                const uint32_t id = all.empty() ? ids[i] : (uint32_t)i;
                const double sco = all.empty() ? s1[i] : all[i];
                if (deleted_[id]) continue;
                if (sco < upper || heap.size() < k_eff) {
                    heap.emplace(sco, (size_t)labels_[id]);
                    if (heap.size() > k_eff) heap.pop();
                    upper = heap.top().first;
                }
            }
            cnt[q] = (uint32_t)heap.size();
            for (size_t i = heap.size(); i-- > 0;) {
                labs[q * k_eff + i] = heap.top().second;
                sc[q * k_eff + i] = heap.top().first;
                heap.pop();
            }
        }
        last_dist_evals_ = (uint64_t)nq * n_;
    }
    if (rc) {
        std::fprintf(stderr, "vecsim_amd: GPU HNSW search failed: %s\n", vsgpu_last_error());
        for (auto *r : reps) delete r;
        return rc;
    }
    if (timed_out(tctx)) {
        for (auto *r : reps) r->code = VecSim_QueryReply_TimedOut;
        return finish();
    }
    for (size_t q = 0; q < nq; q++) {
        reps[q]->results.resize(cnt[q]);
        for (uint32_t i = 0; i < cnt[q]; i++) {
            reps[q]->results[i].id = (size_t)labs[q * k_eff + i];
            reps[q]->results[i].score = sc[q * k_eff + i];
        }
        if (order == BY_ID) sort_reply(reps[q], BY_ID);
    }
    return finish();
}

VecSimQueryReply *HnswIndex::topKQuery(const void *query, size_t k, VecSimQueryParams *qp) {
    VecSimQueryReply *rep = nullptr;
    if (topKQueryBatch(query, 1, 0, k, qp, BY_SCORE, &rep)) {
        rep = new VecSimQueryReply();
        rep->code = VecSim_QueryReply_TimedOut;
    }
    return rep;
}

// rangeQuery (hnsw.h:2153-2187): greedy descent to the bottom-layer entry point, then the epsilon-bounded
// range search, both on the GPU (k_hnsw_search in range mode); the reply is then ordered like every range reply.
VecSimQueryReply *HnswIndex::rangeQuery(const void *query, double radius, VecSimQueryParams *qp, VecSimQueryReply_Order order) {
    std::unique_lock<std::recursive_mutex> gpu_lock(gpu_mu_, std::defer_lock);   // (reader lanes as in topKQueryBatch)
    Lane *lane = nullptr;
    if (!gpu_lock.try_lock()) {
        if (!graph_dirty_ && uploaded_rows_ == n_) lane = tryLane();
        if (!lane) gpu_lock.lock();
    }
    struct LaneRelease {
        Lane *l;
        ~LaneRelease() {
            if (l) l->mu.unlock();
        }
    } lane_release{lane};
    vsgpu_graph *gr = graph_;
    vsgpu_table *tbl = table_;
    if (lane) {
        vsgpu_table_view_sync(lane->view);
        gr = lane->graph;
        tbl = lane->view;
    }
    auto *rep = new VecSimQueryReply();
    if (!lane) last_mode_ = RANGE_QUERY;
    if (n_ == 0) return rep;
    void *tctx = qp ? qp->timeoutCtx : nullptr;
    if (timed_out(tctx)) {
        rep->code = VecSim_QueryReply_TimedOut;
        return rep;
    }
    double eps = epsilon_;
    if (qp && qp->hnswRuntimeParams.epsilon != 0.0) eps = qp->hnswRuntimeParams.epsilon;
    std::vector<char> qbuf = preprocess(query);
    if (!lane && syncDevice()) {
        std::fprintf(stderr, "vecsim_amd: GPU HNSW range query failed: %s\n", vsgpu_last_error());
        return rep;
    }
    size_t cap = 1024;
    std::vector<uint64_t> labs;
    std::vector<double> sc;
    uint32_t cnt = 0;
    for (;;) {
        labs.resize(cap);
        sc.resize(cap);
        uint64_t evals = 0;
        const int rrc = vsgpu_graph_range(gr, qbuf.data(), 1, blob_bytes_, radius, eps, cap, labs.data(), sc.data(), &cnt, &evals);
        last_dist_evals_ = evals;
        if (rrc) {
            std::fprintf(stderr, "vecsim_amd: GPU HNSW range query failed: %s\n", vsgpu_last_error());
            return rep;
        }
        if ((cnt & 0x7FFFFFFFu) <= cap) break;
        cap = std::max<size_t>(cnt & 0x7FFFFFFFu, 2 * cap);
    }
    if (timed_out(tctx)) {
        rep->code = VecSim_QueryReply_TimedOut;
        return rep;
    }
    if (cnt & 0x80000000u) {
        // More live candidates than the kernel's LDS window holds (a radius covering a large part of the index):
        // the graph walk can no longer be replayed exactly, so the query is answered by the exact GPU scan of
        // the table instead -- every vector within the radius, a superset of what the graph walk would reach.
        std::vector<uint32_t> ids(n_);
        std::vector<double> s2(n_);
        uint32_t c2 = 0;
        if (vsgpu_range(tbl, qbuf.data(), radius, n_, ids.data(), s2.data(), &c2) || c2 == VSGPU_COUNT_OVERFLOW) return rep;
        for (uint32_t i = 0; i < c2; i++)
            if (!deleted_[ids[i]]) rep->results.push_back(VecSimQueryResult{(size_t)labels_[ids[i]], s2[i]});
    } else {
        rep->results.resize(cnt);
        for (uint32_t i = 0; i < cnt; i++) {
            rep->results[i].id = (size_t)labs[i];
            rep->results[i].score = sc[i];
        }
    }
    if (multi_) {   // unique_results_container (vecsim_results_container.h:33-62): one result per label, its lowest score
        std::unordered_map<size_t, size_t> at;
        size_t w = 0;
        for (size_t i = 0; i < rep->results.size(); i++) {
            auto f = at.find(rep->results[i].id);
            if (f == at.end()) {
                at.emplace(rep->results[i].id, w);
                rep->results[w++] = rep->results[i];
            } else if (rep->results[i].score < rep->results[f->second].score) {
                rep->results[f->second].score = rep->results[i].score;
            }
        }
        rep->results.resize(w);
    }
    sort_reply(rep, order);
    return rep;
}

// Batch iterator.  The reference walks the graph incrementally (hnsw_batch_iterator.h:96-230) and hands out approximate
// next-best batches: hnsw_iter.cpp does the same walk with the GPU's distances (rounds 1-3 answered with the EXACT next-best set
// from one score pass over all rows, through the Flat index's iterator machinery; VECSIM_HNSW_ITER_EXACT=1 still does).
VecSimBatchIterator *HnswIndex::newBatchIterator(const void *query, VecSimQueryParams *qp) {
    auto *it = new VecSimBatchIterator();
    it->index = this;
    it->query = preprocess(query);
    it->timeout_ctx = qp ? qp->timeoutCtx : nullptr;
    it->label_count = indexLabelCount();
    if (!std::getenv("VECSIM_HNSW_ITER_EXACT")) it->walker.reset(newWalker(it->query, qp));
    return it;
}
int HnswIndex::iteratorScores(const void *processed_query, std::vector<std::pair<double, size_t>> &out) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);  // readers may call concurrently (vec_sim.h contract)
    out.clear();
    if (n_ == 0) return 0;
    if (syncDevice()) return -1;
    std::vector<double> s(n_);
    if (vsgpu_scores(table_, processed_query, 0, n_, s.data())) return -1;
    if (multi_) {   // one entry per label, its lowest score (as the Flat multi-value iterator: bfm_batch_iterator.h:24-53)
        std::unordered_map<size_t, double> best;
        for (size_t i = 0; i < n_; i++) {
            if (deleted_[i]) continue;
            auto f = best.find((size_t)labels_[i]);
            if (f == best.end()) best.emplace((size_t)labels_[i], s[i]);
            else if (f->second > s[i]) f->second = s[i];
        }
        out.reserve(best.size());
        for (auto &p : best) out.emplace_back(p.second, p.first);
        return 0;
    }
    out.reserve(n_ - n_deleted_);
    for (size_t i = 0; i < n_; i++)
        if (!deleted_[i]) out.emplace_back(s[i], (size_t)labels_[i]);
    return 0;
}

// device-resident iterator state: the same score buffer as the Flat index, deleted nodes retired up front
vsgpu_scorebuf *HnswIndex::iteratorDeviceBegin(const void *processed_query) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);
    if (multi_) return nullptr;   // (a label's best vector has to win: the host array path de-duplicates by label, as for Flat)
    if (n_ == 0 || syncDevice()) return nullptr;
    vsgpu_scorebuf *b = vsgpu_scorebuf_create(table_, processed_query);
    if (b && n_deleted_) {
        std::vector<uint32_t> dead;
        for (size_t i = 0; i < n_; i++)
            if (deleted_[i]) dead.push_back((uint32_t)i);
        if (vsgpu_scorebuf_retire(b, dead.data(), dead.size())) {
            vsgpu_scorebuf_destroy(b);
            return nullptr;
        }
    }
    return b;
}
int HnswIndex::iteratorDeviceNext(vsgpu_scorebuf *b, size_t k, size_t cap, uint32_t *ids, double *scores, uint32_t *count) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);
    return vsgpu_scorebuf_next(b, k, cap, ids, scores, count);
}
int HnswIndex::iteratorDeviceRetire(vsgpu_scorebuf *b, const uint32_t *rows, size_t m) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);
    return vsgpu_scorebuf_retire(b, rows, m);
}
int HnswIndex::iteratorDeviceRead(vsgpu_scorebuf *b, double *all) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);
    return vsgpu_scorebuf_read(b, all);
}
void HnswIndex::iteratorDeviceEnd(vsgpu_scorebuf *b) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);
    vsgpu_scorebuf_destroy(b);
}

double HnswIndex::getDistanceFrom(size_t label, const void *blob) {
    std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mu_);  // readers may call concurrently (vec_sim.h contract)
    if (multi_) {   // hnsw_multi.h:138-162: the minimum over the label's vectors (std::fmin from INVALID_SCORE = NaN)
        auto f = label_to_ids_.find(label);
        if (f == label_to_ids_.end() || syncDevice()) return std::numeric_limits<double>::quiet_NaN();
        std::vector<double> s(f->second.size());
        if (vsgpu_scores_of(table_, blob, f->second.data(), f->second.size(), s.data())) return std::numeric_limits<double>::quiet_NaN();
        double best = std::numeric_limits<double>::quiet_NaN();
        for (double v : s) best = std::fmin(best, v);
        return best;
    }
    auto it = label_to_id_.find(label);
    if (it == label_to_id_.end() || syncDevice()) return std::numeric_limits<double>::quiet_NaN();
    uint32_t id = it->second;
    double s = std::numeric_limits<double>::quiet_NaN();
    if (vsgpu_scores_of(table_, blob, &id, 1, &s)) return std::numeric_limits<double>::quiet_NaN();
    return s;
}

// The reference's 20-leaf decision tree (hnsw.h:2275-2408, fitted by scripts/HNSW_batches_clf.py), kept as data:
// each node tests one feature against a threshold and names the next node for "yes" / "no"; negative entries are
// the leaves (-1: ad-hoc brute force, -2: batches).
bool HnswIndex::preferAdHocSearch(size_t subsetSize, size_t k, bool initial_check) {
    enum Feature { N, R, K, D, MM };
    enum { ADHOC = -1, BATCHES = -2 };
    struct Node { Feature f; double thr; bool strict; int yes, no; };
    static const Node tree[] = {
        /* 0*/ {N, 30000, false, 1, 7},
        /* 1*/ {N, 5500, false, ADHOC, 2},
        /* 2*/ {R, 0.17, false, ADHOC, 3},
        /* 3*/ {K, 12, false, 4, ADHOC},
        /* 4*/ {D, 55, false, BATCHES, 5},
        /* 5*/ {MM, 10, false, BATCHES, ADHOC},
        /* 6*/ {N, 0, false, ADHOC, ADHOC},  // unused slot
        /* 7*/ {R, 0.07, true, 8, 11},
        /* 8*/ {N, 750000, false, ADHOC, 9},
        /* 9*/ {K, 7, false, BATCHES, 10},
        /*10*/ {R, 0.03, false, ADHOC, BATCHES},
        /*11*/ {D, 75, false, BATCHES, 12},
        /*12*/ {K, 12, false, 13, 16},
        /*13*/ {R, 0.21, false, 14, BATCHES},
        /*14*/ {MM, 57, false, 15, ADHOC},
        /*15*/ {N, 75000, false, ADHOC, BATCHES},
        /*16*/ {MM, 10, false, 17, 18},
        /*17*/ {R, 0.17, false, ADHOC, BATCHES},
        /*18*/ {N, 300000, false, ADHOC, 19},
        /*19*/ {R, 0.17, false, ADHOC, BATCHES},
    };
    const size_t index_size = indexSize();
    subsetSize = std::min(subsetSize, index_size);
    const float r = index_size == 0 ? 0.0f : (float)subsetSize / (float)indexLabelCount();
    int at = 0;
    while (at >= 0) {
        const Node &nd = tree[at];
        bool yes;
        switch (nd.f) {
        case N: yes = (double)index_size <= nd.thr; break;
        case K: yes = (double)k <= nd.thr; break;
        case D: yes = (double)dim_ <= nd.thr; break;
        case MM: yes = (double)M_ <= nd.thr; break;
        default: yes = nd.strict ? ((double)r < nd.thr) : ((double)r <= nd.thr); break;  // float r against a double literal;
        }                                                                                // `r < 0.07` is the one strict test
        at = yes ? nd.yes : nd.no;
    }
    const bool adhoc = at == ADHOC;
    last_mode_ = adhoc ? (initial_check ? HYBRID_ADHOC_BF : HYBRID_BATCHES_TO_ADHOC_BF) : HYBRID_BATCHES;
    return adhoc;
}

VecSimIndexBasicInfo HnswIndex::basicInfo() const {
    VecSimIndexBasicInfo b{};
    b.algo = VecSimAlgo_HNSWLIB;
    b.metric = metric_;
    b.type = type_;
    b.blockSize = block_size_;
    b.dim = dim_;
    return b;
}
VecSimIndexStatsInfo HnswIndex::statsInfo() const {
    VecSimIndexStatsInfo s{};
    s.memory = host_vecs_.capacity() * 4 + raw_.capacity() + links0_.capacity() * 4 + upper_.capacity() * 4 + (table_ ? vsgpu_table_bytes(table_) : 0);
    s.numberOfMarkedDeleted = n_deleted_;
    return s;
}
VecSimIndexDebugInfo HnswIndex::debugInfo() const {
    VecSimIndexDebugInfo d{};
    d.commonInfo.basicInfo = basicInfo();
    d.commonInfo.indexSize = indexSize();
    d.commonInfo.indexLabelCount = indexLabelCount();
    d.commonInfo.memory = statsInfo().memory;
    d.commonInfo.lastMode = last_mode_;
    d.hnswInfo.M = M_;
    d.hnswInfo.efConstruction = ef_c_;
    d.hnswInfo.efRuntime = ef_;
    d.hnswInfo.epsilon = epsilon_;
    d.hnswInfo.max_level = max_level_ < 0 ? HNSW_INVALID_LEVEL : (size_t)max_level_;
    d.hnswInfo.entrypoint = entry_ == NONE ? INVALID_LABEL : (size_t)labels_[entry_];
    d.hnswInfo.numberOfMarkedDeletedNodes = n_deleted_;
    return d;
}

}  // namespace vsa
