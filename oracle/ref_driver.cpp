/*
 * ref_driver.cpp -- C entry points over the REFERENCE's own translation units, compiled where they lie.
 *
 * TEST INFRASTRUCTURE ONLY (see vso.h).  Built by oracle/build_ref.sh into oracle/_ref/libvsref.so together with
 * /root/reference/src/VecSim/spaces/L2/L2.cpp, spaces/IP/IP.cpp and memory/vecsim_malloc.cpp -- the reference
 * translation units that compile with plain g++ and no third-party header (they include nothing from the
 * un-vendored cpu_features dependency).  No reference source is copied: this file only #includes reference
 * headers by path and forwards to the reference's functions, so every number that leaves this library was
 * computed by the reference's own code:
 *   scalar distance kernels            spaces/L2/L2.cpp:76-201, spaces/IP/IP.cpp:34-286
 *   normalisation                      spaces/normalize/normalize_naive.h:24-88, compute_norm.h:18-31
 *   bf16 / fp16 conversions            types/bfloat16.h:23-39, types/float16.h:33-117
 *   the Flat top-k container           utils/vecsim_stl.h:63-83 (max_priority_queue)
 *   the multi-value top-k container    utils/updatable_heap.h:20-113
 *   the min-heap of the HNSW iterator  utils/vecsim_stl.h:85-89 (min_priority_queue; vsref_heap_script)
 * The two heap loops below are the caller side of those containers: the statements of brute_force.h:257-288
 * (insert when `score < upperBound || size < k`, pop when over k, upperBound = top) -- BruteForceIndex itself
 * cannot be compiled here (vec_sim_index.h reaches spaces/space_includes.h:13, the cpu_features headers).
 *
 * What is NOT reachable this way: every SIMD tier (all of them include space_includes.h).  Those stay pinned by
 * the intrinsics twins of vso_fast.c.
 */
#include "VecSim/spaces/L2/L2.h"
#include "VecSim/spaces/IP/IP.h"
#include "VecSim/spaces/normalize/normalize_naive.h"
#include "VecSim/types/bfloat16.h"
#include "VecSim/types/float16.h"
#include "VecSim/utils/vecsim_stl.h"
#include "VecSim/utils/updatable_heap.h"
#include "VecSim/memory/vecsim_malloc.h"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace {
enum { T_F32 = 0, T_F64 = 1, T_BF16 = 2, T_F16 = 3, T_I8 = 4, T_U8 = 5 }; /* == VecSimType */
enum { M_L2 = 0, M_IP = 1, M_COSINE = 2 };                              /* == VecSimMetric */
} // namespace

/* the Flat top-k loop over the reference's containers; `multi` picks updatable_max_heap (brute_force_multi.h:108-112)
 * instead of max_priority_queue (brute_force_single.h:108-112).  DistT = float or double as the index would have it
 * (`wide` = 1: double).  Results ascending as the reply is drained (brute_force.h:284-288). */
template <typename DistT>
static size_t topk_loop(const double *scores, const size_t *labels, size_t n, size_t k, int multi, size_t *out_labels,
                        double *out_scores) {
    if (k == 0)
        return 0;
    auto alloc = VecSimAllocator::newVecsimAllocator();
    vecsim_stl::abstract_priority_queue<DistT, size_t> *top;
    if (multi)
        top = new (alloc) vecsim_stl::updatable_max_heap<DistT, size_t>(alloc);
    else
        top = new (alloc) vecsim_stl::max_priority_queue<DistT, size_t>(alloc);
    DistT upperBound = std::numeric_limits<DistT>::lowest();
    for (size_t id = 0; id < n; id++) {
        DistT score = (DistT)scores[id];
        if (score < upperBound || top->size() < k) {
            top->emplace(score, labels ? labels[id] : id);
            if (top->size() > k)
                top->pop();
            upperBound = top->top().first;
        }
    }
    size_t m = top->size();
    for (size_t i = m; i-- > 0;) {
        auto t = top->top();
        out_scores[i] = (double)t.first;
        out_labels[i] = t.second;
        top->pop();
    }
    delete top;
    return m;
}
extern "C" {

/* one distance by the reference's scalar kernel for (type, metric); fp Cosine == IP kernel (spaces.cpp:24-148) */
double vsref_distance(int type, int metric, size_t dim, const void *a, const void *b) {
    switch (type) {
    case T_F32:
        return metric == M_L2 ? FP32_L2Sqr(a, b, dim) : FP32_InnerProduct(a, b, dim);
    case T_F64:
        return metric == M_L2 ? FP64_L2Sqr(a, b, dim) : FP64_InnerProduct(a, b, dim);
    case T_BF16:
        return metric == M_L2 ? BF16_L2Sqr_LittleEndian(a, b, dim) : BF16_InnerProduct_LittleEndian(a, b, dim);
    case T_F16:
        return metric == M_L2 ? FP16_L2Sqr(a, b, dim) : FP16_InnerProduct(a, b, dim);
    case T_I8:
        return metric == M_L2 ? INT8_L2Sqr(a, b, dim)
                              : (metric == M_IP ? INT8_InnerProduct(a, b, dim) : INT8_Cosine(a, b, dim));
    case T_U8:
        return metric == M_L2 ? UINT8_L2Sqr(a, b, dim)
                              : (metric == M_IP ? UINT8_InnerProduct(a, b, dim) : UINT8_Cosine(a, b, dim));
    }
    return std::numeric_limits<double>::quiet_NaN();
}

/* in-place normalisation, the switch of VecSim_Normalize (vec_sim.cpp:238-254) over the reference's templates */
void vsref_normalize(void *blob, size_t dim, int type) {
    switch (type) {
    case T_F32: spaces::normalizeVector_imp<float>(blob, dim); break;
    case T_F64: spaces::normalizeVector_imp<double>(blob, dim); break;
    case T_BF16: spaces::bfloat16_normalizeVector<true>(blob, dim); break;
    case T_F16: spaces::float16_normalizeVector(blob, dim); break;
    case T_I8: spaces::integer_normalizeVector<int8_t>(blob, dim); break;
    case T_U8: spaces::integer_normalizeVector<uint8_t>(blob, dim); break;
    }
}

uint16_t vsref_f32_to_bf16(float f) { return vecsim_types::float_to_bf16(f).val; }
float vsref_bf16_to_f32(uint16_t h) { return vecsim_types::bfloat16_to_float32<true>(vecsim_types::bfloat16(h)); }
uint16_t vsref_f32_to_f16(float f) { return vecsim_types::FP32_to_FP16(f).val; }
float vsref_f16_to_f32(uint16_t h) { return vecsim_types::FP16_to_FP32(vecsim_types::float16(h)); }

void vsref_f32_to_bf16_n(const float *in, size_t n, uint16_t *out) { for (size_t i = 0; i < n; i++) out[i] = vecsim_types::float_to_bf16(in[i]).val; }
void vsref_f32_to_f16_n(const float *in, size_t n, uint16_t *out) { for (size_t i = 0; i < n; i++) out[i] = vecsim_types::FP32_to_FP16(in[i]).val; }
void vsref_bf16_to_f32_n(const uint16_t *in, size_t n, float *out) { for (size_t i = 0; i < n; i++) out[i] = vsref_bf16_to_f32(in[i]); }
void vsref_f16_to_f32_n(const uint16_t *in, size_t n, float *out) { for (size_t i = 0; i < n; i++) out[i] = vsref_f16_to_f32(in[i]); }

/* SQ8 scalar kernels (IP.cpp:34-183, L2.cpp:30-74,185-201); blobs laid out as types/sq8.h:19-62 says */
double vsref_sq8_fp32_distance(int metric, size_t dim, const void *storage, const void *query) {
    return metric == M_L2 ? SQ8_FP32_L2Sqr(storage, query, dim)
                          : (metric == M_IP ? SQ8_FP32_InnerProduct(storage, query, dim)
                                            : SQ8_FP32_Cosine(storage, query, dim));
}
double vsref_sq8_fp16_distance(int metric, size_t dim, const void *storage, const void *query) {
    return metric == M_L2 ? SQ8_FP16_L2Sqr(storage, query, dim)
                          : (metric == M_IP ? SQ8_FP16_InnerProduct(storage, query, dim)
                                            : SQ8_FP16_Cosine(storage, query, dim));
}
double vsref_sq8_sq8_distance(int metric, size_t dim, const void *a, const void *b) {
    return metric == M_L2 ? SQ8_SQ8_L2Sqr(a, b, dim)
                          : (metric == M_IP ? SQ8_SQ8_InnerProduct(a, b, dim) : SQ8_SQ8_Cosine(a, b, dim));
}

size_t vsref_topk(const double *scores, const size_t *labels, size_t n, size_t k, int multi, int wide,
                  size_t *out_labels, double *out_scores) {
    return wide ? topk_loop<double>(scores, labels, n, k, multi, out_labels, out_scores)
                : topk_loop<float>(scores, labels, n, k, multi, out_labels, out_scores);
}

/* A script of container operations on the reference's own heaps -- the ones the HNSW batch iterator keeps between batches
 * (hnsw_batch_iterator.h:33-52: candidates and top_candidates_extras are vecsim_stl::min_priority_queue, top_candidates is
 * max_priority_queue or, for multi-value indexes, updatable_max_heap).  kind 0 = min_priority_queue<double, size_t>,
 * 1 = max_priority_queue, 2 = updatable_max_heap.  op[i] = 0: emplace(score[i], label[i]); 1: pop (ignored when empty).
 * After every operation: out_size[i], and the top's score / label (NaN / ~0 when empty). */
void vsref_heap_script(int kind, const int *op, const double *score, const size_t *label, size_t n, size_t *out_size,
                       double *out_top_score, size_t *out_top_label) {
    auto alloc = VecSimAllocator::newVecsimAllocator();
    vecsim_stl::min_priority_queue<double, size_t> mn(alloc);
    vecsim_stl::max_priority_queue<double, size_t> mx(alloc);
    vecsim_stl::updatable_max_heap<double, size_t> up(alloc);
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

} /* extern "C" */
