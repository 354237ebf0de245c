// c_api.cpp -- extern "C" surface of libvecsim_amd.so (VecSim/vec_sim.h, query_results.h,
// vec_sim_gpu.h).  Thin: argument checks, dispatch to the index object, reply accessors.
#include <strings.h>

#include <algorithm>
#include <cassert>
#include <cerrno>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <queue>
#include <stdexcept>
#include <unordered_map>

#include "blob_prep.h"
#include "ref_heap.h"
#include "flat_index.h"
#include "host_tier.h"
#include "host_lane_eval.h"
#include "sq8_prep.h"
#include "hnsw_index.h"
#include "sharded_index.h"

using vsa::FlatIndex;

// ------------------------------------------------------------------ lifetime
extern "C" VecSimIndex *VecSimIndex_New(const VecSimParams *params) {
    if (!params) return nullptr;
    // index_factory.cpp:17-46 swallows construction failures and returns NULL; so do we
    if (params->algo == VecSimAlgo_HNSWLIB) {
        vsa::HnswIndex *hx = vsa::HnswIndex::create(params->algoParams.hnswParams, params->logCtx);
        if (!hx) std::fprintf(stderr, "vecsim_amd: cannot create HNSW index: %s\n", vsgpu_last_error());
        return hx;
    }
    if (params->algo != VecSimAlgo_BF) {
        std::fprintf(stderr, "vecsim_amd: only Flat and HNSW indexes are built by this back end\n");
        return nullptr;
    }
    const BFParams &bf = params->algoParams.bfParams;
    FlatIndex *ix = FlatIndex::create(bf, params->logCtx);
    if (!ix) std::fprintf(stderr, "vecsim_amd: cannot create GPU index: %s\n", vsgpu_last_error());
    return ix;
}
static std::mutex g_lifetime_mu;   // (index / iterator lifetime bookkeeping: flat_index.h live_iterators_)
extern "C" void VecSimIndex_Free(VecSimIndex *index) {
    if (!index) return;
    {
        std::lock_guard<std::mutex> lk(g_lifetime_mu);
        if (index->live_iterators_ > 0) {   // freed under its batch iterators: the last of them destroys it
            index->orphaned_ = true;
            return;
        }
    }
    delete index;
}

// ---- SQ8 storage (vec_sim_gpu.h): the reference has the spaces and the preprocessor (types/sq8.h, preprocessors.h:259-649,
// IP.cpp:34-183, L2.cpp:30-45,185-201) but no RAM index factory that selects them, so the constructor is an extension
extern "C" VecSimIndex *VecSimGpu_NewFlatSQ8(const BFParams *params, void *logCtx) {
    if (!params) return nullptr;
    FlatIndex *ix = FlatIndex::createSQ8(*params, logCtx);
    if (!ix) std::fprintf(stderr, "vecsim_amd: cannot create SQ8 GPU index: %s\n", vsgpu_last_error());
    return ix;
}
extern "C" VecSimIndex *VecSimGpu_NewFlatSQ8Centered(const BFParams *params, const float *mean, float mean_sum_squares, void *logCtx) {
    if (!params || !mean) return nullptr;
    FlatIndex *ix = FlatIndex::createSQ8(*params, logCtx, mean, mean_sum_squares);
    if (!ix) std::fprintf(stderr, "vecsim_amd: cannot create mean-centred SQ8 GPU index: %s\n", vsgpu_last_error());
    return ix;
}
extern "C" size_t VecSimGpu_SQ8_StorageBlobSizeCentered(size_t dim, VecSimMetric metric) { return vsa::sq8_storage_bytes(dim, metric, true); }
extern "C" size_t VecSimGpu_SQ8_QueryBlobSizeCentered(size_t dim, VecSimMetric metric) { return vsa::sq8_query_bytes(dim, metric, true, false); }
extern "C" void VecSimGpu_SQ8_QuantizeCentered(const float *vector, const float *mean, size_t dim, VecSimMetric metric, void *storage_blob) {
    std::vector<float> scratch(dim);
    vsa::sq8_quantize_centred(vector, mean, dim, metric, static_cast<uint8_t *>(storage_blob), scratch.data());
}
extern "C" void VecSimGpu_SQ8_QueryBlobCentered(const float *vector, const float *mean, size_t dim, VecSimMetric metric, float *query_blob) {
    std::vector<float> body(dim);
    for (size_t i = 0; i < dim; i++) body[i] = metric == VecSimMetric_L2 ? vector[i] - mean[i] : vector[i];
    float meta[2];
    vsa::sq8_query_meta_centred(body.data(), vector, mean, dim, metric, meta);
    std::memcpy(query_blob, body.data(), dim * sizeof(float));
    query_blob[dim] = meta[0];
    query_blob[dim + 1] = meta[1];
}
extern "C" double VecSimGpu_SQ8_StoredDistance(VecSimIndex *index, size_t label_a, size_t label_b) {
    auto *f = dynamic_cast<FlatIndex *>(index);
    return f ? f->storedDistance(label_a, label_b) : std::numeric_limits<double>::quiet_NaN();
}
extern "C" size_t VecSimGpu_SQ8_StorageBlobSize(size_t dim, VecSimMetric metric) { return vsa::sq8_storage_bytes(dim, metric); }
extern "C" size_t VecSimGpu_SQ8_QueryBlobSize(size_t dim, VecSimMetric metric) { return vsa::sq8_query_bytes(dim, metric); }
extern "C" void VecSimGpu_SQ8_Quantize(const float *vector, size_t dim, VecSimMetric metric, void *storage_blob) {
    vsa::sq8_quantize(vector, dim, metric, static_cast<uint8_t *>(storage_blob));
}
extern "C" void VecSimGpu_SQ8_QueryBlob(const float *vector, size_t dim, VecSimMetric metric, float *query_blob) {
    vsa::sq8_query_blob(vector, dim, metric, query_blob);
}

extern "C" size_t VecSimIndex_EstimateInitialSize(const VecSimParams *params) {
    (void)params;
    return sizeof(FlatIndex);
}
extern "C" size_t VecSimIndex_EstimateElementSize(const VecSimParams *params) {
    if (!params || params->algo != VecSimAlgo_BF) return 0;
    const BFParams &bf = params->algoParams.bfParams;
    // stored row + id->label slot (brute_force_factory.cpp:113-135 counts the same two terms)
    return vsa::blob_bytes(bf.type, bf.dim, bf.metric) + sizeof(labelType);
}

// ------------------------------------------------------------------ ingest
extern "C" int VecSimIndex_AddVector(VecSimIndex *index, const void *blob, size_t label) {
    return index->addVector(blob, label);
}
extern "C" int VecSimIndex_DeleteVector(VecSimIndex *index, size_t label) { return index->deleteVector(label); }
extern "C" size_t VecSimIndex_IndexSize(VecSimIndex *index) { return index->indexSize(); }
extern "C" long VecSimIndex_AddVectorsBulk(VecSimIndex *index, const void *blobs, const size_t *labels, size_t n) {
    return index->addBulk(blobs, labels, n);
}
extern "C" long VecSimIndex_AddSyntheticVectors(VecSimIndex *index, size_t n, uint64_t seed) {
    return index->addSynthetic(n, seed);
}

// ------------------------------------------------------------------ queries
extern "C" VecSimQueryReply *VecSimIndex_TopKQuery(VecSimIndex *index, const void *queryBlob, size_t k,
                                                   VecSimQueryParams *queryParams, VecSimQueryReply_Order order) {
    assert((order == BY_ID || order == BY_SCORE) && "Possible order values are only 'BY_ID' or 'BY_SCORE'");
    VecSimQueryReply *rep = index->topKQuery(queryBlob, k, queryParams);
    if (order == BY_ID) vsa::sort_reply(rep, BY_ID);
    return rep;
}
extern "C" int VecSimIndex_TopKQueryBatch(VecSimIndex *index, const void *queryBlobs, size_t nq, size_t queryStride,
                                          size_t k, VecSimQueryParams *queryParams, VecSimQueryReply_Order order,
                                          VecSimQueryReply **replies) {
    if (order != BY_ID && order != BY_SCORE) return -1;
    return index->topKQueryBatch(queryBlobs, nq, queryStride, k, queryParams, order, replies);
}
extern "C" int VecSimIndex_TopKQueryBatchArrays(VecSimIndex *index, const void *queryBlobs, size_t nq, size_t queryStride,
                                                size_t k, VecSimQueryParams *queryParams, VecSimQueryReply_Order order,
                                                int64_t *labels, double *scores, int *codes) {
    if (order != BY_ID && order != BY_SCORE) return -1;
    std::vector<VecSimQueryReply *> reps(nq, nullptr);
    int rc = index->topKQueryBatch(queryBlobs, nq, queryStride, k, queryParams, order, reps.data());
    if (rc) return rc;
    for (size_t q = 0; q < nq; q++) {
        const auto &r = reps[q]->results;
        for (size_t j = 0; j < k; j++) {
            labels[q * k + j] = j < r.size() ? (int64_t)r[j].id : -1;
            scores[q * k + j] = j < r.size() ? r[j].score : -1.0;
        }
        if (codes) codes[q] = (int)reps[q]->code;
        delete reps[q];
    }
    return 0;
}
extern "C" int VecSimIndex_TopKCandidatesBatch(VecSimIndex *index, const void *queryBlobs, size_t nq, size_t queryStride,
                                               size_t k, size_t cap, uint32_t *ids, size_t *labels, double *scores,
                                               uint32_t *counts) {
    return index->topKCandidates(queryBlobs, nq, queryStride, k, cap, ids, labels, scores, counts);
}

// Global replay over the shards' candidate lists; the merge itself lives in sharded_index.cpp
extern "C" int VecSimGpu_MergeTopK(size_t nq, size_t parts, size_t cap, const uint64_t *gids, const size_t *labels,
                                   const double *scores, const uint32_t *counts, size_t k, int64_t *out_labels,
                                   double *out_scores) {
    std::vector<size_t> ol(nq * k);
    std::vector<double> os(nq * k);
    std::vector<uint32_t> oc(nq);
    for (size_t i = 0; i < nq * k; i++) {
        out_labels[i] = -1;
        out_scores[i] = -1.0;
    }
    if (vsa::merge_topk(nq, parts, cap, gids, labels, scores, counts, k, ol.data(), os.data(), oc.data())) return -1;
    for (size_t q = 0; q < nq; q++)
        for (uint32_t j = 0; j < oc[q]; j++) {
            out_labels[q * k + j] = (int64_t)ol[q * k + j];
            out_scores[q * k + j] = os[q * k + j];
        }
    return 0;
}

extern "C" VecSimQueryReply *VecSimIndex_RangeQuery(VecSimIndex *index, const void *queryBlob, double radius,
                                                    VecSimQueryParams *queryParams, VecSimQueryReply_Order order) {
    // same contract as vec_sim.cpp:359-367: C++ exceptions cross the boundary on bad arguments
    if (order != BY_ID && order != BY_SCORE) throw std::runtime_error("Possible order values are only 'BY_ID' or 'BY_SCORE'");
    if (radius < 0) throw std::runtime_error("radius must be non-negative");
    return index->rangeQuery(queryBlob, radius, queryParams, order);
}
extern "C" double VecSimIndex_GetDistanceFrom_Unsafe(VecSimIndex *index, size_t label, const void *blob) {
    return index->getDistanceFrom(label, blob);
}
extern "C" bool VecSimIndex_PreferAdHocSearch(VecSimIndex *index, size_t subsetSize, size_t k, bool initial_check) {
    return index->preferAdHocSearch(subsetSize, k, initial_check);
}

extern "C" VecSimAdhocBfCtx *VecSimIndex_AdhocBfCtx_New(VecSimIndex *, const void *) { return nullptr; }
extern "C" void VecSimIndex_AdhocBfCtx_Free(VecSimAdhocBfCtx *) {}
extern "C" double VecSimIndex_AdhocBfCtx_GetDistanceFrom(VecSimAdhocBfCtx *, size_t) {
    return std::numeric_limits<double>::quiet_NaN();
}
extern "C" void VecSimIndex_AdhocBfCtx_GetExactDistances(VecSimAdhocBfCtx *, const size_t *, double *out, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = std::numeric_limits<double>::quiet_NaN();
}

// ------------------------------------------------------------------ runtime parameter strings
static bool positive_integer(const VecSimRawParam &p, long long *out) {
    char *end = nullptr;
    errno = 0;
    *out = std::strtoll(p.value, &end, 0);
    return !(*out <= 0 || *out == LLONG_MAX || errno != 0 || end != p.value + p.valLen);
}
extern "C" VecSimResolveCode VecSimIndex_ResolveParams(VecSimIndex *index, VecSimRawParam *rparams, int paramNum,
                                                       VecSimQueryParams *qparams, VecsimQueryType query_type) {
    if (!qparams || (!rparams && paramNum != 0)) return VecSimParamResolverErr_NullParam;
    std::memset(qparams, 0, sizeof *qparams);
    for (int i = 0; i < paramNum; i++) {
        const VecSimRawParam &p = rparams[i];
        if (!strcasecmp(p.name, "BATCH_SIZE")) {
            if (query_type != QUERY_TYPE_HYBRID) return VecSimParamResolverErr_InvalidPolicy_NHybrid;
            if (qparams->batchSize != 0) return VecSimParamResolverErr_AlreadySet;
            long long v;
            if (!positive_integer(p, &v)) return VecSimParamResolverErr_BadValue;
            qparams->batchSize = (size_t)v;
        } else if (!strcasecmp(p.name, "HYBRID_POLICY")) {
            if (query_type != QUERY_TYPE_HYBRID) return VecSimParamResolverErr_InvalidPolicy_NHybrid;
            if (qparams->searchMode != 0) return VecSimParamResolverErr_AlreadySet;
            if (!strcasecmp(p.value, VECSIM_POLICY_BATCHES)) qparams->searchMode = HYBRID_BATCHES;
            else if (!strcasecmp(p.value, VECSIM_POLICY_ADHOC_BF)) qparams->searchMode = HYBRID_ADHOC_BF;
            else return VecSimParamResolverErr_InvalidPolicy_NExits;
        } else if (!strcasecmp(p.name, "EF_RUNTIME")) {
            // vec_sim.cpp:47-66: HNSW only, not for range queries
            if (index->basicInfo().algo != VecSimAlgo_HNSWLIB || query_type == QUERY_TYPE_RANGE)
                return VecSimParamResolverErr_UnknownParam;
            if (qparams->hnswRuntimeParams.efRuntime != 0) return VecSimParamResolverErr_AlreadySet;
            long long v;
            if (!positive_integer(p, &v)) return VecSimParamResolverErr_BadValue;
            qparams->hnswRuntimeParams.efRuntime = (size_t)v;
        } else if (!strcasecmp(p.name, "EPSILON")) {
            // vec_sim.cpp:_ResolveParams_Epsilon: HNSW (or SVS) only, range queries only, a positive double
            if (index->basicInfo().algo != VecSimAlgo_HNSWLIB) return VecSimParamResolverErr_UnknownParam;
            if (query_type != QUERY_TYPE_RANGE) return VecSimParamResolverErr_InvalidPolicy_NRange;
            if (qparams->hnswRuntimeParams.epsilon != 0) return VecSimParamResolverErr_AlreadySet;
            char *end = nullptr;
            errno = 0;
            const double v = std::strtod(p.value, &end);
            if (v <= 0 || v == DBL_MAX || errno != 0 || end != p.value + p.valLen) return VecSimParamResolverErr_BadValue;
            qparams->hnswRuntimeParams.epsilon = v;
        } else {
            // RERANK / SVS knobs belong to paths this build does not have (vec_sim.cpp:72-165); like any
            // unknown name they are rejected
            return VecSimParamResolverErr_UnknownParam;
        }
    }
    if (qparams->searchMode == HYBRID_ADHOC_BF && qparams->batchSize > 0)
        return VecSimParamResolverErr_InvalidPolicy_AdHoc_With_BatchSize;
    if (qparams->searchMode == HYBRID_ADHOC_BF && index->basicInfo().algo == VecSimAlgo_HNSWLIB &&
        qparams->hnswRuntimeParams.efRuntime > 0)
        return VecSimParamResolverErr_InvalidPolicy_AdHoc_With_EfRuntime;
    if (qparams->searchMode != 0) index->setLastMode(qparams->searchMode);
    return VecSimParamResolver_OK;
}

// ------------------------------------------------------------------ blobs
extern "C" void VecSim_Normalize(void *blob, size_t dim, VecSimType type) { vsa::normalize_blob(blob, dim, type); }
extern "C" size_t VecSimParams_GetQueryBlobSize(VecSimType type, size_t dim, VecSimMetric metric) {
    return vsa::blob_bytes(type, dim, metric);
}

// ------------------------------------------------------------------ info
extern "C" VecSimIndexDebugInfo VecSimIndex_DebugInfo(VecSimIndex *index) { return index->debugInfo(); }
extern "C" VecSimIndexBasicInfo VecSimIndex_BasicInfo(VecSimIndex *index) { return index->basicInfo(); }
extern "C" VecSimIndexStatsInfo VecSimIndex_StatsInfo(VecSimIndex *index) { return index->statsInfo(); }
// Debug-info iterator: the same field names, types and order as the reference's
// BruteForceIndex::debugInfoIterator (brute_force.h:348-365), HNSWIndex::debugInfoIterator (hnsw.h:2216-2273) and
// addCommonInfoToIterator (vec_sim_index.h:271-310); strings from utils/vec_utils.cpp:22-75,186-250.
struct VecSimDebugInfoIterator {
    std::vector<VecSim_InfoField> fields;
    size_t pos = 0;
};
namespace {
const char *type_name(VecSimType t) {
    static const char *n[] = {"FLOAT32", "FLOAT64", "BFLOAT16", "FLOAT16", "INT8", "UINT8", "INT32", "INT64"};
    return (unsigned)t < 8 ? n[t] : nullptr;
}
const char *metric_name(VecSimMetric m) {
    return m == VecSimMetric_Cosine ? "COSINE" : m == VecSimMetric_IP ? "IP" : m == VecSimMetric_L2 ? "L2" : nullptr;
}
const char *mode_name(VecSearchMode m) {
    static const char *n[] = {"EMPTY_MODE", "STANDARD_KNN", "HYBRID_ADHOC_BF", "HYBRID_BATCHES", "HYBRID_BATCHES_TO_ADHOC_BF",
                              "RANGE_QUERY"};
    return (unsigned)m < 6 ? n[m] : nullptr;
}
VecSim_InfoField str_field(const char *name, const char *v) {
    VecSim_InfoField f{};
    f.fieldName = name;
    f.fieldType = INFOFIELD_STRING;
    f.fieldValue.stringValue = v;
    return f;
}
VecSim_InfoField u64_field(const char *name, uint64_t v) {
    VecSim_InfoField f{};
    f.fieldName = name;
    f.fieldType = INFOFIELD_UINT64;
    f.fieldValue.uintegerValue = v;
    return f;
}
VecSim_InfoField f64_field(const char *name, double v) {
    VecSim_InfoField f{};
    f.fieldName = name;
    f.fieldType = INFOFIELD_FLOAT64;
    f.fieldValue.floatingPointValue = v;
    return f;
}
}  // namespace
// the tier a new index would get on this host (VECSIM_GPU_TIER override, else CPUID): "AVX512" | "AVX512_BF16" | "AVX512_FP16" | "SCALAR"
extern "C" const char *VecSimGpu_HostTier(void) { return vsa::tier_name(vsa::resolve_tier()); }
// what this host lacks for its own reference build to run the restated order for `type` ("" = nothing; host_tier.h)
extern "C" const char *VecSimGpu_HostTierNote(VecSimType type) {
    static thread_local std::string note;
    note = vsa::reference_order_missing(vsa::host_features(), (int)type);
    return note.c_str();
}
extern "C" const char *VecSimGpu_IndexTier(VecSimIndex *index) { return index ? vsa::tier_name(index->distanceTier()) : ""; }
extern "C" int VecSimGpu_IndexDevice(VecSimIndex *index) {
    if (!index) return -1;
    const auto g = index->gpus();
    return g.empty() ? -1 : vsgpu_ctx_device(g[0]);
}
extern "C" VecSimDebugInfoIterator *VecSimIndex_DebugInfoIterator(VecSimIndex *index) {
    const VecSimIndexDebugInfo info = index->debugInfo();
    const CommonInfo &ci = info.commonInfo;
    const bool hnsw = ci.basicInfo.algo == VecSimAlgo_HNSWLIB;
    auto *it = new VecSimDebugInfoIterator;
    auto &f = it->fields;
    f.push_back(str_field("ALGORITHM", hnsw ? "HNSW" : "FLAT"));
    f.push_back(str_field("TYPE", type_name(ci.basicInfo.type)));
    f.push_back(u64_field("DIMENSION", ci.basicInfo.dim));
    f.push_back(str_field("METRIC", metric_name(ci.basicInfo.metric)));
    f.push_back(u64_field("IS_MULTI_VALUE", ci.basicInfo.isMulti));
    f.push_back(u64_field("IS_DISK", ci.basicInfo.isDisk));
    f.push_back(u64_field("INDEX_SIZE", ci.indexSize));
    f.push_back(u64_field("INDEX_LABEL_COUNT", ci.indexLabelCount));
    f.push_back(u64_field("MEMORY", ci.memory));
    f.push_back(str_field("LAST_SEARCH_MODE", mode_name(ci.lastMode)));
    f.push_back(u64_field("BLOCK_SIZE", ci.basicInfo.blockSize));
    if (hnsw) {
        f.push_back(u64_field("M", info.hnswInfo.M));
        f.push_back(u64_field("EF_CONSTRUCTION", info.hnswInfo.efConstruction));
        f.push_back(u64_field("EF_RUNTIME", info.hnswInfo.efRuntime));
        f.push_back(u64_field("MAX_LEVEL", info.hnswInfo.max_level));
        f.push_back(u64_field("ENTRYPOINT", info.hnswInfo.entrypoint));
        f.push_back(f64_field("EPSILON", info.hnswInfo.epsilon));
        f.push_back(u64_field("NUMBER_OF_MARKED_DELETED", info.hnswInfo.numberOfMarkedDeletedNodes));
    }
    // (exactly the reference's fields, info_iterator / vec_sim_index.h debugInfoIterator: consumers count them.  The tier an
    // index answers in is VecSimGpu_IndexTier.)
    return it;
}
extern "C" size_t VecSimDebugInfoIterator_NumberOfFields(VecSimDebugInfoIterator *it) { return it->fields.size(); }
extern "C" bool VecSimDebugInfoIterator_HasNextField(VecSimDebugInfoIterator *it) { return it->pos < it->fields.size(); }
extern "C" VecSim_InfoField *VecSimDebugInfoIterator_NextField(VecSimDebugInfoIterator *it) {
    return it->pos < it->fields.size() ? &it->fields[it->pos++] : nullptr;
}
extern "C" void VecSimDebugInfoIterator_Free(VecSimDebugInfoIterator *it) { delete it; }
extern "C" void VecSimTieredIndex_GC(VecSimIndex *) {}
extern "C" void VecSimTieredIndex_AcquireSharedLocks(VecSimIndex *) {}
extern "C" void VecSimTieredIndex_ReleaseSharedLocks(VecSimIndex *) {}

// ------------------------------------------------------------------ process-wide hooks
extern "C" void VecSim_SetMemoryFunctions(VecSimMemoryFunctions) {
    // host bookkeeping uses the C++ allocator; vector bytes live in HBM (pass-through)
}
extern "C" void VecSim_SetTimeoutCallbackFunction(timeoutCallbackFunction cb) { vsa::globals().timeout_cb = cb; }
extern "C" void VecSim_SetLogCallbackFunction(logCallbackFunction cb) { vsa::globals().log_cb = cb; }
extern "C" void VecSim_SetTestLogContext(const char *, const char *) {}
extern "C" void VecSim_SetWriteMode(VecSimWriteMode mode) { vsa::globals().write_mode = mode; }
extern "C" void VecSim_UpdateThreadPoolSize(size_t n) {
    vsa::globals().write_mode = n == 0 ? VecSim_WriteInPlace : VecSim_WriteAsync;
}
extern "C" size_t VecSim_GetSharedMemory(void) { return 0; }

// ------------------------------------------------------------------ GPU extension
extern "C" int VecSimGpu_SetDevice(int device) {
    if (device < 0 || device >= vsgpu_device_count()) return -1;
    vsa::globals().device = device;
    return 0;
}
extern "C" int VecSimGpu_DeviceCount(void) { return vsgpu_device_count(); }
extern "C" int VecSimGpu_DeviceSynchronize(void) {
    const int n = vsgpu_device_count();
    if (n <= 0) return -1;
    int d = vsa::globals().device;   // (as FlatIndex resolves it)
    if (d < 0) {
        if (const char *e = std::getenv("VECSIM_GPU_DEVICE")) d = std::atoi(e);
        else if (const char *e2 = std::getenv("LOCAL_RANK")) d = std::atoi(e2) % n;
        else d = 0;
    }
    return vsgpu_device_synchronize(d) == VSGPU_OK ? 0 : -1;
}
extern "C" const char *VecSimGpu_LastError(void) { return vsgpu_last_error(); }
extern "C" void VecSimGpu_ResetStats(VecSimIndex *index) {
    for (vsgpu_ctx *c : index->gpus()) vsgpu_stats_reset(c);
}
extern "C" void VecSimGpu_GetStats(VecSimIndex *index, VecSimGpuStats *out) {
    static_assert(sizeof(VecSimGpuStats) == sizeof(vsgpu_stats), "stats structs must stay in sync");
    vsgpu_stats s{};
    bool first = true;
    for (vsgpu_ctx *c : index->gpus()) {   // summed over the reader lanes
        vsgpu_stats l;
        vsgpu_stats_get(c, &l);
        if (first) {
            s = l;
            first = false;
            continue;
        }
        s.scan_ms += l.scan_ms;
        s.scan_launches += l.scan_launches;
        s.scan_rows += l.scan_rows;
        s.scan_bytes += l.scan_bytes;
        s.other_ms += l.other_ms;
        s.candidates += l.candidates;
        s.fallbacks += l.fallbacks;
        s.retries += l.retries;
        if (!s.scan_kernel[0]) std::memcpy(s.scan_kernel, l.scan_kernel, sizeof s.scan_kernel);
    }
    std::memcpy(out, &s, sizeof s);
}
extern "C" int VecSimGpu_SetOption(VecSimIndex *index, const char *name, long value) {
    int rc = 0;
    for (vsgpu_ctx *c : index->gpus()) rc |= vsgpu_set_option(c, name, value);
    return rc;
}

extern "C" int VecSimGpu_HnswGraphInfo(VecSimIndex *index, uint64_t info[6]) {
    auto *h = dynamic_cast<vsa::HnswIndex *>(index);
    if (!h) return -1;
    auto e = h->exportGraph();
    info[0] = e.n; info[1] = e.M; info[2] = e.M0; info[3] = e.entry;
    info[4] = e.max_level < 0 ? 0xFFFFFFFFull : (uint64_t)e.max_level;
    info[5] = e.upper_words;
    return 0;
}
extern "C" int VecSimGpu_HnswGraphCopy(VecSimIndex *index, uint32_t *links0, uint16_t *cnt0, uint32_t *upper_off,
                                       uint32_t *upper, uint8_t *deleted, uint64_t *labels) {
    auto *h = dynamic_cast<vsa::HnswIndex *>(index);
    if (!h) return -1;
    auto e = h->exportGraph();
    std::memcpy(links0, e.links0, (size_t)e.n * e.M0 * 4);
    std::memcpy(cnt0, e.cnt0, (size_t)e.n * 2);
    std::memcpy(upper_off, e.upper_off, (size_t)e.n * 4);
    if (e.upper_words) std::memcpy(upper, e.upper, e.upper_words * 4);
    std::memcpy(deleted, e.deleted, e.n);
    std::memcpy(labels, e.labels, (size_t)e.n * 8);
    return 0;
}
// Test hook (CPU, no GPU needed): the distance of two STORED blobs as the HNSW reference-order insert path computes it on the host
// (csrc/host/host_lane_eval.h walking csrc/lane_program.h).  type / metric: VecSimType / VecSimMetric of the index; tier: VSGPU_TIER_*
// (-1: the host's).  NaN when this (type, tier) has no host walker (the AVX512-FP16 tier's half accumulators).
extern "C" double VecSimGpu_HostLaneDistance(int type, int metric, int tier, size_t dim, const void *a, const void *b) {
    vsa::HostLaneEval ev;
    if (tier < 0) tier = vsa::resolve_tier(-1);
    if (tier == VSGPU_TIER_SCALAR) tier = VSGPU_TIER_AVX512;   // (as HnswIndex::create: small dims take the scalar order by themselves)
    if (!ev.init(type, metric, tier, dim)) return std::numeric_limits<double>::quiet_NaN();
    std::vector<float> wa, wb;
    if (type == VecSimType_FLOAT32 || type == VecSimType_BFLOAT16 || type == VecSimType_FLOAT16) {
        wa.resize(dim);
        wb.resize(dim);
        for (size_t i = 0; i < dim; i++) {
            if (type == VecSimType_FLOAT32) {
                std::memcpy(&wa[i], (const char *)a + 4 * i, 4);
                std::memcpy(&wb[i], (const char *)b + 4 * i, 4);
            } else {
                uint16_t ha, hb;
                std::memcpy(&ha, (const char *)a + 2 * i, 2);
                std::memcpy(&hb, (const char *)b + 2 * i, 2);
                wa[i] = type == VecSimType_BFLOAT16 ? vsa::bf16_widen(ha) : vsa::fp16_widen(ha);
                wb[i] = type == VecSimType_BFLOAT16 ? vsa::bf16_widen(hb) : vsa::fp16_widen(hb);
            }
        }
    }
    return ev.score((const char *)a, (const char *)b, wa.data(), wb.data());
}
// per-node top level [n] (with the arrays of VecSimGpu_HnswGraphCopy the whole graph); returns 1 when single AddVector calls follow the
// reference's insert path in the tier's distance order (hnsw_ref_build.cpp), 0 when the fast builder runs, -1: not an HNSW index
extern "C" int VecSimGpu_HnswLevels(VecSimIndex *index, uint8_t *levels) {
    auto *h = dynamic_cast<vsa::HnswIndex *>(index);
    if (!h) return -1;
    if (levels) std::memcpy(levels, h->levels(), h->exportGraph().n);
    return h->referenceOrderBuild() ? 1 : 0;
}
extern "C" long VecSimGpu_GetStoredVectors(VecSimIndex *index, size_t label, void *out, size_t cap_bytes, size_t *blob_bytes) {
    if (blob_bytes) *blob_bytes = index->storedBlobBytes();
    return out ? index->storedVectors(label, out, cap_bytes) : 0;
}
extern "C" long VecSimGpu_ReadStoredRows(VecSimIndex *index, size_t first_id, size_t n, void *out, size_t cap_bytes) {
    if (auto *h = dynamic_cast<vsa::HnswIndex *>(index)) {   // the graph's own row order (host copy)
        if (!out || cap_bytes < n * h->storedBlobBytes()) return -1;
        return h->readRows(first_id, n, out) ? -1 : (long)n;
    }
    auto *f = dynamic_cast<vsa::FlatIndex *>(index);
    if (!f || !out || cap_bytes < n * f->storedBlobBytes()) return -1;
    return f->readRows((uint32_t)first_id, n, out) ? -1 : (long)n;
}
// vec_sim_debug.h (reference: src/VecSim/vec_sim_debug.cpp:15-80, hnsw.h getHNSWElementNeighbors)
extern "C" int VecSimDebug_GetElementNeighborsInHNSWGraph(VecSimIndex *index, size_t label, int ***neighborsData) {
    *neighborsData = nullptr;
    auto *h = dynamic_cast<vsa::HnswIndex *>(index);
    if (!h) return VecSimDebugCommandCode_BadIndex;
    std::vector<std::vector<size_t>> levels;
    const int rc = h->neighborLabels(label, levels);
    if (rc == -2) return VecSimDebugCommandCode_MultiNotSupported;
    if (rc) return VecSimDebugCommandCode_LabelNotExists;
    int **out = new int *[levels.size() + 1];
    for (size_t l = 0; l < levels.size(); l++) {
        out[l] = new int[levels[l].size() + 1];
        out[l][0] = (int)levels[l].size();
        for (size_t i = 0; i < levels[l].size(); i++) out[l][i + 1] = (int)levels[l][i];
    }
    out[levels.size()] = nullptr;
    *neighborsData = out;
    return VecSimDebugCommandCode_OK;
}
extern "C" void VecSimDebug_ReleaseElementNeighborsInHNSWGraph(int **neighborsData) {
    if (!neighborsData) return;
    for (size_t l = 0; neighborsData[l]; l++) delete[] neighborsData[l];
    delete[] neighborsData;
}
extern "C" uint64_t VecSimGpu_HnswLastDistanceEvals(VecSimIndex *index) {
    auto *h = dynamic_cast<vsa::HnswIndex *>(index);
    return h ? h->lastDistanceEvals() : 0;
}

// ------------------------------------------------------------------ sharded Flat index (vec_sim_gpu.h)
namespace {
struct CallbackExchange final : vsa::Exchange {
    VecSimGpu_AllGatherFn ag;
    VecSimGpu_BroadcastFn bc;
    void *user;
    int allgather(const void *send, size_t bytes, void *recv) override { return ag(user, send, bytes, recv); }
    int broadcast(void *buf, size_t bytes, int root) override { return bc ? bc(user, buf, bytes, root) : -1; }
    bool canBroadcast() const override { return bc != nullptr; }
};
struct CallbackShard final : vsa::ShardOps {
    VecSimGpu_ShardAddFn add_fn;
    VecSimGpu_ShardCandidatesFn cand_fn;
    void *user;
    size_t rows = 0, stored = 0;
    int add(const void *blob, size_t label) override {
        const int rc = add_fn(user, blob, label);
        if (rc == 1) rows++;
        return rc;
    }
    int candidates(const void *queries, size_t nq, size_t stride, size_t k, size_t cap, uint32_t *ids, size_t *labels,
                   double *scores, uint32_t *counts) override {
        return cand_fn(user, queries, nq, stride, k, cap, ids, labels, scores, counts);
    }
    size_t size() const override { return rows; }
    size_t storedBytes() const override { return stored; }
};
VecSimShardedIndex *wrap(vsa::ShardedIndex *sx) {
    if (!sx) {
        std::fprintf(stderr, "vecsim_amd: cannot create sharded index: %s\n", vsgpu_last_error());
        return nullptr;
    }
    auto *w = new VecSimShardedIndex();
    w->impl.reset(sx);
    return w;
}
}  // namespace
extern "C" int VecSimGpu_ShardedGetUniqueId(void *id128) { return vsgpu_comm_unique_id(id128); }
extern "C" VecSimShardedIndex *VecSimGpu_ShardedNew(const VecSimParams *params, int rank, int world, int device,
                                                    const void *id128) {
    if (!params || params->algo != VecSimAlgo_BF) return nullptr;
    // the shard first (it owns the GPU context), then the communicator on that context
    vsa::ShardedIndex *sx = vsa::ShardedIndex::createDistributed(params->algoParams.bfParams, params->logCtx, rank, world,
                                                                 device, nullptr);
    if (!sx) return wrap(nullptr);
    auto ex = vsa::make_rccl_exchange(sx->localIndex(rank)->gpu(), rank, world, id128);
    if (!ex) {
        delete sx;
        return wrap(nullptr);
    }
    sx->setExchange(std::move(ex));
    return wrap(sx);
}
extern "C" VecSimShardedIndex *VecSimGpu_ShardedNewWithTransport(const VecSimParams *params, int rank, int world, int device,
                                                                 VecSimGpu_AllGatherFn allgather, VecSimGpu_BroadcastFn broadcast,
                                                                 void *user) {
    if (!params || params->algo != VecSimAlgo_BF || !allgather) return nullptr;
    auto ex = std::make_unique<CallbackExchange>();
    ex->ag = allgather;
    ex->bc = broadcast;
    ex->user = user;
    return wrap(vsa::ShardedIndex::createDistributed(params->algoParams.bfParams, params->logCtx, rank, world, device,
                                                     std::move(ex)));
}
extern "C" VecSimShardedIndex *VecSimGpu_ShardedNewExternal(const VecSimParams *params, int rank, int world,
                                                            VecSimGpu_ShardAddFn add, VecSimGpu_ShardCandidatesFn candidates,
                                                            VecSimGpu_AllGatherFn allgather, void *user) {
    if (!params || params->algo != VecSimAlgo_BF || !add || !candidates || !allgather) return nullptr;
    const BFParams &bf = params->algoParams.bfParams;
    auto ex = std::make_unique<CallbackExchange>();
    ex->ag = allgather;
    ex->bc = nullptr;
    ex->user = user;
    auto sh = std::make_unique<CallbackShard>();
    sh->add_fn = add;
    sh->cand_fn = candidates;
    sh->user = user;
    sh->stored = vsa::blob_bytes(bf.type, bf.dim, bf.metric);
    return wrap(vsa::ShardedIndex::createDistributed(bf, params->logCtx, rank, world, -1, std::move(ex), std::move(sh)));
}
extern "C" VecSimShardedIndex *VecSimGpu_ShardedNewLocal(const VecSimParams *params, int n_shards, const int *devices) {
    if (!params || params->algo != VecSimAlgo_BF) return nullptr;
    return wrap(vsa::ShardedIndex::createLocal(params->algoParams.bfParams, params->logCtx, n_shards, devices));
}
extern "C" void VecSimGpu_ShardedFree(VecSimShardedIndex *ix) { delete ix; }
extern "C" void VecSimGpu_ShardedAbort(VecSimShardedIndex *ix) { ix->impl->abortExchange(); }
extern "C" int VecSimGpu_ShardedExchangeSelfTest(VecSimShardedIndex *ix, size_t bytes) { return ix->impl->exchangeSelfTest(bytes); }
extern "C" const char *VecSimGpu_ShardedExchangeMode(VecSimShardedIndex *ix) { return ix->impl->exchangeMode(); }
extern "C" int VecSimGpu_ShardedAddVector(VecSimShardedIndex *ix, const void *blob, size_t label) {
    return ix->impl->addVector(blob, label);
}
extern "C" long VecSimGpu_ShardedAddVectorsBulk(VecSimShardedIndex *ix, const void *blobs, const size_t *labels, size_t n) {
    return ix->impl->addBulk(blobs, labels, n);
}
extern "C" long VecSimGpu_ShardedAddSyntheticLocal(VecSimShardedIndex *ix, size_t rows_per_shard, uint64_t seed_base) {
    return ix->impl->addSyntheticLocal(rows_per_shard, seed_base);
}
extern "C" int VecSimGpu_ShardedDeleteVector(VecSimShardedIndex *ix, size_t label) { return ix->impl->deleteVector(label); }
extern "C" size_t VecSimGpu_ShardedIndexSize(VecSimShardedIndex *ix) { return ix->impl->indexSize(); }
extern "C" int VecSimGpu_ShardedTopKQueryBatch(VecSimShardedIndex *ix, const void *queryBlobs, size_t nq, size_t queryStride,
                                               size_t k, VecSimQueryParams *queryParams, VecSimQueryReply_Order order,
                                               VecSimQueryReply **replies) {
    if (order != BY_ID && order != BY_SCORE) return -1;
    return ix->impl->topKQueryBatch(queryBlobs, nq, queryStride, k, queryParams, order, replies);
}
extern "C" int VecSimGpu_ShardedTopKQueryBatchArrays(VecSimShardedIndex *ix, const void *queryBlobs, size_t nq,
                                                     size_t queryStride, size_t k, VecSimQueryParams *queryParams,
                                                     VecSimQueryReply_Order order, int64_t *labels, double *scores, int *codes) {
    if (order != BY_ID && order != BY_SCORE) return -1;
    std::vector<VecSimQueryReply *> reps(nq, nullptr);
    int rc = ix->impl->topKQueryBatch(queryBlobs, nq, queryStride, k, queryParams, order, reps.data());
    if (rc) return rc;
    for (size_t q = 0; q < nq; q++) {
        const auto &r = reps[q]->results;
        for (size_t j = 0; j < k; j++) {
            labels[q * k + j] = j < r.size() ? (int64_t)r[j].id : -1;
            scores[q * k + j] = j < r.size() ? r[j].score : -1.0;
        }
        if (codes) codes[q] = (int)reps[q]->code;
        delete reps[q];
    }
    return 0;
}
extern "C" int VecSimGpu_ShardedTopKQueryBatchArraysSeq(VecSimShardedIndex *ix, const void *queryBlobs, size_t nq,
                                                        size_t queryStride, size_t k, VecSimQueryParams *queryParams,
                                                        VecSimQueryReply_Order order, int64_t *labels, double *scores, int *codes,
                                                        uint64_t seq) {
    if (order != BY_ID && order != BY_SCORE) return -1;
    std::vector<VecSimQueryReply *> reps(nq, nullptr);
    int rc = ix->impl->topKQueryBatch(queryBlobs, nq, queryStride, k, queryParams, order, reps.data(), seq);
    if (rc) return rc;
    for (size_t q = 0; q < nq; q++) {
        const auto &r = reps[q]->results;
        for (size_t j = 0; j < k; j++) {
            labels[q * k + j] = j < r.size() ? (int64_t)r[j].id : -1;
            scores[q * k + j] = j < r.size() ? r[j].score : -1.0;
        }
        if (codes) codes[q] = (int)reps[q]->code;
        delete reps[q];
    }
    return 0;
}
extern "C" void VecSimGpu_ShardedGetStats(VecSimShardedIndex *ix, double out[6]) { ix->impl->stats(out); }
extern "C" void VecSimGpu_ShardedResetStats(VecSimShardedIndex *ix) { ix->impl->resetStats(); }
extern "C" void VecSimGpu_ShardedResetSeq(VecSimShardedIndex *ix) { ix->impl->resetSeq(); }
extern "C" VecSimIndex *VecSimGpu_ShardedLocalIndex(VecSimShardedIndex *ix, int shard) { return ix->impl->localIndex(shard); }
extern "C" int VecSimGpu_ShardedWorld(VecSimShardedIndex *ix) { return ix->impl->world(); }
extern "C" int VecSimGpu_ShardedRank(VecSimShardedIndex *ix) { return ix->impl->rank(); }

// ------------------------------------------------------------------ replies
extern "C" int64_t VecSimQueryResult_GetId(const VecSimQueryResult *item) { return item ? (int64_t)item->id : (int64_t)INVALID_ID; }
extern "C" double VecSimQueryResult_GetScore(const VecSimQueryResult *item) {
    return item ? item->score : std::numeric_limits<double>::quiet_NaN();
}
extern "C" size_t VecSimQueryReply_Len(VecSimQueryReply *r) { return r->results.size(); }
extern "C" VecSimQueryReply_Code VecSimQueryReply_GetCode(VecSimQueryReply *r) { return r->code; }
extern "C" void VecSimQueryReply_Free(VecSimQueryReply *r) { delete r; }
extern "C" VecSimQueryReply_Iterator *VecSimQueryReply_GetIterator(VecSimQueryReply *r) {
    return new VecSimQueryReply_Iterator{r, 0};
}
extern "C" bool VecSimQueryReply_IteratorHasNext(VecSimQueryReply_Iterator *it) { return it->pos < it->reply->results.size(); }
extern "C" VecSimQueryResult *VecSimQueryReply_IteratorNext(VecSimQueryReply_Iterator *it) {
    if (it->pos >= it->reply->results.size()) return nullptr;
    return &it->reply->results[it->pos++];
}
extern "C" void VecSimQueryReply_IteratorReset(VecSimQueryReply_Iterator *it) { it->pos = 0; }
extern "C" void VecSimQueryReply_IteratorFree(VecSimQueryReply_Iterator *it) { delete it; }

// ------------------------------------------------------------------ batch iterator
// Behaviour of brute_force/bf_batch_iterator.h:61-199: all scores once (on the GPU), then per call
// either a bounded max-heap pass (few results out of many remaining) or an nth_element partition.
extern "C" VecSimBatchIterator *VecSimBatchIterator_New(VecSimIndex *index, const void *queryBlob,
                                                        VecSimQueryParams *queryParams) {
    VecSimBatchIterator *it = index->newBatchIterator(queryBlob, queryParams);
    if (it) {
        std::lock_guard<std::mutex> lk(g_lifetime_mu);
        index->live_iterators_++;
    }
    return it;
}

using ScoredLabel = std::pair<double, size_t>;

static VecSimQueryReply *next_by_heap(VecSimBatchIterator *it, size_t n_res) {
    auto *rep = new VecSimQueryReply();
    auto &sc = it->scores;
    vsa::RefMaxHeap<ScoredLabel> best;                 // max-heap on (score, label)
    std::unordered_map<size_t, size_t> slot_of_label;  // label -> position in sc
    double upper = std::numeric_limits<double>::lowest();
    for (size_t i = it->valid_start; i < sc.size(); i++) {
        if (best.size() >= n_res) {
            if (!(sc[i].first < upper)) continue;
            slot_of_label.erase(best.top().second);
            best.pop();
        }
        best.emplace(sc[i].first, sc[i].second);
        slot_of_label[sc[i].second] = i;
        upper = best.top().first;
    }
    const size_t got = best.size();
    rep->results.resize(got);
    for (size_t i = got; i-- > 0;) {
        rep->results[i].score = best.top().first;
        rep->results[i].id = best.top().second;
        best.pop();
    }
    // retire the returned entries: survivors sitting in the first `got` live slots move, in order,
    // into the slots (beyond that prefix) vacated by returned entries
    std::vector<size_t> taken;
    taken.reserve(got);
    for (auto &kv : slot_of_label) taken.push_back(kv.second);
    std::sort(taken.begin(), taken.end());
    const size_t next_start = it->valid_start + got;
    size_t hole = std::lower_bound(taken.begin(), taken.end(), next_start) - taken.begin();
    size_t t = 0;
    for (size_t pos = it->valid_start; pos < next_start; pos++) {
        if (t < taken.size() && taken[t] == pos) {
            t++;
        } else {
            sc[taken[hole++]] = sc[pos];
        }
    }
    it->valid_start = next_start;
    return rep;
}

static VecSimQueryReply *next_by_select(VecSimBatchIterator *it, size_t n_res) {
    auto *rep = new VecSimQueryReply();
    auto &sc = it->scores;
    const size_t remaining = sc.size() - it->valid_start;
    n_res = std::min(n_res, remaining);
    auto first = sc.begin() + (std::ptrdiff_t)it->valid_start;
    std::nth_element(first, first + (std::ptrdiff_t)n_res, sc.end());
    rep->results.reserve(n_res);
    for (size_t i = it->valid_start; i < it->valid_start + n_res; i++)
        rep->results.push_back(VecSimQueryResult{sc[i].second, sc[i].first});
    it->valid_start += n_res;
    return rep;
}

// ---- sparse mode: the score vector stays on the GPU --------------------------------------------------------------
// The reference's array after some batches = original order, minus a dead prefix of `valid_start` slots, with the
// survivors of each retired prefix moved into the slots vacated by returned entries (next_by_heap above).  Only
// those moves are tracked; a row's array position is moved_to[row] or the row id itself.
static bool sparse_next_by_heap(VecSimBatchIterator *it, size_t n_res, VecSimQueryReply **out) {
    const size_t cap = 2 * n_res + 64;
    std::vector<uint32_t> ids(cap);
    std::vector<double> sc(cap);
    uint32_t cnt = 0;
    if (it->index->iteratorDeviceNext(it->dev, n_res, cap, ids.data(), sc.data(), &cnt) || cnt == VSGPU_COUNT_OVERFLOW) return false;
    struct Ent {
        size_t pos;
        double score;
        uint32_t row;
    };
    std::vector<Ent> live(cnt);
    for (uint32_t i = 0; i < cnt; i++) {
        auto f = it->moved_to.find(ids[i]);
        live[i] = Ent{f == it->moved_to.end() ? (size_t)ids[i] : f->second, sc[i], ids[i]};
    }
    std::sort(live.begin(), live.end(), [](const Ent &a, const Ent &b) { return a.pos < b.pos; });
    // the reference's loop over the live range, restricted to the entries at or below the n_res-th smallest score
    using Item = std::pair<ScoredLabel, size_t>;  // ((score, label), index into live)
    vsa::RefMaxHeap<Item> best;
    double upper = std::numeric_limits<double>::lowest();
    for (size_t i = 0; i < live.size(); i++) {
        if (best.size() >= n_res) {
            if (!(live[i].score < upper)) continue;
            best.pop();
        }
        best.emplace(ScoredLabel(live[i].score, it->index->rowLabel(live[i].row)), i);
        upper = best.top().first.first;
    }
    auto *rep = new VecSimQueryReply();
    const size_t got = best.size();
    rep->results.resize(got);
    std::vector<size_t> taken;  // array positions of the returned entries
    std::vector<uint32_t> rows;
    taken.reserve(got);
    rows.reserve(got);
    for (size_t i = got; i-- > 0;) {
        rep->results[i].score = best.top().first.first;
        rep->results[i].id = best.top().first.second;
        taken.push_back(live[best.top().second].pos);
        rows.push_back(live[best.top().second].row);
        best.pop();
    }
    std::sort(taken.begin(), taken.end());
    for (uint32_t r : rows) it->moved_to.erase(r);
    const size_t next_start = it->valid_start + got;
    size_t hole = std::lower_bound(taken.begin(), taken.end(), next_start) - taken.begin();
    size_t t = 0;
    for (size_t pos = it->valid_start; pos < next_start; pos++) {
        auto at = it->moved_at.find(pos);
        const uint32_t row = at == it->moved_at.end() ? (uint32_t)pos : at->second;
        if (at != it->moved_at.end()) it->moved_at.erase(at);
        if (t < taken.size() && taken[t] == pos) {
            t++;  // a returned entry inside the retired prefix: nothing to move
        } else {
            const size_t h = taken[hole++];
            it->moved_to[row] = h;
            it->moved_at[h] = row;
        }
    }
    it->valid_start = next_start;
    if (it->index->iteratorDeviceRetire(it->dev, rows.data(), rows.size())) {
        delete rep;
        return false;
    }
    *out = rep;
    return true;
}
// leave sparse mode: rebuild the reference's array (same order, same dead prefix) on the host
static bool sparse_materialize(VecSimBatchIterator *it) {
    const size_t n = it->dev_rows;
    std::unique_ptr<double[]> all(new double[n]);
    if (it->index->iteratorDeviceRead(it->dev, all.get())) return false;
    it->scores.assign(n, ScoredLabel(0.0, 0));
    for (size_t pos = it->valid_start; pos < n; pos++) {
        auto at = it->moved_at.find(pos);
        const uint32_t row = at == it->moved_at.end() ? (uint32_t)pos : at->second;
        it->scores[pos] = ScoredLabel(all[row], it->index->rowLabel(row));
    }
    if (it->label_count < n) {
        // rows retired before the first batch (deleted HNSW nodes) carry NaN: they were never part of the array
        size_t w = it->valid_start;
        for (size_t pos = it->valid_start; pos < n; pos++)
            if (!std::isnan(it->scores[pos].first)) it->scores[w++] = it->scores[pos];
        it->scores.resize(w);
    }
    it->index->iteratorDeviceEnd(it->dev);
    it->dev = nullptr;
    it->moved_to.clear();
    it->moved_at.clear();
    it->scored = true;
    return true;
}

extern "C" VecSimQueryReply *VecSimBatchIterator_Next(VecSimBatchIterator *it, size_t n_results,
                                                      VecSimQueryReply_Order order) {
    assert((order == BY_ID || order == BY_SCORE) && "Possible order values are only 'BY_ID' or 'BY_SCORE'");
    if (it->walker) return it->walker->next(n_results, order);
    auto timed_out_reply = []() {
        auto *r = new VecSimQueryReply();
        r->code = VecSim_QueryReply_TimedOut;
        return r;
    };
    if (!it->scored && !it->dev_tried) {
        if (vsa::timed_out(it->timeout_ctx)) return timed_out_reply();
        it->dev_tried = true;
        // VECSIM_ITER_HOST=1 keeps the reference's host-side array from the first batch on (tests compare the two)
        it->dev = std::getenv("VECSIM_ITER_HOST") ? nullptr : it->index->iteratorDeviceBegin(it->query.data());
        if (it->dev) {
            it->dev_rows = vsgpu_scorebuf_rows(it->dev);
            it->label_count = it->index->indexLabelCount();  // (HNSW: rows minus deleted nodes, retired up front)
        }
    }
    if (it->dev) {
        if (vsa::timed_out(it->timeout_ctx)) return timed_out_reply();
        // small batches out of many live entries: the heap regime of the reference (bf_batch_iterator.h:140-150)
        if ((it->label_count - it->returned) / 1000 > n_results) {
            VecSimQueryReply *rep = nullptr;
            if (sparse_next_by_heap(it, n_results, &rep)) {
                it->returned += rep->results.size();
                if (order == BY_ID) vsa::sort_reply(rep, BY_ID);
                return rep;
            }
        }
        if (!sparse_materialize(it)) {
            std::fprintf(stderr, "vecsim_amd: GPU score pass failed: %s\n", vsgpu_last_error());
            return timed_out_reply();
        }
    }
    if (!it->scored) {
        if (vsa::timed_out(it->timeout_ctx)) return timed_out_reply();
        if (it->index->iteratorScores(it->query.data(), it->scores)) {
            std::fprintf(stderr, "vecsim_amd: GPU score pass failed: %s\n", vsgpu_last_error());
            return timed_out_reply();
        }
        it->label_count = it->scores.size();
        it->scored = true;
    }
    if (vsa::timed_out(it->timeout_ctx)) return timed_out_reply();
    VecSimQueryReply *rep;
    if ((it->label_count - it->returned) / 1000 > n_results) {
        rep = next_by_heap(it, n_results);  // already ascending by score
    } else {
        rep = next_by_select(it, n_results);
        if (order == BY_SCORE) vsa::sort_reply(rep, BY_SCORE);
        else if (order == BY_SCORE_THEN_ID) vsa::sort_reply(rep, BY_SCORE_THEN_ID);
    }
    it->returned += rep->results.size();
    if (order == BY_ID) vsa::sort_reply(rep, BY_ID);
    return rep;
}
extern "C" bool VecSimBatchIterator_HasNext(VecSimBatchIterator *it) {
    return it->walker ? !it->walker->depleted() : it->returned < it->label_count;
}
extern "C" void VecSimBatchIterator_Free(VecSimBatchIterator *it) {
    if (!it) return;
    VecSimIndexInterface *index = it->index;
    if (it->dev) index->iteratorDeviceEnd(it->dev);
    delete it;
    bool last_of_orphan;
    {
        std::lock_guard<std::mutex> lk(g_lifetime_mu);
        last_of_orphan = --index->live_iterators_ == 0 && index->orphaned_;
    }
    if (last_of_orphan) delete index;
}
extern "C" void VecSimBatchIterator_Reset(VecSimBatchIterator *it) {
    if (it->walker) {
        it->walker->reset();
        return;
    }
    if (it->dev) it->index->iteratorDeviceEnd(it->dev);
    it->dev = nullptr;
    it->dev_tried = false;
    it->moved_to.clear();
    it->moved_at.clear();
    it->scores.clear();
    it->scored = false;
    it->valid_start = 0;
    it->returned = 0;
}
