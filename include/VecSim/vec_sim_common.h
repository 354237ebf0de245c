/*
 * VecSim/vec_sim_common.h -- parameter, info and callback types of the C boundary.
 *
 * Binary-compatible restatement of the reference's public types (src/VecSim/vec_sim_common.h:60-502):
 * identical names, field order, enum values and therefore identical sizeof/offsetof, so a caller
 * compiled against the reference headers can link against libvecsim_amd.so unchanged.
 * tests/test_abi.py checks every size and offset below against a fixture generated from the
 * reference header (tests/golden/abi_layout.json, made by tests/golden/make_abi_layout.py).
 *
 * Flat (VecSimAlgo_BF) and HNSW (VecSimAlgo_HNSWLIB) indexes are constructible in this build; the tiered / SVS
 * parameter blocks are declared because they fix the size of the AlgoParams union.
 */
#pragma once

#ifdef __cplusplus
extern "C" {
#endif
#include <limits.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

/* ---- constants ---- */
#define DEFAULT_BLOCK_SIZE 1024
#define INVALID_ID         UINT_MAX
#define INVALID_LABEL      SIZE_MAX
#define UNUSED(x)          (void)(x)
#define VecSim_OK          0

#define VECSIM_POLICY_ADHOC_BF "adhoc_bf"
#define VECSIM_POLICY_BATCHES  "batches"
#define VECSIM_POLICY_INVALID  "invalid_policy"

#define HNSW_DEFAULT_M       16
#define HNSW_DEFAULT_EF_C    200
#define HNSW_DEFAULT_EF_RT   10
#define HNSW_DEFAULT_EPSILON 0.01
#define HNSW_INVALID_LEVEL   SIZE_MAX
#define INVALID_JOB_ID       UINT_MAX
#define INVALID_INFO         UINT_MAX

/* ---- scalar enums ---- */
typedef enum {
    VecSimType_FLOAT32,
    VecSimType_FLOAT64,
    VecSimType_BFLOAT16,
    VecSimType_FLOAT16,
    VecSimType_INT8,
    VecSimType_UINT8,
    VecSimType_INT32,
    VecSimType_INT64
} VecSimType;

typedef enum { VecSimAlgo_BF, VecSimAlgo_HNSWLIB, VecSimAlgo_TIERED, VecSimAlgo_SVS } VecSimAlgo;
typedef enum { VecSimMetric_L2, VecSimMetric_IP, VecSimMetric_Cosine } VecSimMetric;
typedef enum { VecSimOption_AUTO = 0, VecSimOption_ENABLE = 1, VecSimOption_DISABLE = 2 } VecSimOptionMode;
typedef enum { VecSimBool_TRUE = 1, VecSimBool_FALSE = 0, VecSimBool_UNSET = -1 } VecSimBool;
typedef enum { VecSim_WriteAsync, VecSim_WriteInPlace } VecSimWriteMode;

typedef size_t labelType;
typedef unsigned int idType;

typedef enum {
    VecSimParamResolver_OK = VecSim_OK,
    VecSimParamResolverErr_NullParam,
    VecSimParamResolverErr_AlreadySet,
    VecSimParamResolverErr_UnknownParam,
    VecSimParamResolverErr_BadValue,
    VecSimParamResolverErr_InvalidPolicy_NExits,
    VecSimParamResolverErr_InvalidPolicy_NHybrid,
    VecSimParamResolverErr_InvalidPolicy_NRange,
    VecSimParamResolverErr_InvalidPolicy_AdHoc_With_BatchSize,
    VecSimParamResolverErr_InvalidPolicy_AdHoc_With_EfRuntime
} VecSimResolveCode;

typedef enum {
    VecSimDebugCommandCode_OK = VecSim_OK,
    VecSimDebugCommandCode_BadIndex,
    VecSimDebugCommandCode_LabelNotExists,
    VecSimDebugCommandCode_MultiNotSupported
} VecSimDebugCommandCode;

typedef enum {
    EMPTY_MODE,
    STANDARD_KNN,
    HYBRID_ADHOC_BF,
    HYBRID_BATCHES,
    HYBRID_BATCHES_TO_ADHOC_BF,
    RANGE_QUERY
} VecSearchMode;

typedef enum { QUERY_TYPE_NONE, QUERY_TYPE_KNN, QUERY_TYPE_HYBRID, QUERY_TYPE_RANGE } VecsimQueryType;

typedef enum {
    HNSW_INSERT_VECTOR_JOB,
    HNSW_REPAIR_NODE_CONNECTIONS_JOB,
    HNSW_SEARCH_JOB,
    HNSW_SWAP_JOB,
    HNSW_DISK_JOB,
    SVS_BATCH_UPDATE_JOB,
    SVS_GC_JOB,
    INVALID_JOB
} JobType;

typedef enum {
    VecSimSvsQuant_NONE = 0,
    VecSimSvsQuant_Scalar = 1,
    VecSimSvsQuant_4 = 4,
    VecSimSvsQuant_8 = 8,
    VecSimSvsQuant_4x4 = 4 | (4 << 8),
    VecSimSvsQuant_4x8 = 4 | (8 << 8),
    VecSimSvsQuant_4x8_LeanVec = 4 | (8 << 8) | (1 << 16),
    VecSimSvsQuant_8x8_LeanVec = 8 | (8 << 8) | (1 << 16)
} VecSimSvsQuantBits;

/* ---- callbacks ---- */
typedef struct AsyncJob AsyncJob;
typedef void (*JobCallback)(AsyncJob *);
typedef int (*SubmitCB)(void *job_queue, void *index_ctx, AsyncJob **jobs, JobCallback *CBs, size_t jobs_len);
typedef int (*ThrottleCB)(void);
typedef int (*timeoutCallbackFunction)(void *ctx); /* non-zero => the query timed out */
typedef void (*logCallbackFunction)(void *ctx, const char *level, const char *message);

typedef void *(*allocFn)(size_t n);
typedef void *(*callocFn)(size_t nelem, size_t elemsz);
typedef void *(*reallocFn)(void *p, size_t n);
typedef void (*freeFn)(void *p);
typedef char *(*strdupFn)(const char *s);
typedef struct {
    allocFn allocFunction;
    callocFn callocFunction;
    reallocFn reallocFunction;
    freeFn freeFunction;
} VecSimMemoryFunctions;

/* ---- raw (string) runtime parameter, input of VecSimIndex_ResolveParams ---- */
typedef struct {
    const char *name;
    size_t nameLen;
    const char *value;
    size_t valLen;
} VecSimRawParam;

/* ---- construction parameters ---- */
typedef struct VecSimParams VecSimParams;

typedef struct {
    VecSimType type;
    size_t dim;
    VecSimMetric metric;
    bool multi;
    size_t initialCapacity; /* deprecated upstream, ignored here */
    size_t blockSize;
} BFParams;

typedef struct {
    VecSimType type;
    size_t dim;
    VecSimMetric metric;
    bool multi;
    size_t initialCapacity;
    size_t blockSize;
    size_t M;
    size_t efConstruction;
    size_t efRuntime;
    double epsilon;
} HNSWParams;

typedef struct {
    VecSimType type;
    size_t dim;
    VecSimMetric metric;
    bool multi;
    size_t blockSize;
    VecSimSvsQuantBits quantBits;
    float alpha;
    size_t graph_max_degree;
    size_t construction_window_size;
    size_t max_candidate_pool_size;
    size_t prune_to;
    VecSimOptionMode use_search_history;
    size_t num_threads;
    size_t search_window_size;
    size_t search_buffer_capacity;
    size_t leanvec_dim;
    double epsilon;
} SVSParams;

typedef struct { size_t swapJobThreshold; } TieredHNSWParams;
typedef struct { char _placeholder; } TieredHNSWDiskParams;
typedef struct {
    size_t trainingTriggerThreshold;
    size_t updateTriggerThreshold;
    size_t updateJobWaitTime;
} TieredSVSParams;

typedef struct {
    void *jobQueue;
    void *jobQueueCtx;
    SubmitCB submitCb;
    size_t flatBufferLimit;
    VecSimParams *primaryIndexParams;
    union {
        TieredHNSWParams tieredHnswParams;
        TieredSVSParams tieredSVSParams;
        TieredHNSWDiskParams tieredHnswDiskParams;
    } specificParams;
} TieredIndexParams;

typedef union {
    HNSWParams hnswParams;
    BFParams bfParams;
    TieredIndexParams tieredParams;
    SVSParams svsParams;
} AlgoParams;

struct VecSimParams {
    VecSimAlgo algo;
    AlgoParams algoParams;
    void *logCtx;
};

typedef struct {
    void *storage;
    const char *indexName;
    size_t indexNameLen;
    uint32_t userData;
    bool rerank;
} VecSimDiskContext;

typedef struct {
    VecSimParams *indexParams;
    VecSimDiskContext *diskContext;
} VecSimParamsDisk;

/* ---- per-query runtime parameters ---- */
typedef struct {
    size_t efRuntime;
    double epsilon;
} HNSWRuntimeParams;
typedef struct {
    size_t efRuntime;
    double epsilon;
    VecSimBool shouldRerank;
} HNSWDiskRuntimeParams;
typedef struct {
    size_t windowSize;
    size_t bufferCapacity;
    VecSimOptionMode searchHistory;
    double epsilon;
} SVSRuntimeParams;

typedef struct {
    union {
        HNSWRuntimeParams hnswRuntimeParams;
        HNSWDiskRuntimeParams hnswDiskRuntimeParams;
        SVSRuntimeParams svsRuntimeParams;
    };
    size_t batchSize;
    VecSearchMode searchMode;
    void *timeoutCtx; /* handed to the timeout callback; polled once per kernel launch here */
} VecSimQueryParams;

/* ---- info structs ---- */
typedef struct {
    VecSimAlgo algo;
    VecSimMetric metric;
    VecSimType type;
    bool isMulti;
    bool isTiered;
    bool isDisk;
    size_t blockSize;
    size_t dim;
} VecSimIndexBasicInfo;

typedef struct {
    size_t memory;
    size_t numberOfMarkedDeleted;
    size_t directHNSWInsertions;
    size_t flatBufferSize;
} VecSimIndexStatsInfo;

typedef struct {
    VecSimIndexBasicInfo basicInfo;
    size_t indexSize;
    size_t indexLabelCount;
    uint64_t memory;
    VecSearchMode lastMode;
} CommonInfo;

typedef struct {
    size_t M;
    size_t efConstruction;
    size_t efRuntime;
    double epsilon;
    size_t max_level;
    size_t entrypoint;
    size_t visitedNodesPoolSize;
    size_t numberOfMarkedDeletedNodes;
} hnswInfoStruct;

typedef struct { char dummy; } bfInfoStruct;

typedef struct {
    VecSimSvsQuantBits quantBits;
    float alpha;
    size_t graphMaxDegree;
    size_t constructionWindowSize;
    size_t maxCandidatePoolSize;
    size_t pruneTo;
    bool useSearchHistory;
    size_t numThreads;
    size_t lastReservedThreads;
    size_t numberOfMarkedDeletedNodes;
    size_t searchWindowSize;
    size_t searchBufferCapacity;
    size_t leanvecDim;
    double epsilon;
} svsInfoStruct;

typedef struct HnswTieredInfo { size_t pendingSwapJobsThreshold; } HnswTieredInfo;
typedef struct SvsTieredInfo {
    size_t trainingTriggerThreshold;
    size_t updateTriggerThreshold;
    size_t updateJobWaitTime;
    bool indexUpdateScheduled;
} SvsTieredInfo;

typedef struct {
    union {
        hnswInfoStruct hnswInfo;
        svsInfoStruct svsInfo;
    } backendInfo;
    union {
        HnswTieredInfo hnswTieredInfo;
        SvsTieredInfo svsTieredInfo;
    } specificTieredBackendInfo;
    CommonInfo backendCommonInfo;
    CommonInfo frontendCommonInfo;
    bfInfoStruct bfInfo;
    uint64_t management_layer_memory;
    VecSimBool backgroundIndexing;
    size_t bufferLimit;
} tieredInfoStruct;

typedef struct {
    CommonInfo commonInfo;
    union {
        bfInfoStruct bfInfo;
        hnswInfoStruct hnswInfo;
        svsInfoStruct svsInfo;
        tieredInfoStruct tieredInfo;
    };
} VecSimIndexDebugInfo;

static inline size_t RoundUpInitialCapacity(size_t initialCapacity, size_t blockSize) {
    size_t rem = initialCapacity % blockSize;
    return rem ? initialCapacity + (blockSize - rem) : initialCapacity;
}

#ifdef __cplusplus
}
#endif
