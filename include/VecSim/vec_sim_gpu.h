/*
 * VecSim/vec_sim_gpu.h -- additions to the reference C API that the GPU back end needs.
 *
 * Nothing in the reference can express more than one query per call (the Python binding loops
 * single queries, src/python_bindings/bindings.cpp:182-190), and one query cannot fill an MI355X:
 * the batched entry point below is the only new *query* call.  The rest is ingest/measurement
 * plumbing.  Existing structs keep their layout; all knobs come through these calls or the
 * VECSIM_GPU_* environment variables.
 */
#pragma once
#include "vec_sim.h"

#ifdef __cplusplus
extern "C" {
#endif

/* nq queries, `queryStride` bytes apart (>= VecSimParams_GetQueryBlobSize).  replies[i] receives
 * exactly what VecSimIndex_TopKQuery(index, query_i, k, queryParams, order) would return.
 * Returns 0, or non-zero after a GPU failure (replies untouched). */
int VecSimIndex_TopKQueryBatch(VecSimIndex *index, const void *queryBlobs, size_t nq, size_t queryStride,
                               size_t k, VecSimQueryParams *queryParams, VecSimQueryReply_Order order,
                               VecSimQueryReply **replies);

/* Same query, results written straight into caller arrays [nq][k] (labels padded with -1, scores with
 * -1.0, like the reference's Python wrap_results, bindings.cpp:36-72); codes[q] (may be NULL) receives
 * the VecSimQueryReply_Code.  Saves nq reply objects + iterators per batch for array-oriented callers. */
int VecSimIndex_TopKQueryBatchArrays(VecSimIndex *index, const void *queryBlobs, size_t nq, size_t queryStride,
                                     size_t k, VecSimQueryParams *queryParams, VecSimQueryReply_Order order,
                                     int64_t *labels, double *scores, int *codes);

/* Shard-side half of a multi-GPU query: for every query, EVERY local row with score <= T_local
 * (T_local = the k-th smallest local score; all rows when the shard holds fewer than k), ascending
 * internal id.  That is the superset the global sequential replay needs.
 * ids/labels/scores are [nq][cap]; counts[q] = rows written, or 0xFFFFFFFF when more than `cap`
 * rows tie at or below T_local. */
int VecSimIndex_TopKCandidatesBatch(VecSimIndex *index, const void *queryBlobs, size_t nq, size_t queryStride,
                                    size_t k, size_t cap, uint32_t *ids, size_t *labels, double *scores,
                                    uint32_t *counts);

/* Merge side: `parts` shards' candidate lists (layout [part][nq][cap], counts [part][nq], gids =
 * the row's internal id in the equivalent single index) -> the reply the reference's sequential
 * heap would give over the union scanned in gid order.  out_* are [nq][k], padded with -1 / -1.0.
 * Returns 0, or -1 if any count is the overflow marker. */
int VecSimGpu_MergeTopK(size_t nq, size_t parts, size_t cap, const uint64_t *gids, const size_t *labels,
                        const double *scores, const uint32_t *counts, size_t k, int64_t *out_labels,
                        double *out_scores);

/* ---- Flat index sharded over the GPUs of a node (BASELINE config 4) ----
 * The reference is single-process and CPU-only, so nothing upstream maps to these; what they preserve is the reply
 * of BruteForceIndex::topKQuery (algorithms/brute_force/brute_force.h:242-291) over the union of the shards,
 * ties included.  Vector number i of the equivalent single index lives in block i / blockSize, block b on shard
 * b % world.  Ingest and queries are SPMD: every process makes the same calls in the same order; each keeps only
 * its own shard's rows (ingest moves no data) and every process receives the full reply.
 *
 *   one process per GPU   VecSimGpu_ShardedNew: rank 0 draws 128 bytes with VecSimGpu_ShardedGetUniqueId and hands
 *                         them to the other ranks out of band (env, file, MPI, a torch store); the per-batch
 *                         exchange is ONE ncclAllGather of fixed-size candidate records over RCCL/xGMI.
 *   own transport         VecSimGpu_ShardedNewWithTransport: same, the caller moves the bytes (MPI, gloo ...).
 *   one process, G shards VecSimGpu_ShardedNewLocal: devices[i] is shard i's GPU (may repeat); no communicator,
 *                         the shards scan concurrently and the partials are merged in-process.
 *   external shard        VecSimGpu_ShardedNewExternal: the caller also provides this rank's storage and scan
 *                         (another back end); append-only.
 */
typedef struct VecSimShardedIndex VecSimShardedIndex;
/* recv receives world * bytes in rank order / buf of `root` reaches every rank; return 0 on success */
typedef int (*VecSimGpu_AllGatherFn)(void *user, const void *send, size_t bytes, void *recv);
typedef int (*VecSimGpu_BroadcastFn)(void *user, void *buf, size_t bytes, int root);
/* external shard: add returns 1 (appended) / 0 (overwrote) / -1; candidates has VecSimIndex_TopKCandidatesBatch's contract */
typedef int (*VecSimGpu_ShardAddFn)(void *user, const void *blob, size_t label);
typedef int (*VecSimGpu_ShardCandidatesFn)(void *user, const void *queryBlobs, size_t nq, size_t queryStride, size_t k,
                                           size_t cap, uint32_t *ids, size_t *labels, double *scores, uint32_t *counts);
int VecSimGpu_ShardedGetUniqueId(void *id128);
VecSimShardedIndex *VecSimGpu_ShardedNew(const VecSimParams *params, int rank, int world, int device, const void *id128);
VecSimShardedIndex *VecSimGpu_ShardedNewWithTransport(const VecSimParams *params, int rank, int world, int device,
                                                      VecSimGpu_AllGatherFn allgather, VecSimGpu_BroadcastFn broadcast,
                                                      void *user);
VecSimShardedIndex *VecSimGpu_ShardedNewExternal(const VecSimParams *params, int rank, int world, VecSimGpu_ShardAddFn add,
                                                 VecSimGpu_ShardCandidatesFn candidates, VecSimGpu_AllGatherFn allgather,
                                                 void *user);
VecSimShardedIndex *VecSimGpu_ShardedNewLocal(const VecSimParams *params, int n_shards, const int *devices);
void VecSimGpu_ShardedFree(VecSimShardedIndex *index);
/* 1 new / 0 overwrite (the row keeps its place, like brute_force_single.h:139-143) / -1 error */
int VecSimGpu_ShardedAddVector(VecSimShardedIndex *index, const void *blob, size_t label);
long VecSimGpu_ShardedAddVectorsBulk(VecSimShardedIndex *index, const void *blobs, const size_t *labels, size_t n);
/* weak-scaling fill: every shard s appends rows_per_shard device-generated rows (seed_base + 1000 s); the equivalent
 * single index is the concatenation of the shards in shard order, label = gid.  Append-only afterwards. */
long VecSimGpu_ShardedAddSyntheticLocal(VecSimShardedIndex *index, size_t rows_per_shard, uint64_t seed_base);
/* swap-delete of the equivalent single index (brute_force.h:196-224): its last row moves into the hole */
int VecSimGpu_ShardedDeleteVector(VecSimShardedIndex *index, size_t label);
size_t VecSimGpu_ShardedIndexSize(VecSimShardedIndex *index);
int VecSimGpu_ShardedTopKQueryBatch(VecSimShardedIndex *index, const void *queryBlobs, size_t nq, size_t queryStride, size_t k,
                                    VecSimQueryParams *queryParams, VecSimQueryReply_Order order, VecSimQueryReply **replies);
int VecSimGpu_ShardedTopKQueryBatchArrays(VecSimShardedIndex *index, const void *queryBlobs, size_t nq, size_t queryStride,
                                          size_t k, VecSimQueryParams *queryParams, VecSimQueryReply_Order order,
                                          int64_t *labels, double *scores, int *codes);
/* Several reader threads per process: `seq` is the batch's position in the stream of batches EVERY process answers (0, 1, 2 ...,
 * no gaps; each number used once on each process).  The scans of concurrent batches overlap on the shard's reader lanes, the
 * exchanges are issued in seq order on every process whatever the threads' relative speed, so batch i's exchange and merge run
 * under batch i+1's scan.  The plain entry points above are the one-reader form (call order is the order). */
int VecSimGpu_ShardedTopKQueryBatchArraysSeq(VecSimShardedIndex *index, const void *queryBlobs, size_t nq, size_t queryStride,
                                             size_t k, VecSimQueryParams *queryParams, VecSimQueryReply_Order order,
                                             int64_t *labels, double *scores, int *codes, uint64_t seq);
/* wall time per phase since the last reset, ms: {shard scans, waiting for the exchange turn, exchange, merge + replies},
 * then {batches answered, bytes this process contributed per exchange, summed} */
void VecSimGpu_ShardedGetStats(VecSimShardedIndex *index, double out[6]);
void VecSimGpu_ShardedResetStats(VecSimShardedIndex *index);
/* Starts a new stream of numbered batches at 0 (every process, with no batch in flight).  A ...Seq call whose number has
 * already been answered returns -1 instead of waiting for ever. */
void VecSimGpu_ShardedResetSeq(VecSimShardedIndex *index);
/* the Flat index of a shard held by this process (stats, options); NULL for shards of other processes */
VecSimIndex *VecSimGpu_ShardedLocalIndex(VecSimShardedIndex *index, int shard);
/* Gives up on the other processes: exchanges in flight on this process return an error, later queries are refused (RCCL:
 * ncclCommAbort, which also lets the peers' collectives end).  The library does this itself when an exchange fails or the
 * peers do not arrive within $VECSIM_GPU_EXCHANGE_TIMEOUT_MS (default 120 000 ms); the call is for a caller that learns of a
 * dead peer some other way. */
void VecSimGpu_ShardedAbort(VecSimShardedIndex *index);
/* SPMD check of the transport: every process all-gathers `bytes` rank-stamped bytes, checks every rank's slice and agrees on the
 * verdict: 0 on every process when the exchange moves bytes correctly, -1 on every process otherwise (or after an error /
 * $VECSIM_GPU_EXCHANGE_TIMEOUT_MS: the communicator is then aborted).  How bench.py decides between the two buffer forms. */
int VecSimGpu_ShardedExchangeSelfTest(VecSimShardedIndex *index, size_t bytes);
/* "rccl-staged" | "rccl-mapped" (csrc/vsgpu_comm.hip, $VECSIM_GPU_EXCHANGE) | "transport" (caller's callbacks) | "local" */
const char *VecSimGpu_ShardedExchangeMode(VecSimShardedIndex *index);
int VecSimGpu_ShardedWorld(VecSimShardedIndex *index);
int VecSimGpu_ShardedRank(VecSimShardedIndex *index); /* -1: all shards live in this process */

/* n new vectors at once; labels[i] must not exist yet (returns the number added, -1 on error) */
long VecSimIndex_AddVectorsBulk(VecSimIndex *index, const void *blobs, const size_t *labels, size_t n);

/* append n synthetic fp32 rows generated on the device (labels = internal ids); element j of row i
 * is synth(seed, i*dim + j) in U[-1,1), reproducible on the host (oracle/vso.c:vso_synth_f32) */
long VecSimIndex_AddSyntheticVectors(VecSimIndex *index, size_t n, uint64_t seed);

/* HNSW tooling (tests, benchmarks): the graph the host index built, copied into caller arrays.
 * info = {n, M, M0, entry (0xFFFFFFFF none), max_level (as uint32, 0xFFFFFFFF none), upper_words}.
 * Arrays: links0 [n][M0] u32, cnt0 [n] u16, upper_off [n] u32, upper [upper_words] u32 (blocks of 1+M:
 * count, links), deleted [n] u8, labels [n] u64.  Returns 0, -1 if the index is not an HNSW index. */
int VecSimGpu_HnswGraphInfo(VecSimIndex *index, uint64_t info[6]);
int VecSimGpu_HnswGraphCopy(VecSimIndex *index, uint32_t *links0, uint16_t *cnt0, uint32_t *upper_off, uint32_t *upper,
                            uint8_t *deleted, uint64_t *labels);
/* Per-node top level (n bytes; NULL: none wanted).  Returns 1 when single VecSimIndex_AddVector calls on this index follow the
 * reference's insert path with the tier's own distances (hnsw.h:1567-1610, 889-963; $VECSIM_GPU_HNSW_BUILD = reference | fast,
 * default reference), 0 when the fast builder runs, -1 when `index` is not an HNSW index. */
int VecSimGpu_HnswLevels(VecSimIndex *index, uint8_t *levels);
/* Test hook, needs no GPU: the distance of two STORED blobs as the reference-order HNSW insert path evaluates it on the host
 * (the lane program of (type, metric, tier, dim) walked by csrc/host/host_lane_eval.h).  tier: 0 AVX512, 2 AVX512_BF16, -1 the host's.
 * NaN where the tier has no host walker (AVX512-FP16 half accumulators). */
double VecSimGpu_HostLaneDistance(int type, int metric, int tier, size_t dim, const void *a, const void *b);
uint64_t VecSimGpu_HnswLastDistanceEvals(VecSimIndex *index);

/* ---- SQ8 storage: scalar-quantised 8-bit rows (uint8 codes + FP32 metadata) scored against FP32 queries.
 * The reference defines the layouts (src/VecSim/types/sq8.h:19-62), the quantising preprocessor
 * (spaces/computer/preprocessors.h:259-649) and the distance kernels (spaces/IP/IP.cpp:34-183, spaces/L2/L2.cpp:30-45,
 * 185-201 and the AVX-512 twins chosen by L2_space.cpp:41-107, 518-571 / IP_space.cpp:41-176), but none of its RAM index
 * factories selects them, so the constructor is an extension.  A FlatSQ8 index takes fp32 vectors (params->type =
 * VecSimType_FLOAT32), or fp16 vectors (VecSimType_FLOAT16: QuantPreprocessor<float16>, scores = SQ8_FP16_*,
 * IP.cpp:82-144 / IP_AVX512F_SQ8_FP16.h), through the ordinary VecSimIndex_AddVector / _TopKQuery / _RangeQuery / batch-iterator calls:
 * rows are stored as SQ8 blobs (Cosine: normalised first), queries get y_sum / y_sum_squares appended, every score is the
 * reference's asymmetric SQ8_FP32 distance in its AVX-512 tier order (scalar tier below dim 8), evaluated on the GPU.
 * VecSimIndex_GetDistanceFrom_Unsafe takes the fp32 vector as given (caller-normalised for Cosine).
 * Quarter-size rows: four times the distances per byte of HBM traffic of an fp32 index. */
VecSimIndex *VecSimGpu_NewFlatSQ8(const BFParams *params, void *logCtx);
/* symmetric SQ8_SQ8 distance (IP.cpp:146-183, L2.cpp:185-201) between the stored vectors of two labels; NaN if unknown */
double VecSimGpu_SQ8_StoredDistance(VecSimIndex *index, size_t label_a, size_t label_b);
/* the preprocessor on its own (host, no GPU): storage blob = dim + 12 bytes (16 for L2), query blob = dim + 1 (2) floats */
size_t VecSimGpu_SQ8_StorageBlobSize(size_t dim, VecSimMetric metric);
size_t VecSimGpu_SQ8_QueryBlobSize(size_t dim, VecSimMetric metric);
void VecSimGpu_SQ8_Quantize(const float *vector, size_t dim, VecSimMetric metric, void *storage_blob);
void VecSimGpu_SQ8_QueryBlob(const float *vector, size_t dim, VecSimMetric metric, float *query_blob);
/* Mean-centred SQ8 (QuantPreprocessor<..., WithNorm = true>, preprocessors.h:484-495, 574-640, + DistanceCalculatorWithNorm,
 * spaces/computer/calculator.h:126-232; metric L2 or IP): rows are SQ8 blobs of x - mean (IP rows carry x_mean_ip in a
 * fourth FP32 slot), L2 queries are centred, IP queries stay raw and carry y_mean_ip; every score is the calculator's
 * calcDistanceForQuery (base kernel, minus y_mean_ip for IP), VecSimGpu_SQ8_StoredDistance its calcDistance
 * (base - x_mean_ip - y_mean_ip + mean_sum_squares for IP).  mean: dim FP32 values, copied. */
VecSimIndex *VecSimGpu_NewFlatSQ8Centered(const BFParams *params, const float *mean, float mean_sum_squares, void *logCtx);
size_t VecSimGpu_SQ8_StorageBlobSizeCentered(size_t dim, VecSimMetric metric);   /* dim + 16 bytes */
size_t VecSimGpu_SQ8_QueryBlobSizeCentered(size_t dim, VecSimMetric metric);     /* (dim + 2) floats */
void VecSimGpu_SQ8_QuantizeCentered(const float *vector, const float *mean, size_t dim, VecSimMetric metric, void *storage_blob);
void VecSimGpu_SQ8_QueryBlobCentered(const float *vector, const float *mean, size_t dim, VecSimMetric metric, float *query_blob);

/* Stored (preprocessed) blobs of a label in internal-id order -- what the reference's Python `get_vector`
 * reads through getStoredVectorDataByLabel (bindings.cpp:201-214).  *blob_bytes receives the stored blob size;
 * returns the number of vectors written (0: unknown label, -1: cap_bytes too small / device error); with
 * out == NULL only *blob_bytes is filled. */
long VecSimGpu_GetStoredVectors(VecSimIndex *index, size_t label, void *out, size_t cap_bytes, size_t *blob_bytes);
/* Flat and HNSW indexes: the stored blobs of internal ids [first_id, first_id + n) in one go (n, or -1); measurement / test hook
 * (HNSW: the graph's own node order, VecSimGpu_HnswGraphCopy's -- deletes renumber it when dead nodes are removed) */
long VecSimGpu_ReadStoredRows(VecSimIndex *index, size_t first_id, size_t n, void *out, size_t cap_bytes);

/* device selection for indexes created afterwards on this thread/process (default: $VECSIM_GPU_DEVICE or 0) */
int VecSimGpu_SetDevice(int device);
int VecSimGpu_DeviceCount(void);
/* waits for everything this process has queued on the device it uses (VECSIM_GPU_DEVICE / VecSimGpu_SetDevice): the query calls
 * are synchronous, so this only matters to timing brackets (bench.py) */
int VecSimGpu_DeviceSynchronize(void);
const char *VecSimGpu_LastError(void);
/* Which reference ISA tier's summation order a new index reproduces on this host: "AVX512" | "AVX512_BF16" | "AVX512_FP16" | "SCALAR".
 * Chosen like the reference chooses its kernels -- from the host CPU's features at run time (spaces.h:68-78,
 * IP_space.cpp:554-615, L2_space.cpp:185-241): avx512f -> the AVX-512 kernels' order, avx512_bf16 && avx512vl on top ->
 * vdpbf16ps for bf16 IP / Cosine.  "AVX512_FP16" (half-precision accumulators for fp16 rows of dim >= 32, what a reference built by
 * gcc >= 12 runs on an avx512_fp16 host) is never chosen from CPUID: $VECSIM_GPU_TIER=avx512_fp16 opts in (unpinned order).  A host without AVX-512 also gets "AVX512" (its reference build would run AVX2 / SSE
 * kernels, whose orders are not restated; one line on stderr says so).  $VECSIM_GPU_TIER = avx512 | avx512_bf16 | avx512_fp16 |
 * scalar overrides.  VecSimGpu_IndexTier: the tier an existing index answers in (VecSimIndex_DebugInfoIterator carries exactly the
 * reference's fields). */
const char *VecSimGpu_HostTier(void);
/* Feature names (comma separated) this host lacks for ITS OWN reference build to run the kernel order restated for `type`
 * ("" = none: the reference here runs exactly the restated order; "avx512bw,avx512vbmi2" = its bf16 chooser would fall to a
 * lower tier, L2_space.cpp:332-337).  Index creation prints the same once per type on stderr.  Valid until the thread's next call. */
const char *VecSimGpu_HostTierNote(VecSimType type);
const char *VecSimGpu_IndexTier(VecSimIndex *index);
/* The HIP device ordinal the index's rows live on (-1: none); a multi-rank job prints it per rank so that a first run on an
 * 8-GPU node shows at a glance that every rank drove its own device. */
int VecSimGpu_IndexDevice(VecSimIndex *index);

/* HIP-event timing of the dominant scan kernel since the last reset (bench.py roofline leg) */
typedef struct {
    double scan_ms;
    uint64_t scan_launches;
    uint64_t scan_rows;
    uint64_t scan_bytes;
    double other_ms;
    uint64_t candidates;
    uint64_t fallbacks;
    char scan_kernel[64];
    uint64_t retries;   /* offset 120.  LAYOUT HISTORY: rounds 1-2 ended at scan_kernel (120 bytes); round 3 put `retries` in FRONT of
                         * scan_kernel, round 4 moved it here -- a caller compiled against the round-3 header alone must be rebuilt.
                         * From here on fields are only appended; tests/test_abi.py pins every offset and the size (128). */
} VecSimGpuStats;
void VecSimGpu_ResetStats(VecSimIndex *index);
void VecSimGpu_GetStats(VecSimIndex *index, VecSimGpuStats *out);
int VecSimGpu_SetOption(VecSimIndex *index, const char *name, long value);

#ifdef __cplusplus
}
#endif
