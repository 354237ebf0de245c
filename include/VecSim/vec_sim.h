/*
 * VecSim/vec_sim.h -- the index C API (drop-in boundary).
 *
 * Same symbols and signatures as the reference's src/VecSim/vec_sim.h:28-331.  Behind it, Flat
 * (VecSimAlgo_BF, single- and multi-label) and HNSW (VecSimAlgo_HNSWLIB, single- and multi-label) indexes keep
 * their vector blocks in MI355X HBM and every distance a QUERY needs (top-k, range, batch iterator,
 * GetDistanceFrom) is evaluated by the gfx950 kernels reached through include/vsgpu.h; no query entry point
 * has a CPU distance path and none falls back to one.  The one place host code evaluates distances is HNSW
 * INGEST: the graph an AddVector call leaves depends on the build-time distances (hnsw.h:1567-1610), so the
 * insert path ranks its candidates on the host -- in the index's own tier order (csrc/host/host_lane_eval.h)
 * for single adds, with a fast any-order routine for the parallel bulk build (csrc/host/hnsw_index.cpp).
 * VecSimIndex_New returns NULL for algorithms this build does not construct (tiered, SVS) and when no GPU is
 * visible.
 */
#pragma once
#include <stdlib.h>

#include "query_results.h"
#include "vec_sim_common.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct VecSimIndexInterface VecSimIndex;
#include "info_iterator.h"
typedef struct VecSimAdhocBfCtx VecSimAdhocBfCtx;

/* lifetime (vec_sim.cpp:213-215, 369-373) */
VecSimIndex *VecSimIndex_New(const VecSimParams *params);
void VecSimIndex_Free(VecSimIndex *index);
size_t VecSimIndex_EstimateInitialSize(const VecSimParams *params);
size_t VecSimIndex_EstimateElementSize(const VecSimParams *params);

/* ingest (vec_sim.cpp:221-227): 1 = new label, 0 = overwrite / nothing deleted */
int VecSimIndex_AddVector(VecSimIndex *index, const void *blob, size_t label);
int VecSimIndex_DeleteVector(VecSimIndex *index, size_t label);
size_t VecSimIndex_IndexSize(VecSimIndex *index);

/* queries (vec_sim.cpp:345-367) */
VecSimQueryReply *VecSimIndex_TopKQuery(VecSimIndex *index, const void *queryBlob, size_t k,
                                        VecSimQueryParams *queryParams, VecSimQueryReply_Order);
VecSimQueryReply *VecSimIndex_RangeQuery(VecSimIndex *index, const void *queryBlob, double radius,
                                         VecSimQueryParams *queryParams, VecSimQueryReply_Order);
double VecSimIndex_GetDistanceFrom_Unsafe(VecSimIndex *index, size_t label, const void *blob);
VecSimBatchIterator *VecSimBatchIterator_New(VecSimIndex *index, const void *queryBlob,
                                             VecSimQueryParams *queryParams);
bool VecSimIndex_PreferAdHocSearch(VecSimIndex *index, size_t subsetSize, size_t k, bool initial_check);
VecSimResolveCode VecSimIndex_ResolveParams(VecSimIndex *index, VecSimRawParam *rparams, int paramNum,
                                            VecSimQueryParams *qparams, VecsimQueryType query_type);

/* ad-hoc brute force context: RAM indexes return NULL upstream (vec_sim_interface.h), same here */
VecSimAdhocBfCtx *VecSimIndex_AdhocBfCtx_New(VecSimIndex *index, const void *queryBlob);
void VecSimIndex_AdhocBfCtx_Free(VecSimAdhocBfCtx *ctx);
double VecSimIndex_AdhocBfCtx_GetDistanceFrom(VecSimAdhocBfCtx *ctx, size_t label);
void VecSimIndex_AdhocBfCtx_GetExactDistances(VecSimAdhocBfCtx *ctx, const size_t *labels,
                                              double *distances_out, size_t count);

/* blobs (vec_sim.cpp:238-266) */
void VecSim_Normalize(void *blob, size_t dim, VecSimType type);
size_t VecSimParams_GetQueryBlobSize(VecSimType type, size_t dim, VecSimMetric metric);

/* info */
VecSimIndexDebugInfo VecSimIndex_DebugInfo(VecSimIndex *index);
VecSimIndexBasicInfo VecSimIndex_BasicInfo(VecSimIndex *index);
VecSimIndexStatsInfo VecSimIndex_StatsInfo(VecSimIndex *index);
VecSimDebugInfoIterator *VecSimIndex_DebugInfoIterator(VecSimIndex *index);

/* tiered-only entry points: accepted and ignored for Flat indexes */
void VecSimTieredIndex_GC(VecSimIndex *index);
void VecSimTieredIndex_AcquireSharedLocks(VecSimIndex *index);
void VecSimTieredIndex_ReleaseSharedLocks(VecSimIndex *index);

/* process-wide hooks (vec_sim.h:281-313) */
void VecSim_SetMemoryFunctions(VecSimMemoryFunctions memoryfunctions);
void VecSim_SetTimeoutCallbackFunction(timeoutCallbackFunction callback);
void VecSim_SetLogCallbackFunction(logCallbackFunction callback);
void VecSim_SetTestLogContext(const char *test_name, const char *test_type);
void VecSim_SetWriteMode(VecSimWriteMode mode);
void VecSim_UpdateThreadPoolSize(size_t new_size);
size_t VecSim_GetSharedMemory(void);

#ifdef __cplusplus
}
#endif
