"""ctypes view of libvecsim_amd.so (the VecSim C API of include/VecSim/*.h).

This is the reference-side binding a Python caller uses instead of the pybind11 module: same C
entry points, plain pointers and sizes.  No torch, no numpy magic -- arrays are passed by address.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# VECSIM_AMD_LIBDIR: another build of the two libraries (the TUNING=1 build with its extra kernel variants lives in
# vectorsimilarity_amd/tuning/: make -C vectorsimilarity_amd/csrc TUNING=1 OUT=../tuning OBJ=build_tuning)
_LIBDIR = os.environ.get("VECSIM_AMD_LIBDIR", _PKG)
LIB_PATH = os.path.join(_LIBDIR, "libvecsim_amd.so")
GPU_LIB_PATH = os.path.join(_LIBDIR, "libvsgpu.so")

# ---- enums (vec_sim_common.h) ----
VecSimType_FLOAT32, VecSimType_FLOAT64, VecSimType_BFLOAT16, VecSimType_FLOAT16, \
    VecSimType_INT8, VecSimType_UINT8, VecSimType_INT32, VecSimType_INT64 = range(8)
VecSimAlgo_BF, VecSimAlgo_HNSWLIB, VecSimAlgo_TIERED, VecSimAlgo_SVS = range(4)
VecSimMetric_L2, VecSimMetric_IP, VecSimMetric_Cosine = range(3)
BY_SCORE, BY_ID, BY_SCORE_THEN_ID = range(3)
VecSim_QueryReply_OK, VecSim_QueryReply_TimedOut = range(2)
QUERY_TYPE_NONE, QUERY_TYPE_KNN, QUERY_TYPE_HYBRID, QUERY_TYPE_RANGE = range(4)


class BFParams(C.Structure):
    _fields_ = [("type", C.c_int), ("dim", C.c_size_t), ("metric", C.c_int), ("multi", C.c_bool),
                ("initialCapacity", C.c_size_t), ("blockSize", C.c_size_t)]


class HNSWParams(C.Structure):
    _fields_ = [("type", C.c_int), ("dim", C.c_size_t), ("metric", C.c_int), ("multi", C.c_bool),
                ("initialCapacity", C.c_size_t), ("blockSize", C.c_size_t), ("M", C.c_size_t),
                ("efConstruction", C.c_size_t), ("efRuntime", C.c_size_t), ("epsilon", C.c_double)]


class SVSParams(C.Structure):
    _fields_ = [("type", C.c_int), ("dim", C.c_size_t), ("metric", C.c_int), ("multi", C.c_bool),
                ("blockSize", C.c_size_t), ("quantBits", C.c_int), ("alpha", C.c_float),
                ("graph_max_degree", C.c_size_t), ("construction_window_size", C.c_size_t),
                ("max_candidate_pool_size", C.c_size_t), ("prune_to", C.c_size_t),
                ("use_search_history", C.c_int), ("num_threads", C.c_size_t),
                ("search_window_size", C.c_size_t), ("search_buffer_capacity", C.c_size_t),
                ("leanvec_dim", C.c_size_t), ("epsilon", C.c_double)]


class _TieredSpecific(C.Union):
    _fields_ = [("swapJobThreshold", C.c_size_t), ("svs", C.c_size_t * 3), ("disk", C.c_char)]


class TieredIndexParams(C.Structure):
    _fields_ = [("jobQueue", C.c_void_p), ("jobQueueCtx", C.c_void_p), ("submitCb", C.c_void_p),
                ("flatBufferLimit", C.c_size_t), ("primaryIndexParams", C.c_void_p),
                ("specificParams", _TieredSpecific)]


class AlgoParams(C.Union):
    _fields_ = [("hnswParams", HNSWParams), ("bfParams", BFParams),
                ("tieredParams", TieredIndexParams), ("svsParams", SVSParams)]


class VecSimParams(C.Structure):
    _fields_ = [("algo", C.c_int), ("algoParams", AlgoParams), ("logCtx", C.c_void_p)]


class HNSWRuntimeParams(C.Structure):
    _fields_ = [("efRuntime", C.c_size_t), ("epsilon", C.c_double)]


class _RuntimeUnion(C.Union):
    _fields_ = [("hnswRuntimeParams", HNSWRuntimeParams), ("hnswDisk", C.c_size_t * 3), ("svs", C.c_size_t * 4)]


class VecSimQueryParams(C.Structure):
    _anonymous_ = ("u",)
    _fields_ = [("u", _RuntimeUnion), ("batchSize", C.c_size_t), ("searchMode", C.c_int),
                ("timeoutCtx", C.c_void_p)]


class VecSimRawParam(C.Structure):
    _fields_ = [("name", C.c_char_p), ("nameLen", C.c_size_t), ("value", C.c_char_p),
                ("valLen", C.c_size_t)]


class VecSimIndexBasicInfo(C.Structure):
    _fields_ = [("algo", C.c_int), ("metric", C.c_int), ("type", C.c_int), ("isMulti", C.c_bool),
                ("isTiered", C.c_bool), ("isDisk", C.c_bool), ("blockSize", C.c_size_t),
                ("dim", C.c_size_t)]


class VecSimGpuStats(C.Structure):
    _fields_ = [("scan_ms", C.c_double), ("scan_launches", C.c_uint64), ("scan_rows", C.c_uint64),
                ("scan_bytes", C.c_uint64), ("other_ms", C.c_double), ("candidates", C.c_uint64),
                ("fallbacks", C.c_uint64), ("scan_kernel", C.c_char * 64), ("retries", C.c_uint64)]


TIMEOUT_CB = C.CFUNCTYPE(C.c_int, C.c_void_p)
# transport / external-shard callbacks of the sharded index (vec_sim_gpu.h)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
BROADCAST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)
SHARD_ADD_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
SHARD_CAND_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
LOG_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_char_p)

class VecSimIndexStatsInfo(C.Structure):  # vec_sim_common.h:276-281
    _fields_ = [("memory", C.c_size_t), ("numberOfMarkedDeleted", C.c_size_t), ("directHNSWInsertions", C.c_size_t),
                ("flatBufferSize", C.c_size_t)]


class FieldValue(C.Union):
    _fields_ = [("floatingPointValue", C.c_double), ("integerValue", C.c_int64), ("uintegerValue", C.c_uint64),
                ("stringValue", C.c_char_p), ("iteratorValue", C.c_void_p)]


class VecSim_InfoField(C.Structure):  # include/VecSim/info_iterator.h
    _fields_ = [("fieldName", C.c_char_p), ("fieldType", C.c_int), ("fieldValue", FieldValue)]


# every symbol include/VecSim/{vec_sim,query_results,info_iterator,vec_sim_gpu}.h declares
EXPORTS = [
    "VecSimIndex_New", "VecSimIndex_Free", "VecSimIndex_EstimateInitialSize",
    "VecSimIndex_EstimateElementSize", "VecSimIndex_AddVector", "VecSimIndex_DeleteVector",
    "VecSimIndex_IndexSize", "VecSimIndex_TopKQuery", "VecSimIndex_RangeQuery",
    "VecSimIndex_GetDistanceFrom_Unsafe", "VecSimBatchIterator_New", "VecSimIndex_PreferAdHocSearch",
    "VecSimIndex_ResolveParams", "VecSimIndex_AdhocBfCtx_New", "VecSimIndex_AdhocBfCtx_Free",
    "VecSimIndex_AdhocBfCtx_GetDistanceFrom", "VecSimIndex_AdhocBfCtx_GetExactDistances",
    "VecSim_Normalize", "VecSimParams_GetQueryBlobSize", "VecSimIndex_DebugInfo",
    "VecSimIndex_BasicInfo", "VecSimIndex_StatsInfo", "VecSimIndex_DebugInfoIterator",
    "VecSimTieredIndex_GC", "VecSimTieredIndex_AcquireSharedLocks",
    "VecSimTieredIndex_ReleaseSharedLocks", "VecSim_SetMemoryFunctions",
    "VecSim_SetTimeoutCallbackFunction", "VecSim_SetLogCallbackFunction", "VecSim_SetTestLogContext",
    "VecSim_SetWriteMode", "VecSim_UpdateThreadPoolSize", "VecSim_GetSharedMemory",
    "VecSimQueryResult_GetId", "VecSimQueryResult_GetScore", "VecSimQueryReply_Len",
    "VecSimQueryReply_GetCode", "VecSimQueryReply_Free", "VecSimQueryReply_GetIterator",
    "VecSimQueryReply_IteratorNext", "VecSimQueryReply_IteratorHasNext",
    "VecSimQueryReply_IteratorReset", "VecSimQueryReply_IteratorFree", "VecSimBatchIterator_Next",
    "VecSimBatchIterator_HasNext", "VecSimBatchIterator_Free", "VecSimBatchIterator_Reset",
    "VecSimIndex_TopKQueryBatch", "VecSimIndex_TopKQueryBatchArrays", "VecSimIndex_TopKCandidatesBatch", "VecSimGpu_MergeTopK",
    "VecSimIndex_AddVectorsBulk", "VecSimIndex_AddSyntheticVectors",
    "VecSimDebug_GetElementNeighborsInHNSWGraph", "VecSimDebug_ReleaseElementNeighborsInHNSWGraph", "VecSimGpu_HnswGraphInfo", "VecSimGpu_HnswGraphCopy", "VecSimGpu_HnswLevels", "VecSimGpu_HostLaneDistance", "VecSimGpu_HnswLastDistanceEvals", "VecSimGpu_GetStoredVectors", "VecSimGpu_ReadStoredRows",
    "VecSimGpu_NewFlatSQ8", "VecSimGpu_SQ8_StoredDistance", "VecSimGpu_SQ8_StorageBlobSize", "VecSimGpu_SQ8_QueryBlobSize",
    "VecSimGpu_SQ8_Quantize", "VecSimGpu_SQ8_QueryBlob", "VecSimGpu_NewFlatSQ8Centered", "VecSimGpu_SQ8_StorageBlobSizeCentered",
    "VecSimGpu_SQ8_QueryBlobSizeCentered", "VecSimGpu_SQ8_QuantizeCentered", "VecSimGpu_SQ8_QueryBlobCentered",
    "VecSimGpu_SetDevice", "VecSimGpu_DeviceCount", "VecSimGpu_DeviceSynchronize", "VecSimGpu_LastError", "VecSimGpu_HostTier", "VecSimGpu_HostTierNote", "VecSimGpu_IndexTier", "VecSimGpu_IndexDevice", "VecSimGpu_ResetStats",
    "VecSimGpu_GetStats", "VecSimGpu_SetOption",
    "VecSimGpu_ShardedGetUniqueId", "VecSimGpu_ShardedNew", "VecSimGpu_ShardedNewWithTransport",
    "VecSimGpu_ShardedNewExternal", "VecSimGpu_ShardedNewLocal", "VecSimGpu_ShardedFree", "VecSimGpu_ShardedAddVector",
    "VecSimGpu_ShardedAddVectorsBulk", "VecSimGpu_ShardedAddSyntheticLocal", "VecSimGpu_ShardedDeleteVector",
    "VecSimGpu_ShardedIndexSize", "VecSimGpu_ShardedTopKQueryBatch", "VecSimGpu_ShardedTopKQueryBatchArrays",
    "VecSimGpu_ShardedTopKQueryBatchArraysSeq", "VecSimGpu_ShardedGetStats", "VecSimGpu_ShardedResetStats", "VecSimGpu_ShardedResetSeq",
    "VecSimGpu_ShardedLocalIndex", "VecSimGpu_ShardedAbort", "VecSimGpu_ShardedExchangeSelfTest", "VecSimGpu_ShardedExchangeMode", "VecSimGpu_ShardedWorld", "VecSimGpu_ShardedRank",
    "VecSimDebugInfoIterator_NumberOfFields", "VecSimDebugInfoIterator_HasNextField",
    "VecSimDebugInfoIterator_NextField", "VecSimDebugInfoIterator_Free",
]
GPU_EXPORTS = [
    "vsgpu_device_count", "vsgpu_device_synchronize", "vsgpu_last_error", "vsgpu_ctx_create", "vsgpu_ctx_destroy",
    "vsgpu_ctx_device", "vsgpu_ctx_sync", "vsgpu_table_create", "vsgpu_table_destroy",
    "vsgpu_table_size", "vsgpu_table_bytes", "vsgpu_table_append", "vsgpu_table_write",
    "vsgpu_table_move", "vsgpu_table_truncate", "vsgpu_table_read", "vsgpu_table_read_range", "vsgpu_table_append_synthetic",
    "vsgpu_table_view_create", "vsgpu_table_view_sync",
    "vsgpu_graph_create", "vsgpu_graph_destroy", "vsgpu_graph_set_multi", "vsgpu_graph_view_create", "vsgpu_graph_upload", "vsgpu_graph_search", "vsgpu_graph_range",
    "vsgpu_scorebuf_create", "vsgpu_scorebuf_destroy", "vsgpu_scorebuf_rows", "vsgpu_scorebuf_next", "vsgpu_scorebuf_retire",
    "vsgpu_scorebuf_read",
    "vsgpu_topk", "vsgpu_range", "vsgpu_scores", "vsgpu_scores_of", "vsgpu_sq8_pair_scores", "vsgpu_table_set_sq8_mean_sum_squares", "vsgpu_table_set_sq8_block_bounds", "vsgpu_stats_reset",
    "vsgpu_stats_get", "vsgpu_set_option", "vsgpu_set_poll",
    "vsgpu_comm_unique_id", "vsgpu_comm_create", "vsgpu_comm_destroy", "vsgpu_comm_rank", "vsgpu_comm_world",
    "vsgpu_comm_allgather", "vsgpu_comm_broadcast", "vsgpu_comm_abort", "vsgpu_comm_staged",
]

_lib = None


def load():
    """Load libvecsim_amd.so (which links libvsgpu.so).  Raises if the native build is missing:
    there is no pure-Python or CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not (os.path.exists(LIB_PATH) and os.path.exists(GPU_LIB_PATH)):
        raise ImportError(
            "vectorsimilarity_amd: native libraries not built (%s). Run __graft_entry__.build() "
            "or `make -C vectorsimilarity_amd/csrc`; there is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, sz, dbl, i = C.c_void_p, C.c_size_t, C.c_double, C.c_int
    L.VecSimIndex_New.restype = vp
    L.VecSimIndex_New.argtypes = [C.POINTER(VecSimParams)]
    L.VecSimIndex_Free.restype = None
    L.VecSimIndex_Free.argtypes = [vp]
    L.VecSimIndex_AddVector.restype = i
    L.VecSimIndex_AddVector.argtypes = [vp, vp, sz]
    L.VecSimIndex_DeleteVector.restype = i
    L.VecSimIndex_DeleteVector.argtypes = [vp, sz]
    L.VecSimIndex_IndexSize.restype = sz
    L.VecSimIndex_IndexSize.argtypes = [vp]
    L.VecSimIndex_TopKQuery.restype = vp
    L.VecSimIndex_TopKQuery.argtypes = [vp, vp, sz, C.POINTER(VecSimQueryParams), i]
    L.VecSimIndex_RangeQuery.restype = vp
    L.VecSimIndex_RangeQuery.argtypes = [vp, vp, dbl, C.POINTER(VecSimQueryParams), i]
    L.VecSimIndex_GetDistanceFrom_Unsafe.restype = dbl
    L.VecSimIndex_GetDistanceFrom_Unsafe.argtypes = [vp, sz, vp]
    L.VecSimBatchIterator_New.restype = vp
    L.VecSimBatchIterator_New.argtypes = [vp, vp, C.POINTER(VecSimQueryParams)]
    L.VecSimIndex_PreferAdHocSearch.restype = C.c_bool
    L.VecSimIndex_PreferAdHocSearch.argtypes = [vp, sz, sz, C.c_bool]
    L.VecSimIndex_ResolveParams.restype = i
    L.VecSimIndex_ResolveParams.argtypes = [vp, C.POINTER(VecSimRawParam), i,
                                            C.POINTER(VecSimQueryParams), i]
    L.VecSim_Normalize.restype = None
    L.VecSim_Normalize.argtypes = [vp, sz, i]
    L.VecSimParams_GetQueryBlobSize.restype = sz
    L.VecSimParams_GetQueryBlobSize.argtypes = [i, sz, i]
    L.VecSimIndex_DebugInfoIterator.restype = vp
    L.VecSimIndex_DebugInfoIterator.argtypes = [vp]
    L.VecSimDebugInfoIterator_NumberOfFields.restype = C.c_size_t
    L.VecSimDebugInfoIterator_NumberOfFields.argtypes = [vp]
    L.VecSimDebugInfoIterator_HasNextField.restype = C.c_bool
    L.VecSimDebugInfoIterator_HasNextField.argtypes = [vp]
    L.VecSimDebugInfoIterator_NextField.restype = C.POINTER(VecSim_InfoField)
    L.VecSimDebugInfoIterator_NextField.argtypes = [vp]
    L.VecSimDebugInfoIterator_Free.restype = None
    L.VecSimDebugInfoIterator_Free.argtypes = [vp]
    L.VecSimGpu_NewFlatSQ8.restype = vp
    L.VecSimGpu_NewFlatSQ8.argtypes = [vp, vp]
    L.VecSimGpu_SQ8_StoredDistance.restype = C.c_double
    L.VecSimGpu_SQ8_StoredDistance.argtypes = [vp, C.c_size_t, C.c_size_t]
    L.VecSimGpu_SQ8_StorageBlobSize.restype = C.c_size_t
    L.VecSimGpu_SQ8_StorageBlobSize.argtypes = [C.c_size_t, C.c_int]
    L.VecSimGpu_SQ8_QueryBlobSize.restype = C.c_size_t
    L.VecSimGpu_SQ8_QueryBlobSize.argtypes = [C.c_size_t, C.c_int]
    L.VecSimGpu_SQ8_Quantize.restype = None
    L.VecSimGpu_SQ8_Quantize.argtypes = [vp, C.c_size_t, C.c_int, vp]
    L.VecSimGpu_SQ8_QueryBlob.restype = None
    L.VecSimGpu_SQ8_QueryBlob.argtypes = [vp, C.c_size_t, C.c_int, vp]
    L.VecSimGpu_NewFlatSQ8Centered.restype = vp
    L.VecSimGpu_NewFlatSQ8Centered.argtypes = [vp, vp, C.c_float, vp]
    L.VecSimGpu_SQ8_StorageBlobSizeCentered.restype = C.c_size_t
    L.VecSimGpu_SQ8_StorageBlobSizeCentered.argtypes = [C.c_size_t, C.c_int]
    L.VecSimGpu_SQ8_QueryBlobSizeCentered.restype = C.c_size_t
    L.VecSimGpu_SQ8_QueryBlobSizeCentered.argtypes = [C.c_size_t, C.c_int]
    L.VecSimGpu_SQ8_QuantizeCentered.restype = None
    L.VecSimGpu_SQ8_QuantizeCentered.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
    L.VecSimGpu_SQ8_QueryBlobCentered.restype = None
    L.VecSimGpu_SQ8_QueryBlobCentered.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
    L.VecSimGpu_GetStoredVectors.restype = C.c_long
    L.VecSimGpu_GetStoredVectors.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.VecSimGpu_ReadStoredRows.restype = C.c_long
    L.VecSimGpu_ReadStoredRows.argtypes = [vp, C.c_size_t, C.c_size_t, vp, C.c_size_t]
    L.VecSimIndex_StatsInfo.restype = VecSimIndexStatsInfo
    L.VecSimIndex_StatsInfo.argtypes = [vp]
    L.VecSimIndex_BasicInfo.restype = VecSimIndexBasicInfo
    L.VecSimIndex_BasicInfo.argtypes = [vp]
    L.VecSim_SetTimeoutCallbackFunction.restype = None
    L.VecSim_SetTimeoutCallbackFunction.argtypes = [vp]
    L.VecSim_SetLogCallbackFunction.restype = None
    L.VecSim_SetLogCallbackFunction.argtypes = [vp]
    L.VecSimQueryResult_GetId.restype = C.c_int64
    L.VecSimQueryResult_GetId.argtypes = [vp]
    L.VecSimQueryResult_GetScore.restype = dbl
    L.VecSimQueryResult_GetScore.argtypes = [vp]
    L.VecSimQueryReply_Len.restype = sz
    L.VecSimQueryReply_Len.argtypes = [vp]
    L.VecSimQueryReply_GetCode.restype = i
    L.VecSimQueryReply_GetCode.argtypes = [vp]
    L.VecSimQueryReply_Free.restype = None
    L.VecSimQueryReply_Free.argtypes = [vp]
    L.VecSimQueryReply_GetIterator.restype = vp
    L.VecSimQueryReply_GetIterator.argtypes = [vp]
    L.VecSimQueryReply_IteratorNext.restype = vp
    L.VecSimQueryReply_IteratorNext.argtypes = [vp]
    L.VecSimQueryReply_IteratorHasNext.restype = C.c_bool
    L.VecSimQueryReply_IteratorHasNext.argtypes = [vp]
    L.VecSimQueryReply_IteratorReset.restype = None
    L.VecSimQueryReply_IteratorReset.argtypes = [vp]
    L.VecSimQueryReply_IteratorFree.restype = None
    L.VecSimQueryReply_IteratorFree.argtypes = [vp]
    L.VecSimBatchIterator_Next.restype = vp
    L.VecSimBatchIterator_Next.argtypes = [vp, sz, i]
    L.VecSimBatchIterator_HasNext.restype = C.c_bool
    L.VecSimBatchIterator_HasNext.argtypes = [vp]
    L.VecSimBatchIterator_Free.restype = None
    L.VecSimBatchIterator_Free.argtypes = [vp]
    L.VecSimBatchIterator_Reset.restype = None
    L.VecSimBatchIterator_Reset.argtypes = [vp]
    L.VecSimIndex_TopKQueryBatch.restype = i
    L.VecSimIndex_TopKQueryBatch.argtypes = [vp, vp, sz, sz, sz, C.POINTER(VecSimQueryParams), i,
                                             C.POINTER(vp)]
    L.VecSimIndex_TopKQueryBatchArrays.restype = i
    L.VecSimIndex_TopKQueryBatchArrays.argtypes = [vp, vp, sz, sz, sz, C.POINTER(VecSimQueryParams), i, vp, vp, vp]
    L.VecSimIndex_TopKCandidatesBatch.restype = i
    L.VecSimIndex_TopKCandidatesBatch.argtypes = [vp, vp, sz, sz, sz, sz, vp, vp, vp, vp]
    L.VecSimGpu_MergeTopK.restype = i
    L.VecSimGpu_MergeTopK.argtypes = [sz, sz, sz, vp, vp, vp, vp, sz, vp, vp]
    L.VecSimIndex_AddVectorsBulk.restype = C.c_long
    L.VecSimIndex_AddVectorsBulk.argtypes = [vp, vp, vp, sz]
    L.VecSimIndex_AddSyntheticVectors.restype = C.c_long
    L.VecSimIndex_AddSyntheticVectors.argtypes = [vp, sz, C.c_uint64]
    L.VecSimGpu_HnswGraphInfo.restype = i
    L.VecSimGpu_HnswGraphInfo.argtypes = [vp, vp]
    L.VecSimGpu_HnswGraphCopy.restype = i
    L.VecSimGpu_HnswGraphCopy.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.VecSimGpu_HnswLevels.restype = i
    L.VecSimGpu_HnswLevels.argtypes = [vp, vp]
    L.VecSimGpu_HostLaneDistance.restype = C.c_double
    L.VecSimGpu_HostLaneDistance.argtypes = [i, i, i, C.c_size_t, vp, vp]
    L.VecSimGpu_HnswLastDistanceEvals.restype = C.c_uint64
    L.VecSimGpu_HnswLastDistanceEvals.argtypes = [vp]
    L.VecSimGpu_SetDevice.restype = i
    L.VecSimGpu_SetDevice.argtypes = [i]
    L.VecSimGpu_DeviceCount.restype = i
    L.VecSimGpu_DeviceSynchronize.restype = i
    L.VecSimGpu_DeviceSynchronize.argtypes = []
    L.VecSimGpu_LastError.restype = C.c_char_p
    L.VecSimGpu_HostTier.restype = C.c_char_p
    L.VecSimGpu_HostTierNote.restype = C.c_char_p
    L.VecSimGpu_HostTierNote.argtypes = [C.c_int]
    L.VecSimGpu_IndexTier.restype = C.c_char_p
    L.VecSimGpu_IndexTier.argtypes = [C.c_void_p]
    L.VecSimGpu_IndexDevice.restype = C.c_int
    L.VecSimGpu_IndexDevice.argtypes = [C.c_void_p]
    L.VecSimGpu_ResetStats.restype = None
    L.VecSimGpu_ResetStats.argtypes = [vp]
    L.VecSimGpu_GetStats.restype = None
    L.VecSimGpu_GetStats.argtypes = [vp, C.POINTER(VecSimGpuStats)]
    L.VecSimGpu_SetOption.restype = i
    L.VecSimGpu_SetOption.argtypes = [vp, C.c_char_p, C.c_long]
    L.VecSimGpu_ShardedGetUniqueId.restype = i
    L.VecSimGpu_ShardedGetUniqueId.argtypes = [vp]
    L.VecSimGpu_ShardedNew.restype = vp
    L.VecSimGpu_ShardedNew.argtypes = [C.POINTER(VecSimParams), i, i, i, vp]
    L.VecSimGpu_ShardedNewWithTransport.restype = vp
    L.VecSimGpu_ShardedNewWithTransport.argtypes = [C.POINTER(VecSimParams), i, i, i, ALLGATHER_FN, BROADCAST_FN, vp]
    L.VecSimGpu_ShardedNewExternal.restype = vp
    L.VecSimGpu_ShardedNewExternal.argtypes = [C.POINTER(VecSimParams), i, i, SHARD_ADD_FN, SHARD_CAND_FN, ALLGATHER_FN, vp]
    L.VecSimGpu_ShardedNewLocal.restype = vp
    L.VecSimGpu_ShardedNewLocal.argtypes = [C.POINTER(VecSimParams), i, vp]
    L.VecSimGpu_ShardedFree.restype = None
    L.VecSimGpu_ShardedFree.argtypes = [vp]
    L.VecSimGpu_ShardedAddVector.restype = i
    L.VecSimGpu_ShardedAddVector.argtypes = [vp, vp, sz]
    L.VecSimGpu_ShardedAddVectorsBulk.restype = C.c_long
    L.VecSimGpu_ShardedAddVectorsBulk.argtypes = [vp, vp, vp, sz]
    L.VecSimGpu_ShardedAddSyntheticLocal.restype = C.c_long
    L.VecSimGpu_ShardedAddSyntheticLocal.argtypes = [vp, sz, C.c_uint64]
    L.VecSimGpu_ShardedDeleteVector.restype = i
    L.VecSimGpu_ShardedDeleteVector.argtypes = [vp, sz]
    L.VecSimGpu_ShardedIndexSize.restype = sz
    L.VecSimGpu_ShardedIndexSize.argtypes = [vp]
    L.VecSimGpu_ShardedTopKQueryBatch.restype = i
    L.VecSimGpu_ShardedTopKQueryBatch.argtypes = [vp, vp, sz, sz, sz, C.POINTER(VecSimQueryParams), i, C.POINTER(vp)]
    L.VecSimGpu_ShardedTopKQueryBatchArrays.restype = i
    L.VecSimGpu_ShardedTopKQueryBatchArrays.argtypes = [vp, vp, sz, sz, sz, C.POINTER(VecSimQueryParams), i, vp, vp, vp]
    L.VecSimGpu_ShardedTopKQueryBatchArraysSeq.restype = i
    L.VecSimGpu_ShardedTopKQueryBatchArraysSeq.argtypes = [vp, vp, sz, sz, sz, C.POINTER(VecSimQueryParams), i, vp, vp, vp, C.c_uint64]
    L.VecSimGpu_ShardedGetStats.restype = None
    L.VecSimGpu_ShardedGetStats.argtypes = [vp, C.POINTER(C.c_double)]
    L.VecSimGpu_ShardedResetStats.restype = None
    L.VecSimGpu_ShardedResetStats.argtypes = [vp]
    L.VecSimGpu_ShardedResetSeq.restype = None
    L.VecSimGpu_ShardedResetSeq.argtypes = [vp]
    L.VecSimGpu_ShardedLocalIndex.restype = vp
    L.VecSimGpu_ShardedLocalIndex.argtypes = [vp, i]
    L.VecSimGpu_ShardedAbort.restype = None
    L.VecSimGpu_ShardedAbort.argtypes = [vp]
    L.VecSimGpu_ShardedExchangeSelfTest.restype = i
    L.VecSimGpu_ShardedExchangeSelfTest.argtypes = [vp, C.c_size_t]
    L.VecSimGpu_ShardedExchangeMode.restype = C.c_char_p
    L.VecSimGpu_ShardedExchangeMode.argtypes = [vp]
    L.VecSimGpu_ShardedWorld.restype = i
    L.VecSimGpu_ShardedWorld.argtypes = [vp]
    L.VecSimGpu_ShardedRank.restype = i
    L.VecSimGpu_ShardedRank.argtypes = [vp]
    _lib = L
    return L
