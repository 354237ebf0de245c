"""Python surface with the names and I/O shapes of the reference's pybind11 module `VecSim`
(src/python_bindings/bindings.cpp:660-868), bound over the C API with ctypes.

    p = BFParams(); p.type = VecSimType_FLOAT32; p.dim = 128; p.metric = VecSimMetric_L2
    ix = BFIndex(p); ix.add_vector(v, label); labels, dists = ix.knn_query(q, 10)

`knn_query` returns (labels int64[nq,k], distances float64[nq,k]) padded with -1 like
wrap_results (bindings.cpp:36-72).  A 2-D query array is answered as ONE batched GPU pass
(VecSimIndex_TopKQueryBatch) -- the reference loops single queries (bindings.cpp:182-190).
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import (BY_ID, BY_SCORE, VecSimAlgo_BF, VecSimAlgo_HNSWLIB, VecSimAlgo_SVS,  # noqa: F401
                    VecSimAlgo_TIERED, VecSimMetric_Cosine, VecSimMetric_IP, VecSimMetric_L2,
                    VecSimType_BFLOAT16, VecSimType_FLOAT16, VecSimType_FLOAT32,
                    VecSimType_FLOAT64, VecSimType_INT8, VecSimType_INT32, VecSimType_INT64,
                    VecSimType_UINT8, BFParams, HNSWParams, VecSimParams, VecSimQueryParams)

_NP = {VecSimType_FLOAT32: np.float32, VecSimType_FLOAT64: np.float64,
       VecSimType_BFLOAT16: np.uint16, VecSimType_FLOAT16: np.uint16, VecSimType_INT8: np.int8,
       VecSimType_UINT8: np.uint8}
_ELEM = {VecSimType_FLOAT32: 4, VecSimType_FLOAT64: 8, VecSimType_BFLOAT16: 2,
         VecSimType_FLOAT16: 2, VecSimType_INT8: 1, VecSimType_UINT8: 1}


def _wrap(lib, replies, num_res):
    nq = len(replies)
    labels = np.full((nq, num_res), -1, dtype=np.int64)
    dists = np.full((nq, num_res), -1.0, dtype=np.float64)
    for qi, rep in enumerate(replies):
        it = lib.VecSimQueryReply_GetIterator(rep)
        j = 0
        while lib.VecSimQueryReply_IteratorHasNext(it):
            item = lib.VecSimQueryReply_IteratorNext(it)
            dists[qi, j] = lib.VecSimQueryResult_GetScore(item)
            labels[qi, j] = lib.VecSimQueryResult_GetId(item)
            j += 1
        lib.VecSimQueryReply_IteratorFree(it)
        lib.VecSimQueryReply_Free(rep)
    return labels, dists


class BatchIterator:
    def __init__(self, index, handle):
        self._index = index  # keep the index alive longer than the iterator
        self._h = handle
        self._lib = _capi.load()

    def has_next(self):
        return bool(self._lib.VecSimBatchIterator_HasNext(self._h))

    def get_next_results(self, n_res, order=BY_SCORE):
        rep = self._lib.VecSimBatchIterator_Next(self._h, n_res, order)
        n = self._lib.VecSimQueryReply_Len(rep)
        return _wrap(self._lib, [rep], n)

    def reset(self):
        self._lib.VecSimBatchIterator_Reset(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.VecSimBatchIterator_Free(self._h)
            self._h = None


class VecSimIndex:
    _owned = True

    def __init__(self, params, borrowed_handle=None, sq8=False, sq8_mean=None, sq8_mean_sum_squares=0.0):
        self._lib = _capi.load()
        if borrowed_handle is not None:   # a shard of a sharded index: the owner frees it
            self._h, self._owned = borrowed_handle, False
        elif sq8 and sq8_mean is not None:   # mean-centred SQ8 (params: BFParams)
            mean = np.ascontiguousarray(sq8_mean, dtype=np.float32)
            assert mean.size == params.dim
            self._h = self._lib.VecSimGpu_NewFlatSQ8Centered(C.byref(params), mean.ctypes.data_as(C.c_void_p),
                                                             float(sq8_mean_sum_squares), None)
        elif sq8:                         # params is a BFParams here
            self._h = self._lib.VecSimGpu_NewFlatSQ8(C.byref(params), None)
        else:
            self._h = self._lib.VecSimIndex_New(C.byref(params))
        if not self._h:
            err = self._lib.VecSimGpu_LastError()
            raise RuntimeError("VecSimIndex_New failed: %s" % (err.decode() if err else "unsupported parameters"))
        info = self._lib.VecSimIndex_BasicInfo(self._h)
        self._type, self._dim, self._metric = info.type, info.dim, info.metric
        self._qbytes = self._lib.VecSimParams_GetQueryBlobSize(self._type, self._dim, self._metric)

    # ---- helpers ----
    def _blob(self, a, rows=None):
        """C-contiguous array of the index dtype (raw u16 for bf16/fp16) holding >= blob bytes per row."""
        a = np.asarray(a)
        want = _NP[self._type]
        if a.dtype != want:
            if self._type in (VecSimType_BFLOAT16, VecSimType_FLOAT16) and a.dtype.itemsize == 2:
                a = a.view(np.uint16)
            else:
                a = a.astype(want)
        return np.ascontiguousarray(a)

    def _padded(self, a):
        """queries/vectors for int8/uint8 Cosine need dim+4 bytes of room (the norm is appended)."""
        a = self._blob(a)
        a2 = a.reshape(-1, a.shape[-1])
        need = self._qbytes
        have = a2.shape[1] * a2.dtype.itemsize
        if have >= need:
            return a2, a2.strides[0]
        buf = np.zeros((a2.shape[0], need), dtype=np.uint8)
        buf[:, :have] = a2.view(np.uint8).reshape(a2.shape[0], have)
        return buf, need

    # ---- reference surface ----
    def add_vector(self, vector, label):
        v, _ = self._padded(vector)
        return self._lib.VecSimIndex_AddVector(self._h, v.ctypes.data_as(C.c_void_p), int(label))

    def add_vectors(self, vectors, labels):
        """bulk ingest extension (one H2D per 8 MiB instead of one call per vector)"""
        v = self._blob(vectors)
        lab = np.ascontiguousarray(labels, dtype=np.uint64)
        assert v.ndim == 2 and v.shape[0] == lab.size
        n = self._lib.VecSimIndex_AddVectorsBulk(self._h, v.ctypes.data_as(C.c_void_p),
                                                 lab.ctypes.data_as(C.c_void_p), lab.size)
        if n < 0:
            raise RuntimeError("bulk add failed (duplicate label or GPU error)")
        return n

    def add_synthetic(self, n, seed):
        r = self._lib.VecSimIndex_AddSyntheticVectors(self._h, n, seed)
        if r < 0:
            raise RuntimeError("synthetic fill failed: %s" % self._lib.VecSimGpu_LastError().decode())
        return r

    def delete_vector(self, label):
        return self._lib.VecSimIndex_DeleteVector(self._h, int(label))

    def knn_query(self, vector, k, query_param=None, order=BY_SCORE):
        q, stride = self._padded(vector)
        nq = q.shape[0]
        qp = C.byref(query_param) if query_param is not None else None
        if nq == 1:
            rep = self._lib.VecSimIndex_TopKQuery(self._h, q.ctypes.data_as(C.c_void_p), k, qp, order)
            return _wrap(self._lib, [rep], k)
        labels = np.empty((nq, k), dtype=np.int64)
        dists = np.empty((nq, k), dtype=np.float64)
        rc = self._lib.VecSimIndex_TopKQueryBatchArrays(self._h, q.ctypes.data_as(C.c_void_p), nq, stride, k, qp, order,
                                                        labels.ctypes.data_as(C.c_void_p),
                                                        dists.ctypes.data_as(C.c_void_p), None)
        if rc != 0:
            raise RuntimeError("GPU batched top-k failed: %s" % self._lib.VecSimGpu_LastError().decode())
        return labels, dists

    def knn_query_replies(self, vector, k, query_param=None, order=BY_SCORE):
        """batched query through the reply-object entry point (VecSimIndex_TopKQueryBatch)"""
        q, stride = self._padded(vector)
        nq = q.shape[0]
        qp = C.byref(query_param) if query_param is not None else None
        reps = (C.c_void_p * nq)()
        rc = self._lib.VecSimIndex_TopKQueryBatch(self._h, q.ctypes.data_as(C.c_void_p), nq, stride, k,
                                                  qp, order, reps)
        if rc != 0:
            raise RuntimeError("GPU batched top-k failed: %s" % self._lib.VecSimGpu_LastError().decode())
        return _wrap(self._lib, list(reps), k)

    def topk_candidates(self, queries, k, cap, ids, labels, scores, counts):
        """shard-side half of a multi-GPU query (VecSimIndex_TopKCandidatesBatch); fills the arrays"""
        q, stride = self._padded(queries)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = self._lib.VecSimIndex_TopKCandidatesBatch(self._h, p(q), q.shape[0], stride, k, cap, p(ids), p(labels),
                                                       p(scores), p(counts))
        if rc != 0:
            raise RuntimeError("GPU candidate pass failed: %s" % self._lib.VecSimGpu_LastError().decode())

    def knn_query_code(self, vector, k, query_param=None):
        """(labels, distances, reply code) for a single query -- lets tests see TimedOut"""
        q, _ = self._padded(vector)
        qp = C.byref(query_param) if query_param is not None else None
        rep = self._lib.VecSimIndex_TopKQuery(self._h, q.ctypes.data_as(C.c_void_p), k, qp, BY_SCORE)
        code = self._lib.VecSimQueryReply_GetCode(rep)
        lab, d = _wrap(self._lib, [rep], k)
        return lab, d, code

    def range_query(self, vector, radius, query_param=None, order=BY_SCORE):
        q, _ = self._padded(vector)
        qp = C.byref(query_param) if query_param is not None else None
        rep = self._lib.VecSimIndex_RangeQuery(self._h, q.ctypes.data_as(C.c_void_p), float(radius), qp, order)
        return _wrap(self._lib, [rep], self._lib.VecSimQueryReply_Len(rep))

    def get_distance_from(self, label, vector):
        q, _ = self._padded(vector)
        return self._lib.VecSimIndex_GetDistanceFrom_Unsafe(self._h, int(label), q.ctypes.data_as(C.c_void_p))

    def index_size(self):
        return self._lib.VecSimIndex_IndexSize(self._h)

    def index_memory(self):
        """bindings.cpp:205 -- bytes held for the index (host bookkeeping + device table)"""
        return int(self._lib.VecSimIndex_StatsInfo(self._h).memory)

    def run_gc(self):
        self._lib.VecSimTieredIndex_GC(self._h)

    def get_vector(self, label):
        """bindings.cpp:219-235 -- the stored vector(s) of a label as a 2-D array: fp32/fp64 as stored, bf16/fp16
        widened to float32, int8 as float32 (the reference's NPArrayType), Cosine rows as normalised at ingest"""
        bb = C.c_size_t(0)
        self._lib.VecSimGpu_GetStoredVectors(self._h, int(label), None, 0, C.byref(bb))
        cap = 64
        while True:
            buf = np.zeros(cap * bb.value, dtype=np.uint8)
            n = self._lib.VecSimGpu_GetStoredVectors(self._h, int(label), buf.ctypes.data_as(C.c_void_p), buf.nbytes, C.byref(bb))
            if n >= 0:
                break
            cap *= 8
            if cap > 1 << 20:
                raise RuntimeError("get_vector failed")
        rows = buf[: n * bb.value].reshape(n, bb.value)
        t, dim = self._type, self._dim
        if t == VecSimType_FLOAT32:
            return rows[:, : dim * 4].copy().view(np.float32)
        if t == VecSimType_FLOAT64:
            return rows[:, : dim * 8].copy().view(np.float64)
        if t == VecSimType_BFLOAT16:
            h = rows[:, : dim * 2].copy().view(np.uint16).astype(np.uint32) << 16
            return h.view(np.float32)
        if t == VecSimType_FLOAT16:
            return rows[:, : dim * 2].copy().view(np.float16).astype(np.float32)
        if t == VecSimType_INT8:
            return rows[:, :dim].copy().view(np.int8).astype(np.float32)
        return rows[:, :dim].copy().astype(np.float32)

    def stored_rows(self, first_id, n, out=None):
        """the stored blobs of internal ids [first_id, first_id + n) as a uint8 array [n, blob bytes] (Flat indexes; Cosine rows
        as normalised at ingest, int8 / uint8 Cosine with the trailing float norm); `out`: a uint8 buffer to reuse"""
        bb = C.c_size_t(0)
        self._lib.VecSimGpu_GetStoredVectors(self._h, 0, None, 0, C.byref(bb))
        if out is None:
            out = np.empty((n, bb.value), dtype=np.uint8)
        else:
            out = out.reshape(-1)[: n * bb.value].reshape(n, bb.value)
        got = self._lib.VecSimGpu_ReadStoredRows(self._h, int(first_id), int(n), out.ctypes.data_as(C.c_void_p), out.nbytes)
        if got != n:
            raise RuntimeError("stored_rows failed")
        return out

    def debug_info_fields(self):
        """[(name, value)] in the order VecSimIndex_DebugInfoIterator yields them (info_iterator.h)"""
        it = self._lib.VecSimIndex_DebugInfoIterator(self._h)
        out = []
        try:
            n = self._lib.VecSimDebugInfoIterator_NumberOfFields(it)
            while self._lib.VecSimDebugInfoIterator_HasNextField(it):
                f = self._lib.VecSimDebugInfoIterator_NextField(it).contents
                v = f.fieldValue
                val = {0: lambda: v.stringValue.decode() if v.stringValue else None, 1: lambda: v.integerValue,
                       2: lambda: v.uintegerValue, 3: lambda: v.floatingPointValue}[f.fieldType]()
                out.append((f.fieldName.decode(), val))
            assert len(out) == n
        finally:
            self._lib.VecSimDebugInfoIterator_Free(it)
        return out

    def distance_tier(self):
        """which reference ISA tier's summation order this index answers in: 'AVX512' | 'AVX512_BF16' | 'AVX512_FP16' | 'SCALAR'"""
        return self._lib.VecSimGpu_IndexTier(self._h).decode()

    def index_type(self):
        return self._type

    def create_batch_iterator(self, query_blob, query_param=None):
        q, _ = self._padded(query_blob)
        qp = C.byref(query_param) if query_param is not None else None
        h = self._lib.VecSimBatchIterator_New(self._h, q.ctypes.data_as(C.c_void_p), qp)
        return BatchIterator(self, h)

    def prefer_adhoc(self, subset, k, initial=True):
        return bool(self._lib.VecSimIndex_PreferAdHocSearch(self._h, subset, k, initial))

    # ---- measurement ----
    def reset_stats(self):
        self._lib.VecSimGpu_ResetStats(self._h)

    def stats(self):
        s = _capi.VecSimGpuStats()
        self._lib.VecSimGpu_GetStats(self._h, C.byref(s))
        return {"scan_ms": s.scan_ms, "scan_launches": s.scan_launches, "scan_rows": s.scan_rows,
                "scan_bytes": s.scan_bytes, "other_ms": s.other_ms, "candidates": s.candidates,
                "fallbacks": s.fallbacks, "retries": s.retries, "scan_kernel": s.scan_kernel.decode()}

    def set_option(self, name, value):
        if self._lib.VecSimGpu_SetOption(self._h, name.encode(), int(value)) != 0:
            raise ValueError(name)

    def __del__(self):
        if getattr(self, "_h", None):
            if self._owned:
                self._lib.VecSimIndex_Free(self._h)
            self._h = None


class BFIndex(VecSimIndex):
    def __init__(self, params):
        p = VecSimParams()
        p.algo = VecSimAlgo_BF
        p.algoParams.bfParams = params
        super().__init__(p)


class SQ8Index(VecSimIndex):
    """Flat index over SQ8 storage (include/VecSim/vec_sim_gpu.h: VecSimGpu_NewFlatSQ8): fp32 vectors in, uint8 codes + FP32
    metadata in HBM, the reference's asymmetric SQ8 x FP32 distances (types/sq8.h, IP.cpp:34-80, L2.cpp:30-45) out."""

    def __init__(self, params, mean=None, mean_sum_squares=None):
        """mean (dim fp32 values): mean-centred blobs and DistanceCalculatorWithNorm scores (L2 / IP; vec_sim_gpu.h:
        VecSimGpu_NewFlatSQ8Centered); mean_sum_squares defaults to the sequential fp32 sum of squares the reference's
        tests use (test_components.cpp computeMeanSumSquares)"""
        if mean is not None and mean_sum_squares is None:
            acc = np.float32(0)
            for m in np.asarray(mean, dtype=np.float32):
                acc = np.float32(acc + np.float32(m * m))
            mean_sum_squares = float(acc)
        super().__init__(params, sq8=True, sq8_mean=mean, sq8_mean_sum_squares=mean_sum_squares or 0.0)

    def stored_distance(self, label_a, label_b):
        """symmetric SQ8 x SQ8 distance between two stored vectors (IP.cpp:146-183, L2.cpp:185-201)"""
        return self._lib.VecSimGpu_SQ8_StoredDistance(self._h, int(label_a), int(label_b))

    def get_vector(self, label):
        """the stored SQ8 blob(s) of a label as a 2-D uint8 array: dim codes + FP32 {min, delta, sum[, sum_squares]}"""
        bb = C.c_size_t(0)
        self._lib.VecSimGpu_GetStoredVectors(self._h, int(label), None, 0, C.byref(bb))
        buf = np.zeros(64 * bb.value, dtype=np.uint8)
        n = self._lib.VecSimGpu_GetStoredVectors(self._h, int(label), buf.ctypes.data_as(C.c_void_p), buf.nbytes, C.byref(bb))
        if n < 0:
            raise RuntimeError("get_vector failed")
        return buf[: n * bb.value].reshape(n, bb.value).copy()


def sq8_quantize(vector, metric):
    """QuantPreprocessor storage blob of one fp32 vector (host only): uint8 array, dim + 12 bytes (16 for L2)."""
    lib = _capi.load()
    v = np.ascontiguousarray(vector, dtype=np.float32)
    out = np.zeros(lib.VecSimGpu_SQ8_StorageBlobSize(v.size, int(metric)), dtype=np.uint8)
    lib.VecSimGpu_SQ8_Quantize(v.ctypes.data_as(C.c_void_p), v.size, int(metric), out.ctypes.data_as(C.c_void_p))
    return out


def sq8_query_blob(vector, metric):
    """Query blob of one fp32 vector (host only): the values followed by y_sum (and y_sum_squares for L2)."""
    lib = _capi.load()
    v = np.ascontiguousarray(vector, dtype=np.float32)
    out = np.zeros(lib.VecSimGpu_SQ8_QueryBlobSize(v.size, int(metric)) // 4, dtype=np.float32)
    lib.VecSimGpu_SQ8_QueryBlob(v.ctypes.data_as(C.c_void_p), v.size, int(metric), out.ctypes.data_as(C.c_void_p))
    return out


def sq8_quantize_centred(vector, mean, metric):
    """mean-centred storage blob (QuantPreprocessor<float, metric, WithNorm = true>): dim + 16 bytes"""
    lib = _capi.load()
    v = np.ascontiguousarray(vector, dtype=np.float32)
    m = np.ascontiguousarray(mean, dtype=np.float32)
    out = np.zeros(lib.VecSimGpu_SQ8_StorageBlobSizeCentered(v.size, int(metric)), dtype=np.uint8)
    lib.VecSimGpu_SQ8_QuantizeCentered(v.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p), v.size, int(metric),
                                       out.ctypes.data_as(C.c_void_p))
    return out


def sq8_query_blob_centred(vector, mean, metric):
    """mean-centred query blob: the values (centred for L2), y_sum, then y_sum_squares (L2) or y_mean_ip (IP)"""
    lib = _capi.load()
    v = np.ascontiguousarray(vector, dtype=np.float32)
    m = np.ascontiguousarray(mean, dtype=np.float32)
    out = np.zeros(lib.VecSimGpu_SQ8_QueryBlobSizeCentered(v.size, int(metric)) // 4, dtype=np.float32)
    lib.VecSimGpu_SQ8_QueryBlobCentered(v.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p), v.size, int(metric),
                                        out.ctypes.data_as(C.c_void_p))
    return out


class HNSWIndex(VecSimIndex):
    """HNSW index (reference: PyHNSWLibIndex, bindings.cpp:243-420): graph built on the host, queries on the GPU"""

    def __init__(self, params):
        p = VecSimParams()
        p.algo = VecSimAlgo_HNSWLIB
        p.algoParams.hnswParams = params
        super().__init__(p)

    def set_ef(self, ef):
        self._ef = int(ef)

    def knn_query(self, vector, k, query_param=None, order=BY_SCORE):
        if query_param is None and getattr(self, "_ef", 0):
            query_param = VecSimQueryParams()
            query_param.hnswRuntimeParams.efRuntime = self._ef
        return super().knn_query(vector, k, query_param, order)

    def knn_parallel(self, queries, k, query_param=None, num_threads=-1):
        """reference signature (bindings.cpp:330-345); here the whole batch is one GPU launch"""
        return self.knn_query(queries, k, query_param)

    def add_vector_parallel(self, vectors, labels, num_threads=-1):
        """reference signature (bindings.cpp:383-426); the bulk builder links on all host cores"""
        return self.add_vectors(vectors, labels)

    def range_parallel(self, queries, radius, query_param=None, num_threads=-1):
        """reference signature (bindings.cpp:347-381): one (labels, distances) pair per query, padded like knn"""
        res = [self.range_query(q, radius, query_param) for q in np.atleast_2d(queries)]
        width = max([r[0].shape[1] for r in res] + [1])
        labels = np.full((len(res), width), -1, dtype=np.int64)
        dists = np.full((len(res), width), -1.0, dtype=np.float64)
        for i, (l, d) in enumerate(res):
            labels[i, : l.shape[1]] = l[0]
            dists[i, : d.shape[1]] = d[0]
        return labels, dists

    def check_integrity(self):
        """structural half of HNSWIndex::checkIntegrity (hnsw_serializer_impl.h:57-140): every link names an existing
        node other than its owner and no neighbour appears twice in a list"""
        g = self.graph()
        n, M0, M = g["n"], g["M0"], g["M"]
        cnt = g["cnt0"].astype(np.int64)
        if np.any(cnt > M0):
            return False
        links = g["links0"].astype(np.int64)
        col = np.arange(M0)[None, :]
        live = col < cnt[:, None]
        if np.any(links[live] >= n) or np.any((links == np.arange(n)[:, None]) & live):
            return False
        srt = np.sort(np.where(live, links, -1 - col), axis=1)
        if np.any((srt[:, 1:] == srt[:, :-1]) & (srt[:, 1:] >= 0)):
            return False
        up = g["upper"].astype(np.int64)
        for b in range(0, len(up) - M, 1 + M):
            c = up[b]
            if c == 0:
                continue
            if c > M:
                return False
            nb = up[b + 1: b + 1 + c]
            if np.any(nb >= n) or len(set(nb.tolist())) != c:
                return False
        return True

    def graph(self):
        """dict of numpy arrays describing the built graph (tests / tooling)"""
        info = (C.c_uint64 * 6)()
        if self._lib.VecSimGpu_HnswGraphInfo(self._h, info) != 0:
            raise RuntimeError("not an HNSW index")
        n, M, M0, entry, max_level, uw = [int(x) for x in info]
        g = {"n": n, "M": M, "M0": M0, "entry": entry, "max_level": -1 if max_level == 0xFFFFFFFF else max_level,
             "links0": np.zeros((n, M0), dtype=np.uint32), "cnt0": np.zeros(n, dtype=np.uint16),
             "upper_off": np.zeros(n, dtype=np.uint32), "upper": np.zeros(max(uw, 1), dtype=np.uint32),
             "deleted": np.zeros(n, dtype=np.uint8), "labels": np.zeros(n, dtype=np.uint64)}
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._lib.VecSimGpu_HnswGraphCopy(self._h, p(g["links0"]), p(g["cnt0"]), p(g["upper_off"]), p(g["upper"]),
                                          p(g["deleted"]), p(g["labels"]))
        g["levels"] = np.zeros(n, dtype=np.uint8)
        g["reference_order_build"] = self._lib.VecSimGpu_HnswLevels(self._h, p(g["levels"])) == 1
        return g

    def last_distance_evals(self):
        return int(self._lib.VecSimGpu_HnswLastDistanceEvals(self._h))


def normalize(vector, vtype):
    """VecSim_Normalize on a copy; int8/uint8 return dim+4 bytes (norm appended)"""
    lib = _capi.load()
    a = np.ascontiguousarray(vector)
    dim = a.size
    if vtype in (VecSimType_INT8, VecSimType_UINT8):
        buf = np.zeros(dim + 4, dtype=np.uint8)
        buf[:dim] = a.view(np.uint8)
        lib.VecSim_Normalize(buf.ctypes.data_as(C.c_void_p), dim, vtype)
        return buf
    out = a.copy()
    lib.VecSim_Normalize(out.ctypes.data_as(C.c_void_p), dim, vtype)
    return out


def set_timeout_callback(fn):
    """fn(ctx:int) -> int, or None to clear.  Keep the returned object alive."""
    lib = _capi.load()
    if fn is None:
        lib.VecSim_SetTimeoutCallbackFunction(None)
        return None
    cb = _capi.TIMEOUT_CB(fn)
    lib.VecSim_SetTimeoutCallbackFunction(C.cast(cb, C.c_void_p))
    return cb
