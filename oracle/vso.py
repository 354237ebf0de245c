"""ctypes loader for the CPU oracle (oracle/libvso.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (vectorsimilarity_amd) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libvso.so")

F32, F64, BF16, F16, I8, U8 = range(6)
L2, IP, COSINE = range(3)
TIER_AVX512, TIER_SCALAR, TIER_AVX512_BF16, TIER_AVX512_FP16 = range(4)

NP_DTYPE = {F32: np.float32, F64: np.float64, BF16: np.uint16, F16: np.uint16, I8: np.int8,
            U8: np.uint8}


def build(force=False):
    """Compile oracle/libvso.so from the C restatement (gcc + make)."""
    srcs = [os.path.join(_HERE, f) for f in ("vso.c", "vso_fast.c", "vso_hnsw.c", "vso_sq8.c", "vso_stdsort.cpp", "vso.h", "Makefile")]
    if (not force and os.path.exists(_LIB)
            and os.path.getmtime(_LIB) >= max(os.path.getmtime(s) for s in srcs)):
        return _LIB
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        vp, sz, dbl, i = C.c_void_p, C.c_size_t, C.c_double, C.c_int
        L.vso_blob_size.restype = sz
        L.vso_blob_size.argtypes = [i, i, sz]
        L.vso_distance.restype = dbl
        L.vso_distance.argtypes = [i, i, i, sz, vp, vp]
        L.vso_distance_fast.restype = dbl
        L.vso_distance_fast.argtypes = [i, i, sz, vp, vp]
        L.vso_distance_fast_tier.restype = dbl
        L.vso_distance_fast_tier.argtypes = [i, i, i, sz, vp, vp]
        L.vso_fast_available.restype = i
        L.vso_fast_available.argtypes = [i, i, i, sz]
        L.vso_flat_topk_batch_fast_tier.restype = i
        L.vso_flat_topk_batch_fast_tier.argtypes = [i, i, i, sz, vp, sz, sz, vp, sz, sz, sz, i, vp, vp]
        L.vso_scan_batch_fast_tier.restype = i
        L.vso_scan_batch_fast_tier.argtypes = [i, i, i, sz, vp, sz, sz, vp, sz, sz, i, vp]
        L.vso_uses_scalar.restype = i
        L.vso_uses_scalar.argtypes = [i, i, i, sz]
        L.vso_scan.restype = None
        L.vso_scan.argtypes = [i, i, i, sz, vp, sz, sz, vp, vp]
        L.vso_normalize.restype = None
        L.vso_normalize.argtypes = [vp, sz, i]
        L.vso_f32_to_bf16.restype = C.c_uint16
        L.vso_f32_to_bf16.argtypes = [C.c_float]
        L.vso_bf16_to_f32.restype = C.c_float
        L.vso_bf16_to_f32.argtypes = [C.c_uint16]
        L.vso_f32_to_f16.restype = C.c_uint16
        L.vso_f32_to_f16.argtypes = [C.c_float]
        L.vso_f16_to_f32.restype = C.c_float
        L.vso_f16_to_f32.argtypes = [C.c_uint16]
        for name, n_args in (("vso_h_fma", 3), ("vso_h_mul", 2), ("vso_h_add", 2), ("vso_h_sub", 2)):
            fn = getattr(L, name)
            fn.restype = C.c_uint16
            fn.argtypes = [C.c_uint16] * n_args
        L.vso_topk_replay.restype = sz
        L.vso_topk_replay.argtypes = [vp, vp, sz, sz, vp, vp]
        L.vso_topk_replay_multi.restype = sz
        L.vso_topk_replay_multi.argtypes = [vp, vp, sz, sz, vp, vp]
        L.vso_range_replay.restype = sz
        L.vso_range_replay.argtypes = [vp, vp, sz, dbl, vp, vp]
        L.vso_flat_topk.restype = sz
        L.vso_flat_topk.argtypes = [i, i, i, sz, vp, sz, sz, vp, vp, sz, vp, vp]
        L.vso_flat_topk_batch_fast.restype = i
        L.vso_flat_topk_batch_fast.argtypes = [i, i, sz, vp, sz, sz, vp, sz, sz, sz, i, vp, vp]
        L.vso_hnsw_search.restype = sz
        L.vso_hnsw_search.argtypes = [i, i, i, sz, vp, sz, C.c_uint32, vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp, vp,
                                      C.c_uint32, i, vp, sz, sz, vp, vp, vp]
        L.vso_hnsw_search_multi.restype = sz
        L.vso_hnsw_search_multi.argtypes = L.vso_hnsw_search.argtypes
        L.vso_hnsw_range.restype = sz
        L.vso_hnsw_range.argtypes = [i, i, i, sz, vp, sz, C.c_uint32, vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp, vp,
                                     C.c_uint32, i, vp, dbl, dbl, vp, vp, sz, vp]
        L.vso_sq8_storage_size.restype = sz
        L.vso_sq8_storage_size.argtypes = [i, sz]
        L.vso_sq8_query_size.restype = sz
        L.vso_sq8_query_size.argtypes = [i, sz]
        L.vso_sq8_quantize.restype = None
        L.vso_sq8_quantize.argtypes = [vp, sz, i, vp]
        L.vso_sq8_query_blob.restype = None
        L.vso_sq8_query_blob.argtypes = [vp, sz, i, vp]
        L.vso_sq8_fp32_distance.restype = dbl
        L.vso_sq8_fp32_distance.argtypes = [i, i, sz, vp, vp]
        L.vso_sq8_sq8_distance.restype = dbl
        L.vso_sq8_sq8_distance.argtypes = [i, i, sz, vp, vp]
        L.vso_sq8_fp32_scan.restype = None
        L.vso_sq8_fp32_scan.argtypes = [i, i, sz, vp, sz, sz, vp, vp]
        L.vso_sq8_query_size_f16.restype = sz
        L.vso_sq8_query_size_f16.argtypes = [i, sz]
        L.vso_sq8_quantize_f16.restype = None
        L.vso_sq8_quantize_f16.argtypes = [vp, sz, i, vp]
        L.vso_sq8_query_blob_f16.restype = None
        L.vso_sq8_query_blob_f16.argtypes = [vp, sz, i, vp]
        L.vso_sq8_fp16_distance.restype = dbl
        L.vso_sq8_fp16_distance.argtypes = [i, i, sz, vp, vp]
        L.vso_sq8_fp16_scan.restype = None
        L.vso_sq8_fp16_scan.argtypes = [i, i, sz, vp, sz, sz, vp, vp]
        for name in ("vso_sq8_storage_size_norm", "vso_sq8_query_size_norm", "vso_sq8_query_size_norm_f16"):
            getattr(L, name).restype = sz
            getattr(L, name).argtypes = [i, sz]
        for name in ("vso_sq8_quantize_norm", "vso_sq8_quantize_norm_f16", "vso_sq8_query_blob_norm", "vso_sq8_query_blob_norm_f16"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [vp, vp, sz, i, vp]
        for name in ("vso_sq8_fp32_distance_norm", "vso_sq8_fp16_distance_norm"):
            getattr(L, name).restype = dbl
            getattr(L, name).argtypes = [i, i, sz, vp, vp]
        L.vso_sq8_sq8_distance_norm.restype = dbl
        L.vso_sq8_sq8_distance_norm.argtypes = [i, i, sz, vp, vp, C.c_float]
        for name in ("vso_sq8_fp32_scan_norm", "vso_sq8_fp16_scan_norm"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [i, i, sz, vp, sz, sz, vp, vp]
        L.vso_has_avx512.restype = i
        L.vso_has_f16c.restype = i
        L.vso_f16c_distance.restype = dbl
        L.vso_f16c_distance.argtypes = [i, sz, vp, vp]
        L.vso_f16c_distance_hw.restype = dbl
        L.vso_f16c_distance_hw.argtypes = [i, sz, vp, vp]
        L.vso_has_avx512_bf16.restype = i
        L.vso_probe_dpbf16.restype = None
        L.vso_probe_dpbf16.argtypes = [vp, vp, vp]
        L.vso_hash32.restype = C.c_uint32
        L.vso_hash32.argtypes = [C.c_uint64, C.c_uint64]
        L.vso_synth_f32.restype = C.c_float
        L.vso_synth_f32.argtypes = [C.c_uint64, C.c_uint64]
        L.vso_synth_rows_f32.restype = None
        L.vso_synth_rows_f32.argtypes = [C.c_uint64, C.c_uint64, sz, vp]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def blob_size(vtype, metric, dim):
    return lib().vso_blob_size(vtype, metric, dim)


def h_fma(a, b, c):
    """IEEE half fused multiply-add on bit patterns (one rounding, ties to even): the AVX512-FP16 tier's vfmadd...ph"""
    return int(lib().vso_h_fma(int(a), int(b), int(c)))


def h_mul(a, b):
    return int(lib().vso_h_mul(int(a), int(b)))


def h_add(a, b):
    return int(lib().vso_h_add(int(a), int(b)))


def h_sub(a, b):
    return int(lib().vso_h_sub(int(a), int(b)))


def distance(vtype, metric, a, b, dim=None, tier=TIER_AVX512):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if dim is None:
        dim = a.size
    return lib().vso_distance(vtype, metric, tier, dim, _ptr(a), _ptr(b))


def distance_fast(vtype, metric, a, b, dim=None, tier=TIER_AVX512):
    """the intrinsics twin of distance() on the host CPU (vso_fast.c); NaN when the host lacks the tier's instructions"""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if dim is None:
        dim = a.size
    return lib().vso_distance_fast_tier(vtype, metric, tier, dim, _ptr(a), _ptr(b))


def fast_available(vtype, metric, dim, tier=TIER_AVX512):
    return bool(lib().vso_fast_available(vtype, metric, tier, dim))


def scan(vtype, metric, rows, query, dim, tier=TIER_AVX512):
    """rows: 2-D C-contiguous array, one stored blob per row (raw bytes view allowed)."""
    rows = np.ascontiguousarray(rows)
    query = np.ascontiguousarray(query)
    n = rows.shape[0]
    out = np.empty(n, dtype=np.float64)
    lib().vso_scan(vtype, metric, tier, dim, _ptr(rows), n, rows.strides[0], _ptr(query), _ptr(out))
    return out


# ---- SQ8 (vso_sq8.c): uint8 codes + FP32 metadata; metric is the index metric (Cosine blobs are normalised first) ----
def sq8_quantize(x, metric):
    """Storage blob of one fp32 vector (QuantPreprocessor<float, metric>): uint8 array of dim + 12 (16 for L2) bytes."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(lib().vso_sq8_storage_size(metric, x.size), dtype=np.uint8)
    lib().vso_sq8_quantize(_ptr(x), x.size, metric, _ptr(out))
    return out


def sq8_query_blob(y, metric):
    """Query blob: the fp32 values followed by y_sum (and y_sum_squares for L2)."""
    y = np.ascontiguousarray(y, dtype=np.float32)
    out = np.zeros(lib().vso_sq8_query_size(metric, y.size) // 4, dtype=np.float32)
    lib().vso_sq8_query_blob(_ptr(y), y.size, metric, _ptr(out))
    return out


def sq8_fp32_distance(metric, storage, query, dim, tier=TIER_AVX512):
    storage = np.ascontiguousarray(storage)
    query = np.ascontiguousarray(query)
    return lib().vso_sq8_fp32_distance(metric, tier, dim, _ptr(storage), _ptr(query))


def sq8_sq8_distance(metric, a, b, dim, tier=TIER_AVX512):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return lib().vso_sq8_sq8_distance(metric, tier, dim, _ptr(a), _ptr(b))


def sq8_fp32_scan(metric, rows, query, dim, tier=TIER_AVX512):
    rows = np.ascontiguousarray(rows)
    query = np.ascontiguousarray(query)
    out = np.empty(rows.shape[0], dtype=np.float64)
    lib().vso_sq8_fp32_scan(metric, tier, dim, _ptr(rows), rows.shape[0], rows.strides[0], _ptr(query), _ptr(out))
    return out


def sq8_quantize_f16(x, metric):
    """storage blob of one fp16 vector (uint16 bit patterns): QuantPreprocessor<float16, metric>"""
    x = np.ascontiguousarray(x, dtype=np.uint16)
    out = np.zeros(lib().vso_sq8_storage_size(metric, x.size), dtype=np.uint8)
    lib().vso_sq8_quantize_f16(_ptr(x), x.size, metric, _ptr(out))
    return out


def sq8_query_blob_f16(y, metric):
    """query blob of one fp16 vector: the fp16 values followed by FP32 y_sum (y_sum_squares), as raw bytes"""
    y = np.ascontiguousarray(y, dtype=np.uint16)
    out = np.zeros(lib().vso_sq8_query_size_f16(metric, y.size), dtype=np.uint8)
    lib().vso_sq8_query_blob_f16(_ptr(y), y.size, metric, _ptr(out))
    return out


def sq8_fp16_distance(metric, storage, query, dim, tier=TIER_AVX512):
    storage = np.ascontiguousarray(storage)
    query = np.ascontiguousarray(query)
    return lib().vso_sq8_fp16_distance(metric, tier, dim, _ptr(storage), _ptr(query))


def sq8_fp16_scan(metric, rows, query, dim, tier=TIER_AVX512):
    rows = np.ascontiguousarray(rows)
    query = np.ascontiguousarray(query)
    out = np.empty(rows.shape[0], dtype=np.float64)
    lib().vso_sq8_fp16_scan(metric, tier, dim, _ptr(rows), rows.shape[0], rows.strides[0], _ptr(query), _ptr(out))
    return out


# mean-centred blobs (QuantPreprocessor<..., WithNorm = true> + DistanceCalculatorWithNorm); metric L2 or IP; f16 = uint16 input
def sq8_quantize_norm(x, mean, metric, f16=False):
    x = np.ascontiguousarray(x, dtype=np.uint16 if f16 else np.float32)
    mean = np.ascontiguousarray(mean, dtype=np.float32)
    out = np.zeros(lib().vso_sq8_storage_size_norm(metric, x.size), dtype=np.uint8)
    (lib().vso_sq8_quantize_norm_f16 if f16 else lib().vso_sq8_quantize_norm)(_ptr(x), _ptr(mean), x.size, metric, _ptr(out))
    return out


def sq8_query_blob_norm(y, mean, metric, f16=False):
    """raw bytes: the query body (centred for L2) followed by { y_sum, y_sum_squares | y_mean_ip }"""
    y = np.ascontiguousarray(y, dtype=np.uint16 if f16 else np.float32)
    mean = np.ascontiguousarray(mean, dtype=np.float32)
    size = (lib().vso_sq8_query_size_norm_f16 if f16 else lib().vso_sq8_query_size_norm)(metric, y.size)
    out = np.zeros(size, dtype=np.uint8)
    (lib().vso_sq8_query_blob_norm_f16 if f16 else lib().vso_sq8_query_blob_norm)(_ptr(y), _ptr(mean), y.size, metric, _ptr(out))
    return out


def sq8_distance_norm(metric, storage, query, dim, f16=False, tier=TIER_AVX512):
    storage = np.ascontiguousarray(storage)
    query = np.ascontiguousarray(query)
    fn = lib().vso_sq8_fp16_distance_norm if f16 else lib().vso_sq8_fp32_distance_norm
    return fn(metric, tier, dim, _ptr(storage), _ptr(query))


def sq8_sq8_distance_norm(metric, a, b, dim, mean_sum_squares, tier=TIER_AVX512):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return lib().vso_sq8_sq8_distance_norm(metric, tier, dim, _ptr(a), _ptr(b), float(mean_sum_squares))


def sq8_scan_norm(metric, rows, query, dim, f16=False, tier=TIER_AVX512):
    rows = np.ascontiguousarray(rows)
    query = np.ascontiguousarray(query)
    out = np.empty(rows.shape[0], dtype=np.float64)
    fn = lib().vso_sq8_fp16_scan_norm if f16 else lib().vso_sq8_fp32_scan_norm
    fn(metric, tier, dim, _ptr(rows), rows.shape[0], rows.strides[0], _ptr(query), _ptr(out))
    return out


def normalize(blob, dim, vtype):
    """In place; for int8/uint8 the array must have dim+4 bytes."""
    assert blob.flags["C_CONTIGUOUS"]
    lib().vso_normalize(_ptr(blob), dim, vtype)
    return blob


def topk_replay(scores, k, labels=None):
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    n = scores.size
    lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.uint64)
    kk = max(1, min(k, n))
    ol = np.empty(kk, dtype=np.uint64)
    osc = np.empty(kk, dtype=np.float64)
    c = lib().vso_topk_replay(_ptr(scores), None if lab is None else _ptr(lab), n, k, _ptr(ol),
                              _ptr(osc))
    return ol[:c].copy(), osc[:c].copy()


def topk_replay_multi(scores, k, labels):
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    lab = np.ascontiguousarray(labels, dtype=np.uint64)
    kk = max(1, k)
    ol = np.empty(kk, dtype=np.uint64)
    osc = np.empty(kk, dtype=np.float64)
    c = lib().vso_topk_replay_multi(_ptr(scores), _ptr(lab), scores.size, k, _ptr(ol), _ptr(osc))
    return ol[:c].copy(), osc[:c].copy()


def range_replay(scores, radius, labels=None):
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    n = scores.size
    lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.uint64)
    ol = np.empty(max(n, 1), dtype=np.uint64)
    osc = np.empty(max(n, 1), dtype=np.float64)
    c = lib().vso_range_replay(_ptr(scores), None if lab is None else _ptr(lab), n, radius,
                               _ptr(ol), _ptr(osc))
    return ol[:c].copy(), osc[:c].copy()


def flat_topk(vtype, metric, rows, query, k, dim, labels=None, tier=TIER_AVX512):
    rows = np.ascontiguousarray(rows)
    scores = scan(vtype, metric, rows, query, dim, tier)
    return topk_replay(scores, k, labels)


def flat_topk_batch_fast(vtype, metric, rows, queries, k, dim, threads=1, tier=TIER_AVX512):
    """Timing leg: returns (labels[nq,k] uint64, scores[nq,k] f64, used_intrinsics)."""
    rows = np.ascontiguousarray(rows)
    queries = np.ascontiguousarray(queries)
    nq = queries.shape[0]
    ol = np.empty((nq, k), dtype=np.uint64)
    osc = np.empty((nq, k), dtype=np.float64)
    fast = lib().vso_flat_topk_batch_fast_tier(vtype, metric, tier, dim, _ptr(rows), rows.shape[0],
                                               rows.strides[0], _ptr(queries), nq, queries.strides[0],
                                               k, threads, _ptr(ol), _ptr(osc))
    return ol, osc, bool(fast)


def scan_batch(vtype, metric, rows, queries, dim, threads=1, tier=TIER_AVX512):
    """scores[nq, n] (f64) of every query against every row, row blocks dealt over `threads` threads"""
    rows = np.ascontiguousarray(rows)
    queries = np.ascontiguousarray(queries)
    n, nq = rows.shape[0], queries.shape[0]
    out = np.empty((nq, n), dtype=np.float64)
    lib().vso_scan_batch_fast_tier(vtype, metric, tier, dim, _ptr(rows), n, rows.strides[0], _ptr(queries), nq,
                                   queries.strides[0], threads, _ptr(out))
    return out


class StreamTopK:
    """brute_force.h:242-291 over a table too large to hold on the host: feed() the stored rows piece by piece in internal-id
    order; the running set keeps every row whose score is <= the k-th smallest seen so far (a superset of what the sequential
    heap can ever hold: SURVEY.md 8a A10), result() replays the reference's heap over that set in id order."""

    def __init__(self, vtype, metric, queries, k, dim, threads=1, tier=TIER_AVX512):
        self.vtype, self.metric, self.k, self.dim, self.threads, self.tier = vtype, metric, k, dim, threads, tier
        self.queries = np.ascontiguousarray(queries)
        nq = self.queries.shape[0]
        self.ids = [np.empty(0, dtype=np.uint64) for _ in range(nq)]
        self.scores = [np.empty(0, dtype=np.float64) for _ in range(nq)]
        self.bound = [np.inf] * nq
        self.rows_seen = 0

    def feed(self, rows, first_id):
        sc = scan_batch(self.vtype, self.metric, rows, self.queries, self.dim, self.threads, self.tier)
        assert not np.isnan(sc).any()
        for q in range(len(self.ids)):
            keep = np.nonzero(sc[q] <= self.bound[q])[0]
            ids = np.concatenate([self.ids[q], keep.astype(np.uint64) + np.uint64(first_id)])
            scs = np.concatenate([self.scores[q], sc[q][keep]])
            if len(scs) > self.k:
                t = np.partition(scs, self.k - 1)[self.k - 1]
                m = scs <= t
                ids, scs, self.bound[q] = ids[m], scs[m], t
            self.ids[q], self.scores[q] = ids, scs
        self.rows_seen += rows.shape[0]

    def result(self, label_of=None):
        """label_of: ids (uint64 array) -> labels, for tables whose labels are not their internal ids (the heap's eviction rule reads
        the label: utils/vecsim_stl.h:63-83)"""
        nq = len(self.ids)
        labels = np.full((nq, self.k), -1, dtype=np.int64)
        scores = np.full((nq, self.k), np.nan)
        for q in range(nq):
            o = np.argsort(self.ids[q], kind="stable")
            ids = self.ids[q][o]
            l, s = topk_replay(self.scores[q][o], self.k, ids if label_of is None else np.asarray(label_of(ids), dtype=np.uint64))
            labels[q, :len(l)] = l.astype(np.int64)
            scores[q, :len(s)] = s
        return labels, scores


def synth_rows_f32(seed, first_row, nrows, dim):
    out = np.empty((nrows, dim), dtype=np.float32)
    lib().vso_synth_rows_f32(seed, first_row * dim, nrows * dim, _ptr(out))
    return out


def _conv(name, a, src, dst):
    a = np.ascontiguousarray(a, dtype=src)
    out = np.empty(a.shape, dtype=dst)
    f = getattr(lib(), name)
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    f(_ptr(a), a.size, _ptr(out))
    return out


def f32_to_bf16(a):
    return _conv("vso_f32_to_bf16_n", a, np.float32, np.uint16)


def f32_to_f16(a):
    return _conv("vso_f32_to_f16_n", a, np.float32, np.uint16)


def bf16_to_f32(a):
    return _conv("vso_bf16_to_f32_n", a, np.uint16, np.float32)


def f16_to_f32(a):
    return _conv("vso_f16_to_f32_n", a, np.uint16, np.float32)


def hnsw_build(vtype, metric, rows, dim, M, ef_construction, tier=TIER_AVX512, seed=100, fast=True, labels=None):
    """the graph the reference builds when the stored blobs `rows` are added in order (hnsw.h insert path, vso_hnsw.c); the same
    dict VecSim.HNSWIndex.graph() returns, plus "levels" and "dist_evals" """
    rows = np.ascontiguousarray(rows)
    n = rows.shape[0]
    L = lib()
    L.vso_hnsw_levels.restype = None
    L.vso_hnsw_levels.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.vso_hnsw_build.restype = C.c_uint64
    L.vso_hnsw_build.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.c_uint32, C.c_int] + [C.c_void_p] * 7
    levels = np.zeros(max(n, 1), dtype=np.uint8)
    L.vso_hnsw_levels(n, M, seed, _ptr(levels))
    g = {"n": n, "M": M, "M0": 2 * M, "links0": np.zeros((n, 2 * M), dtype=np.uint32), "cnt0": np.zeros(n, dtype=np.uint16),
         "upper_off": np.zeros(n, dtype=np.uint32), "upper": np.zeros(max(int(levels[:n].sum()) * (M + 1), 1), dtype=np.uint32),
         "deleted": np.zeros(n, dtype=np.uint8), "labels": np.arange(n, dtype=np.uint64) if labels is None else np.asarray(labels, dtype=np.uint64)}
    entry, maxl = C.c_uint32(0), C.c_int(0)
    g["dist_evals"] = int(L.vso_hnsw_build(vtype, metric, tier, dim, _ptr(rows), rows.strides[0], n, M, ef_construction, seed, int(fast),
                                           _ptr(g["links0"]), _ptr(g["cnt0"]), _ptr(levels), _ptr(g["upper_off"]), _ptr(g["upper"]),
                                           C.byref(entry), C.byref(maxl)))
    g["levels"], g["entry"], g["max_level"] = levels[:n], int(entry.value), int(maxl.value)
    return g


def graph_lists(g, levels=None):
    """{(node, level): [neighbour ids in list order]} of a graph dict (levels: per-node top level; default g["levels"])"""
    levels = g["levels"] if levels is None else levels
    M = g["M"]
    out = {}
    for i in range(g["n"]):
        out[(i, 0)] = [int(x) for x in g["links0"][i, : g["cnt0"][i]]]
        for lv in range(1, int(levels[i]) + 1):
            b = (int(g["upper_off"][i]) + lv - 1) * (M + 1)
            c = int(g["upper"][b])
            out[(i, lv)] = [int(x) for x in g["upper"][b + 1: b + 1 + c]]
    return out


def hnsw_search(vtype, metric, rows, graph, query, k, ef, dim, tier=TIER_AVX512, multi=False):
    """graph: dict from vectorsimilarity_amd.VecSim.HNSWIndex.graph(); rows: stored (preprocessed) blobs by id;
    multi: the graph's labels repeat (multi-value index), results are per label"""
    rows = np.ascontiguousarray(rows)
    query = np.ascontiguousarray(query)
    ol = np.zeros(max(k, 1), dtype=np.uint64)
    osc = np.zeros(max(k, 1), dtype=np.float64)
    ev = C.c_uint64(0)
    g = graph
    fn = lib().vso_hnsw_search_multi if multi else lib().vso_hnsw_search
    c = fn(vtype, metric, tier, dim, _ptr(rows), rows.strides[0], g["n"], _ptr(g["links0"]),
                              _ptr(g["cnt0"]), g["M0"], _ptr(g["upper_off"]), _ptr(g["upper"]), g["M"],
                              _ptr(g["deleted"]), _ptr(g["labels"]), g["entry"], g["max_level"], _ptr(query), k, ef,
                              _ptr(ol), _ptr(osc), C.byref(ev))
    return ol[:c].copy(), osc[:c].copy(), ev.value


def hnsw_iterate(vtype, metric, rows, graph, query, ef, sizes, dim, tier=TIER_AVX512, multi=False):
    """the reference's HNSW batch iterator (hnsw_batch_iterator.h:96-230) over an exported graph: batch b asks for sizes[b]
    results; returns ([(labels, scores) per batch taken], depleted)"""
    rows = np.ascontiguousarray(rows)
    query = np.ascontiguousarray(query)
    g = graph
    live = ~np.asarray(g["deleted"], dtype=bool)
    n_labels = len(set(np.asarray(g["labels"])[live].tolist()))
    sz = np.asarray(sizes, dtype=np.uint64)
    cap = int(sz.sum()) + 1
    ol = np.zeros(cap, dtype=np.uint64)
    osc = np.zeros(cap, dtype=np.float64)
    cnt = np.zeros(len(sz) + 1, dtype=np.uint64)
    dep = C.c_int(0)
    fn = lib().vso_hnsw_iterate
    fn.restype = C.c_size_t
    fn.argtypes = None
    nb = fn(C.c_int(vtype), C.c_int(metric), C.c_int(tier), C.c_size_t(dim), _ptr(rows), C.c_size_t(rows.strides[0]), C.c_uint32(g["n"]),
            _ptr(g["links0"]), _ptr(g["cnt0"]), C.c_uint32(g["M0"]), _ptr(g["upper_off"]), _ptr(g["upper"]), C.c_uint32(g["M"]),
            _ptr(g["deleted"]), _ptr(g["labels"]), C.c_uint32(g["entry"]), C.c_int(g["max_level"]), _ptr(query), C.c_size_t(ef),
            C.c_int(1 if multi else 0), C.c_size_t(n_labels), _ptr(sz), C.c_size_t(len(sz)), _ptr(ol), _ptr(osc), _ptr(cnt), C.byref(dep))
    out, at = [], 0
    for b in range(nb):
        c = int(cnt[b])
        out.append((ol[at:at + c].copy(), osc[at:at + c].copy()))
        at += c
    return out, bool(dep.value)


def hnsw_range(vtype, metric, rows, graph, query, radius, epsilon, dim, tier=TIER_AVX512):
    """restated HNSW range search over an exported graph: (labels, scores) in discovery order"""
    rows = np.ascontiguousarray(rows)
    query = np.ascontiguousarray(query)
    g = graph
    cap = max(int(g["n"]), 1)
    ol = np.zeros(cap, dtype=np.uint64)
    osc = np.zeros(cap, dtype=np.float64)
    ev = C.c_uint64(0)
    c = lib().vso_hnsw_range(vtype, metric, tier, dim, _ptr(rows), rows.strides[0], g["n"], _ptr(g["links0"]),
                             _ptr(g["cnt0"]), g["M0"], _ptr(g["upper_off"]), _ptr(g["upper"]), g["M"],
                             _ptr(g["deleted"]), _ptr(g["labels"]), g["entry"], g["max_level"], _ptr(query),
                             float(radius), float(epsilon), _ptr(ol), _ptr(osc), cap, C.byref(ev))
    return ol[:c].copy(), osc[:c].copy(), ev.value
