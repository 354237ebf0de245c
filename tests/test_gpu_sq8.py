"""GPU: the SQ8 index (VecSimGpu_NewFlatSQ8: uint8 codes + FP32 metadata in HBM, fp32 queries) against oracle/vso_sq8.c,
through the C API.  Labels, order and scores bit-exact (0 ulp): the kernels reproduce the reference's asymmetric SQ8 x FP32
distance in its AVX-512 tier order (IP_AVX512F_BW_VL_VNNI_SQ8_FP32.h:49-104; scalar tier IP.cpp:34-70 below dim 8) and the
symmetric SQ8 x SQ8 distance (IP.cpp:146-183, ..._SQ8_SQ8.h:38-65); the stored bytes are the QuantPreprocessor's."""
import numpy as np
import pytest

from vectorsimilarity_amd import VecSim

pytestmark = pytest.mark.gpu
MET = {"L2": 0, "IP": 1, "Cosine": 2}


def make(metric, dim):
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, dim, MET[metric]
    return VecSim.SQ8Index(p)


def oracle_blobs(vso, rows, queries, metric):
    """what the index stores / what its kernels are handed: Cosine vectors are normalised first"""
    m = MET[metric]
    rows = np.array(rows, dtype=np.float32, copy=True)
    queries = np.array(queries, dtype=np.float32, copy=True)
    if metric == "Cosine":
        for v in rows:
            vso.normalize(v, v.size, vso.F32)
        for v in queries:
            vso.normalize(v, v.size, vso.F32)
    st = np.stack([vso.sq8_quantize(v, m) for v in rows])
    qb = np.stack([vso.sq8_query_blob(v, m) for v in queries])
    return st, qb


@pytest.mark.parametrize("metric", ["L2", "IP", "Cosine"])
@pytest.mark.parametrize("dim", [1, 4, 7, 8, 17, 33, 64, 100, 128])
def test_sq8_all_scores_bit_exact(vso, metric, dim):
    """k = n returns every row: every distance and the full (score, label) order"""
    rng = np.random.default_rng(dim * 5 + len(metric))
    n = 300
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    q = rng.uniform(-1, 1, (3, dim)).astype(np.float32)
    ix = make(metric, dim)
    for i in range(n):
        ix.add_vector(rows[i], i)
    assert ix.index_size() == n
    labels, dists = ix.knn_query(q, n)
    st, qb = oracle_blobs(vso, rows, q, metric)
    for j in range(3):
        sc = vso.sq8_fp32_scan(MET[metric], st, qb[j], dim)
        el, es = vso.topk_replay(sc, n)
        assert np.array_equal(labels[j], el.astype(np.int64)), (metric, dim, j)
        assert np.array_equal(dists[j], es), (metric, dim, j)


@pytest.mark.parametrize("metric,dim", [("L2", 17), ("IP", 64), ("Cosine", 100)])
def test_sq8_scalar_tier(vso, monkeypatch, metric, dim):
    monkeypatch.setenv("VECSIM_GPU_TIER", "scalar")
    rng = np.random.default_rng(dim)
    n = 200
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    q = rng.uniform(-1, 1, (2, dim)).astype(np.float32)
    ix = make(metric, dim)
    ix.add_vectors(rows, np.arange(n))
    labels, dists = ix.knn_query(q, n)
    st, qb = oracle_blobs(vso, rows, q, metric)
    for j in range(2):
        sc = vso.sq8_fp32_scan(MET[metric], st, qb[j], dim, tier=vso.TIER_SCALAR)
        el, es = vso.topk_replay(sc, n)
        assert np.array_equal(labels[j], el.astype(np.int64)) and np.array_equal(dists[j], es), (metric, dim, j)


@pytest.mark.parametrize("metric,dim,n,nq,k", [
    ("L2", 768, 30_000, 40, 10),
    ("IP", 128, 50_000, 9, 10),
    ("Cosine", 1000, 12_000, 70, 5),
    ("L2", 96, 40_000, 1, 100),
])
def test_sq8_topk_filtered_path(vso, metric, dim, n, nq, k):
    """large enough for the probe -> threshold -> filter route; bulk add"""
    rng = np.random.default_rng(dim + n)
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    q = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    ix = make(metric, dim)
    ix.add_vectors(rows, np.arange(n))
    ix.set_option("dense_pairs", 0)
    ix.reset_stats()
    labels, dists = ix.knn_query(q, k)
    stt = ix.stats()
    # dims up to 1024 ride the int8 MFMA filter (one int8 piece per query element, rigorous bound, exact re-rank)
    assert stt["scan_kernel"] == "k_mfma_filter_lowp(sq8)" and stt["fallbacks"] == 0, stt
    ix.set_option("mfma", 0)
    l0, d0 = ix.knn_query(q[:4], k)
    assert np.array_equal(l0, labels[:4]) and np.array_equal(d0, dists[:4])
    st, qb = oracle_blobs(vso, rows, q, metric)
    for j in range(0, nq, 3):
        sc = vso.sq8_fp32_scan(MET[metric], st, qb[j], dim)
        el, es = vso.topk_replay(sc, k)
        assert np.array_equal(labels[j], el.astype(np.int64)), (metric, dim, j)
        assert np.array_equal(dists[j], es), (metric, dim, j)


def test_sq8_range_iterator_distance_delete_overwrite(vso):
    rng = np.random.default_rng(9)
    metric, dim, n = "L2", 48, 4000
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    q = rng.uniform(-1, 1, (1, dim)).astype(np.float32)
    ix = make(metric, dim)
    ix.add_vectors(rows, np.arange(n))
    st, qb = oracle_blobs(vso, rows, q, metric)
    sc = vso.sq8_fp32_scan(MET[metric], st, qb[0], dim)
    # range query: score <= radius, ascending score
    radius = float(np.sort(sc)[57])
    l, d = ix.range_query(q[0], radius)
    el, es = vso.range_replay(sc, radius)
    order = np.lexsort((el, es))
    assert np.array_equal(l[0][:len(el)], el[order].astype(np.int64)) and np.array_equal(d[0][:len(el)], es[order])
    # batch iterator: successive batches are the next-best rows
    it = ix.create_batch_iterator(q[0])
    got_l, got_d = [], []
    for _ in range(3):
        bl, bd = it.get_next_results(25, VecSim.BY_SCORE)
        got_l += list(bl[0])
        got_d += list(bd[0])
    el, es = vso.topk_replay(sc, 75)
    assert got_l == list(el.astype(np.int64)) and got_d == list(es)
    # distance to one stored vector
    assert ix.get_distance_from(123, q[0]) == sc[123]
    # delete (swap with the last row) and overwrite, then everything again
    ix.delete_vector(10)
    new = rng.uniform(-1, 1, dim).astype(np.float32)
    ix.add_vector(new, 20)
    rows2 = rows.copy()
    rows2[20] = new
    st2, _ = oracle_blobs(vso, rows2, q, metric)
    sc2 = vso.sq8_fp32_scan(MET[metric], st2, qb[0], dim)
    labels_ids = list(range(n))
    labels_ids[10] = n - 1          # the last row moved into the hole
    labels_ids = labels_ids[:n - 1]
    sc_by_id = np.array([sc2[lab] for lab in labels_ids])
    el, es = vso.topk_replay(sc_by_id, 30, np.array(labels_ids, dtype=np.uint64))
    l, d = ix.knn_query(q, 30)
    assert np.array_equal(l[0], el.astype(np.int64)) and np.array_equal(d[0], es)


@pytest.mark.parametrize("metric", ["L2", "IP", "Cosine"])
@pytest.mark.parametrize("dim", [5, 63, 64, 100, 768])
def test_sq8_symmetric_stored_distance(vso, metric, dim):
    """SQ8 x SQ8 between stored rows: scalar tier below dim 64, the VNNI tier (exact int32 dot, fused epilogue) from 64"""
    rng = np.random.default_rng(dim + len(metric))
    n = 40
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    ix = make(metric, dim)
    ix.add_vectors(rows, np.arange(n))
    st, _ = oracle_blobs(vso, rows, rows[:1], metric)
    for a, b in [(0, 1), (3, 3), (7, 39), (20, 5)]:
        want = vso.sq8_sq8_distance(MET[metric], st[a], st[b], dim)
        assert ix.stored_distance(a, b) == want, (metric, dim, a, b)
    assert np.isnan(ix.stored_distance(0, 10_000))


def test_sq8_stored_blob_is_the_preprocessors(vso):
    rng = np.random.default_rng(2)
    dim = 33
    rows = rng.uniform(-3, 3, (5, dim)).astype(np.float32)
    for metric in ("L2", "IP", "Cosine"):
        ix = make(metric, dim)
        ix.add_vectors(rows, np.arange(5))
        st, _ = oracle_blobs(vso, rows, rows[:1], metric)
        for i in range(5):
            got = ix.get_vector(i)
            assert got.shape == (1, st.shape[1]) and np.array_equal(got[0], st[i]), (metric, i)


@pytest.mark.parametrize("metric", ["L2", "IP", "Cosine"])
def test_sq8_mfma_filter_hard_inputs(vso, metric):
    """what the bound has to survive: rows of very different scale (delta from 1e-4 to 40), near-duplicates of the query
    (scores crowd around the threshold), queries with one dominant component (coarse int8 piece), k = 100"""
    rng = np.random.default_rng(77)
    dim, n, nq, k = 320, 30_000, 24, 100
    scale = np.exp(rng.uniform(np.log(1e-2), np.log(5e3), n)).astype(np.float32) if metric != "Cosine" else np.ones(n, np.float32)
    rows = (rng.uniform(-1, 1, (n, dim)) * scale[:, None]).astype(np.float32)
    q = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    q[::3, 5] *= 300.0                                     # a dominant component: every other Y_i rounds to 0 or +-1
    rows[1000:1400] = q[1] + rng.normal(0, 1e-3, (400, dim)).astype(np.float32)   # near-duplicates of one query
    ix = make(metric, dim)
    ix.add_vectors(rows, np.arange(n))
    ix.set_option("dense_pairs", 0)
    ix.reset_stats()
    labels, dists = ix.knn_query(q, k)
    assert ix.stats()["scan_kernel"] == "k_mfma_filter_lowp(sq8)"
    st, qb = oracle_blobs(vso, rows, q, metric)
    for j in range(nq):
        sc = vso.sq8_fp32_scan(MET[metric], st, qb[j], dim)
        el, es = vso.topk_replay(sc, k)
        assert np.array_equal(labels[j], el.astype(np.int64)), (metric, j)
        assert np.array_equal(dists[j], es), (metric, j)


@pytest.mark.parametrize("metric", ["L2", "IP"])
@pytest.mark.parametrize("nq", [64, 128])
@pytest.mark.parametrize("shape", ["offset", "uniform", "constant_rows"])
def test_sq8_screen_tight_cases(vso, metric, nq, shape):
    """the filter's screen (round 3: per value, 4 fused operations against a per-query slack from table-wide maxima; rounds 1-2:
    a block pre-screen) is tightest on homogeneous rows: vectors far from the origin with a small spread (min * y_sum dominates, delta tiny),
    plain uniform rows at a size where the k-th score is deep in the tail, and tiles holding constant vectors (delta = 1,
    all codes 0) next to ordinary ones; 4-wave (nq 64) and 8-wave (nq 128) kernels"""
    rng = np.random.default_rng(len(shape) + nq)
    dim, n, k = 96, 120_000, 10
    if shape == "offset":
        rows = (5.0 + rng.normal(0, 0.01, (n, dim))).astype(np.float32)
        q = (5.0 + rng.normal(0, 0.01, (nq, dim))).astype(np.float32)
    else:
        rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
        q = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
        if shape == "constant_rows":
            rows[::37] = rng.uniform(-2, 2, (len(rows[::37]), 1)).astype(np.float32)    # every component equal
            rows[5000:5064] = 0.25
    ix = make(metric, dim)
    ix.add_vectors(rows, np.arange(n))
    ix.set_option("dense_pairs", 0)
    ix.reset_stats()
    labels, dists = ix.knn_query(q, k)
    assert ix.stats()["scan_kernel"] == "k_mfma_filter_lowp(sq8)"
    st, qb = oracle_blobs(vso, rows, q, metric)
    for j in range(0, nq, 5):
        sc = vso.sq8_fp32_scan(MET[metric], st, qb[j], dim)
        el, es = vso.topk_replay(sc, k)
        assert np.array_equal(labels[j], el.astype(np.int64)), (metric, shape, j, labels[j], el)
        assert np.array_equal(dists[j], es), (metric, shape, j)


@pytest.mark.parametrize("metric", ["L2", "IP", "Cosine"])
@pytest.mark.parametrize("nq", [40, 128])
def test_sq8_screen_with_rows_of_very_different_scale(vso, metric, nq):
    """the screen's slack comes from TABLE-WIDE maxima of {delta, |min|, sum_squares}: a few rows hundreds of times larger
    than the rest (and, for L2 / IP, one with astronomically large values) blow it up for every row -- the screen then rejects
    little, the exact re-rank decides, the reply stays the reference's"""
    rng = np.random.default_rng(77 + nq)
    dim, n, k = 96, 50_000, 10
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    big = rng.choice(n, 400, replace=False)
    rows[big] *= rng.uniform(100, 800, (400, 1)).astype(np.float32)
    if metric != "Cosine":
        rows[big[0]] = rng.uniform(-1, 1, dim).astype(np.float32) * np.float32(1e18)
    q = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    q[::7] *= np.float32(300)
    ix = make(metric, dim)
    ix.add_vectors(rows, np.arange(n))
    ix.set_option("dense_pairs", 0)
    ix.reset_stats()
    labels, dists = ix.knn_query(q, k)
    assert ix.stats()["scan_kernel"] == "k_mfma_filter_lowp(sq8)"
    st, qb = oracle_blobs(vso, rows, q, metric)
    for j in range(0, nq, 3):
        sc = vso.sq8_fp32_scan(MET[metric], st, qb[j], dim)
        el, es = vso.topk_replay(sc, k)
        assert np.array_equal(labels[j], el.astype(np.int64)), (metric, j, labels[j], el)
        assert np.array_equal(dists[j], es, equal_nan=True), (metric, j)


# ---------------------------------------------------------------- fp16 vectors / queries (QuantPreprocessor<float16>, SQ8_FP16_*)
def make_f16(metric, dim):
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT16, dim, MET[metric]
    return VecSim.SQ8Index(p)


def oracle_blobs_f16(vso, rows, queries, metric):
    m = MET[metric]
    rows = np.array(rows, dtype=np.uint16, copy=True)
    queries = np.array(queries, dtype=np.uint16, copy=True)
    if metric == "Cosine":
        for v in rows:
            vso.normalize(v, v.size, vso.F16)
        for v in queries:
            vso.normalize(v, v.size, vso.F16)
    st = np.stack([vso.sq8_quantize_f16(v, m) for v in rows])
    qb = np.stack([vso.sq8_query_blob_f16(v, m) for v in queries])
    return st, qb


@pytest.mark.parametrize("metric", ["L2", "IP", "Cosine"])
@pytest.mark.parametrize("dim", [5, 15, 16, 17, 48, 63, 64, 100, 129])
def test_sq8_fp16_all_scores_bit_exact(vso, metric, dim):
    rng = np.random.default_rng(dim * 3 + len(metric))
    n = 300
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float16).view(np.uint16)
    q = rng.uniform(-1, 1, (3, dim)).astype(np.float16).view(np.uint16)
    ix = make_f16(metric, dim)
    ix.add_vectors(rows, np.arange(n))
    labels, dists = ix.knn_query(q, n)
    st, qb = oracle_blobs_f16(vso, rows, q, metric)
    for j in range(3):
        sc = vso.sq8_fp16_scan(MET[metric], st, qb[j], dim)
        el, es = vso.topk_replay(sc, n)
        assert np.array_equal(labels[j], el.astype(np.int64)), (metric, dim, j)
        assert np.array_equal(dists[j], es), (metric, dim, j)
        assert np.array_equal(ix.get_vector(j)[0], st[j])


@pytest.mark.parametrize("metric,dim,n,nq,k", [("L2", 768, 20_000, 40, 10), ("Cosine", 200, 30_000, 9, 20)])
def test_sq8_fp16_filtered_path(vso, metric, dim, n, nq, k):
    rng = np.random.default_rng(dim + n)
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float16).view(np.uint16)
    q = rng.uniform(-1, 1, (nq, dim)).astype(np.float16).view(np.uint16)
    ix = make_f16(metric, dim)
    ix.add_vectors(rows, np.arange(n))
    ix.set_option("dense_pairs", 0)
    ix.reset_stats()
    labels, dists = ix.knn_query(q, k)
    assert ix.stats()["scan_kernel"] == "k_mfma_filter_lowp(sq8)"
    st, qb = oracle_blobs_f16(vso, rows, q, metric)
    for j in range(0, nq, 3):
        sc = vso.sq8_fp16_scan(MET[metric], st, qb[j], dim)
        el, es = vso.topk_replay(sc, k)
        assert np.array_equal(labels[j], el.astype(np.int64)) and np.array_equal(dists[j], es), (metric, j)
