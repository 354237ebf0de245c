"""GPU: mean-centred SQ8 (VecSimGpu_NewFlatSQ8Centered) against oracle/vso_sq8.c's restatement of
QuantPreprocessor<..., WithNorm = true> (preprocessors.h:484-495, 574-640) and DistanceCalculatorWithNorm (calculator.h:126-232),
through the C API.  Stored bytes, labels, order and scores bit-exact (0 ulp); fp32 and fp16 inputs; exact path and MFMA filter."""
import numpy as np
import pytest

from vectorsimilarity_amd import VecSim

pytestmark = pytest.mark.gpu
MET = {"L2": 0, "IP": 1}


def make(metric, dim, mean, f16=False):
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = (VecSim.VecSimType_FLOAT16 if f16 else VecSim.VecSimType_FLOAT32), dim, MET[metric]
    return VecSim.SQ8Index(p, mean=mean)


def mss_of(mean):
    acc = np.float32(0)
    for m in mean:
        acc = np.float32(acc + np.float32(m * m))
    return float(acc)


def data(vso, rng, n, nq, dim, f16, offset=0.0):
    mean = (rng.uniform(-0.5, 0.5, dim) + offset).astype(np.float32)
    rows = (rng.uniform(-1, 1, (n, dim)) + mean).astype(np.float32)
    q = (rng.uniform(-1, 1, (nq, dim)) + mean).astype(np.float32)
    if f16:
        rows, q = rows.astype(np.float16).view(np.uint16), q.astype(np.float16).view(np.uint16)
    return mean, rows, q


def oracle_scores(vso, rows, mean, query, metric, dim, f16):
    m = MET[metric]
    st = np.stack([vso.sq8_quantize_norm(v, mean, m, f16=f16) for v in rows])
    qb = vso.sq8_query_blob_norm(query, mean, m, f16=f16)
    return st, vso.sq8_scan_norm(m, st, qb, dim, f16=f16)


@pytest.mark.parametrize("f16", [False, True])
@pytest.mark.parametrize("metric", ["L2", "IP"])
@pytest.mark.parametrize("dim", [5, 8, 17, 64, 100])
def test_centred_all_scores_and_blobs_bit_exact(vso, metric, dim, f16):
    rng = np.random.default_rng(dim * 3 + len(metric) + f16)
    n = 300
    mean, rows, q = data(vso, rng, n, 3, dim, f16)
    ix = make(metric, dim, mean, f16)
    for i in range(n):
        ix.add_vector(rows[i], i)
    labels, dists = ix.knn_query(q, n)
    for j in range(3):
        st, sc = oracle_scores(vso, rows, mean, q[j], metric, dim, f16)
        el, es = vso.topk_replay(sc, n)
        assert np.array_equal(labels[j], el.astype(np.int64)), (metric, dim, f16, j)
        assert np.array_equal(dists[j], es), (metric, dim, f16, j)
    for i in (0, 7, n - 1):
        got = ix.get_vector(i)
        assert got.shape == (1, dim + 16) and np.array_equal(got[0], st[i]), (metric, dim, i)
    assert ix.get_distance_from(5, q[2]) == sc[5]
    # symmetric: calcDistance between stored blobs (IP: base - x_mean_ip - y_mean_ip + mean_sum_squares)
    for a, b in [(0, 1), (3, 3), (7, n - 1)]:
        want = vso.sq8_sq8_distance_norm(MET[metric], st[a], st[b], dim, mss_of(mean))
        assert ix.stored_distance(a, b) == want, (metric, dim, a, b)


@pytest.mark.parametrize("metric,dim,n,nq,k,f16", [("IP", 128, 30_000, 64, 10, False), ("L2", 128, 30_000, 20, 10, False),
                                                    ("IP", 768, 20_000, 40, 10, False), ("IP", 200, 30_000, 9, 20, True),
                                                    ("IP", 96, 40_000, 33, 100, False)])
def test_centred_filtered_path(vso, metric, dim, n, nq, k, f16):
    """wide batches go through the int8 MFMA filter: its bound must hold for the shifted IP score (base - y_mean_ip), with
    means far from zero (large |y_mean_ip| against O(1) spreads) included"""
    rng = np.random.default_rng(n + dim + k)
    mean, rows, q = data(vso, rng, n, nq, dim, f16, offset=3.0 if dim == 96 else 0.0)
    ix = make(metric, dim, mean, f16)
    ix.add_vectors(rows, np.arange(n))
    ix.set_option("dense_pairs", 0)
    ix.reset_stats()
    labels, dists = ix.knn_query(q, k)
    assert ix.stats()["scan_kernel"].startswith("k_mfma_filter_lowp"), ix.stats()["scan_kernel"]
    for j in range(nq):
        _, sc = oracle_scores(vso, rows, mean, q[j], metric, dim, f16)
        el, es = vso.topk_replay(sc, k)
        assert np.array_equal(labels[j], el.astype(np.int64)), (metric, dim, j, labels[j], el)
        assert np.array_equal(dists[j], es), (metric, dim, j)
    # range query on the same scores
    sc0 = oracle_scores(vso, rows, mean, q[0], metric, dim, f16)[1]
    radius = float(np.sort(sc0)[25])
    if radius >= 0:                   # (a negative radius throws, as upstream: vec_sim.cpp:362-367)
        l, d = ix.range_query(q[0], radius)
        want = np.nonzero(sc0 <= radius)[0]
        assert sorted(l[0].tolist()) == want.tolist()


def test_centred_rejects_cosine_and_zero_mean_equals_plain(vso):
    dim = 48
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, dim, VecSim.VecSimMetric_Cosine
    with pytest.raises(RuntimeError):
        VecSim.SQ8Index(p, mean=np.zeros(dim, dtype=np.float32))
    rng = np.random.default_rng(9)
    rows = rng.uniform(-1, 1, (200, dim)).astype(np.float32)
    q = rng.uniform(-1, 1, (4, dim)).astype(np.float32)
    for metric in ("L2", "IP"):       # test_components.cpp:2369-2415: a zero mean gives the base kernels' results
        p.metric = MET[metric]
        a, b = VecSim.SQ8Index(p, mean=np.zeros(dim, dtype=np.float32)), VecSim.SQ8Index(p)
        a.add_vectors(rows, np.arange(200))
        b.add_vectors(rows, np.arange(200))
        la, da = a.knn_query(q, 200)
        lb, db = b.knn_query(q, 200)
        assert np.array_equal(la, lb) and np.array_equal(da, db)
        assert a.stored_distance(3, 9) == b.stored_distance(3, 9)
