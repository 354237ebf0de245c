"""GPU: HNSW query loops.  (1) the GPU search equals the CPU restatement of the reference's search loops
(oracle/vso_hnsw.c) on the SAME graph -- labels, order, scores and even the number of distance
evaluations, bit for bit; (2) recall@10 against the exact Flat answer, as the reference's own flow tests
measure it (tests/flow/test_hnsw.py:114-117: recall > 0.9)."""
import numpy as np
import pytest

from vectorsimilarity_amd import VecSim

pytestmark = pytest.mark.gpu


def build(dim, n, metric, M=16, efc=100, ef=50, seed=3, rows=None, labels=None):
    rng = np.random.default_rng(seed)
    if rows is None:
        rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    p = VecSim.HNSWParams()
    p.type, p.dim, p.metric, p.M, p.efConstruction, p.efRuntime = VecSim.VecSimType_FLOAT32, dim, metric, M, efc, ef
    ix = VecSim.HNSWIndex(p)
    labels = np.arange(n) if labels is None else labels
    ix.add_vectors(rows, labels)
    return ix, rows


def stored(vso, rows, metric):
    if metric != VecSim.VecSimMetric_Cosine:
        return rows
    out = rows.copy()
    for i in range(len(out)):
        vso.normalize(out[i], out.shape[1], 0)
    return out


@pytest.mark.parametrize("metric,dim,n,M,ef,k", [
    (VecSim.VecSimMetric_L2, 32, 3000, 16, 50, 10),
    (VecSim.VecSimMetric_L2, 128, 5000, 16, 128, 10),
    (VecSim.VecSimMetric_IP, 64, 3000, 8, 40, 5),
    (VecSim.VecSimMetric_Cosine, 100, 3000, 12, 64, 20),
    (VecSim.VecSimMetric_L2, 20, 2000, 4, 10, 10),
    (VecSim.VecSimMetric_L2, 768, 30_000, 16, 128, 10),   # BASELINE config 5's shape (efC 200 below), 24 row chunks per neighbour
    (VecSim.VecSimMetric_Cosine, 768, 8_000, 16, 128, 10),
])
def test_gpu_search_equals_reference_loops_on_same_graph(vso, metric, dim, n, M, ef, k):
    ix, rows = build(dim, n, metric, M=M, efc=200 if dim == 768 else 80, ef=ef)
    g = ix.graph()
    assert g["n"] == n and g["max_level"] >= 1 and g["cnt0"].max() <= 2 * M
    rng = np.random.default_rng(99)
    q = rng.uniform(-1, 1, (64 if dim == 768 else 40, dim)).astype(np.float32)
    labels, dists = ix.knn_query(q, k)
    evals = ix.last_distance_evals()
    srows = stored(vso, rows, metric)
    sq = stored(vso, q, metric)
    km = 0 if metric == VecSim.VecSimMetric_L2 else 1
    total = 0
    for j in range(len(q)):
        el, es, ev = vso.hnsw_search(0, km, srows, g, sq[j], k, ef, dim)
        total += ev
        assert np.array_equal(labels[j][:len(el)], el.astype(np.int64)), (j, labels[j], el)
        assert np.array_equal(dists[j][:len(es)], es), j
        assert np.all(labels[j][len(el):] == -1)
    # the oracle re-evaluates dist(entry point) once per query at level 0 (hnsw.h:1997); the GPU reuses it
    assert evals == total - len(q)


def test_recall_against_flat(vso):
    dim, n, k = 64, 20000, 10
    ix, rows = build(dim, n, VecSim.VecSimMetric_L2, M=16, efc=200, ef=128)
    rng = np.random.default_rng(5)
    q = rng.uniform(-1, 1, (200, dim)).astype(np.float32)
    labels, _ = ix.knn_query(q, k)
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, dim, VecSim.VecSimMetric_L2
    bf = VecSim.BFIndex(p)
    bf.add_vectors(rows, np.arange(n))
    exact, _ = bf.knn_query(q, k)
    hits = sum(len(set(labels[i]) & set(exact[i])) for i in range(len(q)))
    recall = hits / (len(q) * k)
    assert recall > 0.9, recall
    # a larger ef at query time may only help
    ix.set_ef(400)
    l2, _ = ix.knn_query(q, k)
    assert sum(len(set(l2[i]) & set(exact[i])) for i in range(len(q))) >= hits


def test_deleted_and_overwritten_vectors_are_traversed_not_returned(vso):
    dim, n, k = 24, 2500, 10
    rng = np.random.default_rng(8)
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    ix, _ = build(dim, n, VecSim.VecSimMetric_L2, M=8, efc=60, ef=40, rows=rows, labels=np.arange(n) + 5)
    gone = set(int(x) + 5 for x in rng.choice(n, 300, replace=False))
    for lab in gone:
        assert ix.delete_vector(lab) == 1
    assert ix.delete_vector(10 ** 9) == 0
    assert ix.index_size() == n - 300
    q = rng.uniform(-1, 1, (30, dim)).astype(np.float32)
    labels, dists = ix.knn_query(q, k)
    assert not (set(labels.ravel().tolist()) & gone)
    g = ix.graph()
    # (300 of 2500 deletes cross the compaction threshold: dead nodes were removed in one batch and the last live nodes moved into
    # the holes, so the graph's internal order is no longer the insertion order -- the oracle gets the rows in the graph's order)
    assert g["n"] < n and g["n"] >= n - 300
    grows = rows[(g["labels"].astype(np.int64) - 5)]
    for j in range(len(q)):
        el, es, _ = vso.hnsw_search(0, 0, grows, g, q[j], k, 40, dim)
        assert np.array_equal(labels[j][:len(el)], el.astype(np.int64)) and np.array_equal(dists[j][:len(es)], es)
    # overwrite: the old vector is retired, the new one is reachable under the same label
    v = rng.uniform(-1, 1, dim).astype(np.float32)
    keep = next(l for l in range(5, n + 5) if l not in gone)
    assert ix.add_vector(v, keep) == 0
    l, d = ix.knn_query(v, 1)
    assert l[0, 0] == keep and d[0, 0] == 0.0


@pytest.mark.parametrize("multi", [False, True])
def test_deleted_nodes_are_removed_in_batches_and_the_graph_stays_searchable(vso, multi):
    """round-4 review: deletes were mark-only, dead nodes accumulated and were traversed for ever.  Now a sixteenth of dead nodes
    triggers a compaction (csrc/host/hnsw_index.cpp compactDeleted; the reference repairs on every delete, hnsw.h:1796-1852): every
    live node that pointed at a dead one gets its list rebuilt from its own and the dead ones' live neighbours, the entry point is
    replaced if it died, the last live nodes move into the holes.  Checked: the exported graph holds live nodes only (none flagged,
    every link in range, no node links to itself), labels and stored vectors survive the renumbering, the GPU search equals the
    reference's loops on that graph, recall against the exact answer over the live vectors stays where a fresh build puts it,
    evaluations per query do not grow, and the index keeps taking adds, overwrites and deletes afterwards."""
    dim, n, k, ef = 32, 6000, 10, 64
    rng = np.random.default_rng(31)
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    labels = (np.arange(n) // 2 if multi else np.arange(n)) + 100
    p = VecSim.HNSWParams()
    p.type, p.dim, p.metric, p.M, p.efConstruction, p.efRuntime, p.multi = VecSim.VecSimType_FLOAT32, dim, VecSim.VecSimMetric_L2, 12, 80, ef, multi
    ix = VecSim.HNSWIndex(p)
    ix.add_vectors(rows, labels)
    q = rng.uniform(-1, 1, (50, dim)).astype(np.float32)
    ix.knn_query(q, k)
    evals_before = ix.last_distance_evals()
    alive = np.ones(n, bool)
    dead_labels = rng.choice(np.unique(labels), (len(np.unique(labels)) * 2) // 5, replace=False)   # 40 % of the labels go
    for lab in dead_labels:
        assert ix.delete_vector(int(lab)) == (2 if multi else 1)
        alive[labels == lab] = False
    live_n = int(alive.sum())
    assert ix.index_size() == live_n
    g = ix.graph()
    # compactions ran: at most a sixteenth of the graph is dead nodes waiting for the next one
    assert live_n <= g["n"] <= live_n + live_n // 15 + 32, (g["n"], live_n)
    cnt0, links0 = g["cnt0"].astype(int), g["links0"]
    for i in range(g["n"]):
        li = links0[i, :cnt0[i]]
        assert np.all(li < g["n"]) and i not in li and len(set(li.tolist())) == len(li)
    gl = g["labels"].astype(np.int64)
    live_ids = np.nonzero(g["deleted"] == 0)[0]
    assert sorted(gl[live_ids].tolist()) == sorted(labels[alive].tolist())
    if multi:   # a label's two vectors survive the renumbering
        for lab in np.unique(gl[live_ids])[::97]:
            vs = ix.get_vector(int(lab))
            want = rows[labels == lab]
            assert len(vs) == len(want) and sorted(map(bytes, vs)) == sorted(map(bytes, want))
    else:
        grows = rows[gl - 100]   # rows in the graph's order, for the oracle below
        for lab in (int(gl[live_ids[0]]), int(gl[live_ids[-1]]), int(gl[live_ids[len(live_ids) // 2]])):
            assert np.array_equal(ix.get_vector(lab)[0], rows[lab - 100])
    got_l, got_d = ix.knn_query(q, k)
    evals_after = ix.last_distance_evals()
    assert not (set(got_l.ravel().tolist()) & set(int(x) for x in dead_labels))
    if not multi:
        for j in range(0, len(q), 5):
            el, es, _ = vso.hnsw_search(0, 0, grows, g, q[j], k, ef, dim)
            assert np.array_equal(got_l[j][:len(el)], el.astype(np.int64)) and np.array_equal(got_d[j][:len(es)], es), j
    # recall against the exact answer over the live vectors
    bp = VecSim.BFParams()
    bp.type, bp.dim, bp.metric, bp.multi = VecSim.VecSimType_FLOAT32, dim, VecSim.VecSimMetric_L2, multi
    bf = VecSim.BFIndex(bp)
    bf.add_vectors(rows[alive], labels[alive])
    exact, _ = bf.knn_query(q, k)
    recall = sum(len(set(got_l[i]) & set(exact[i])) for i in range(len(q))) / (len(q) * k)
    assert recall > 0.9, recall
    assert evals_after <= evals_before * 1.05, (evals_before, evals_after)   # no dead weight in the walk
    # the index goes on: new vectors, overwrites (each leaves a dead node behind), more deletes
    extra = rng.uniform(-1, 1, (500, dim)).astype(np.float32)
    ix.add_vectors(extra, np.arange(500) + 10 ** 6)
    bf.add_vectors(extra, np.arange(500) + 10 ** 6)
    if not multi:
        for i in range(0, 500, 2):
            v = rng.uniform(-1, 1, dim).astype(np.float32)
            assert ix.add_vector(v, 10 ** 6 + i) == 0
            bf.add_vector(v, 10 ** 6 + i)
    for i in range(1, 500, 5):
        assert ix.delete_vector(10 ** 6 + i) == bf.delete_vector(10 ** 6 + i) == 1
    assert ix.index_size() == bf.index_size()
    got_l, _ = ix.knn_query(q, k)
    exact, _ = bf.knn_query(q, k)
    recall = sum(len(set(got_l[i]) & set(exact[i])) for i in range(len(q))) / (len(q) * k)
    assert recall > 0.9, recall
    l1, d1 = ix.knn_query(extra[3], 1)   # (label 10^6 + 3: neither overwritten nor deleted above)
    assert l1[0, 0] == 10 ** 6 + 3 and d1[0, 0] == 0.0


def test_hnsw_edge_cases():
    p = VecSim.HNSWParams()
    p.type, p.dim, p.metric, p.M = VecSim.VecSimType_FLOAT32, 8, VecSim.VecSimMetric_L2, 4
    ix = VecSim.HNSWIndex(p)
    l, d = ix.knn_query(np.zeros(8, dtype=np.float32), 3)
    assert np.all(l == -1)
    ix.add_vector(np.ones(8, dtype=np.float32), 7)
    l, d = ix.knn_query(np.zeros((2, 8), dtype=np.float32), 3)
    assert list(l[0]) == [7, -1, -1] and d[0, 0] == 8.0
    qp = VecSim.VecSimQueryParams()
    qp.hnswRuntimeParams.efRuntime = 50
    l, d = ix.knn_query(np.zeros(8, dtype=np.float32), 1, qp)
    assert l[0, 0] == 7


def test_hnsw_debug_info_iterator_fields():
    """HNSWIndex::debugInfoIterator (hnsw.h:2216-2273): common fields, BLOCK_SIZE, then the HNSW block"""
    p = VecSim.HNSWParams()
    p.type, p.dim, p.metric, p.M, p.efConstruction, p.efRuntime = VecSim.VecSimType_FLOAT32, 16, VecSim.VecSimMetric_L2, 8, 40, 17
    ix = VecSim.HNSWIndex(p)
    ix.add_vectors(np.random.default_rng(1).uniform(-1, 1, (300, 16)).astype(np.float32), np.arange(300))
    f = ix.debug_info_fields()
    names = [n for n, _ in f]
    assert names[0] == "ALGORITHM" and names[10] == "BLOCK_SIZE"
    assert names[11:] == ["M", "EF_CONSTRUCTION", "EF_RUNTIME", "MAX_LEVEL", "ENTRYPOINT", "EPSILON", "NUMBER_OF_MARKED_DELETED"]   # exactly the reference's fields
    d = dict(f)
    assert d["ALGORITHM"] == "HNSW" and d["M"] == 8 and d["EF_CONSTRUCTION"] == 40 and d["EF_RUNTIME"] == 17
    assert d["INDEX_SIZE"] == 300 and d["NUMBER_OF_MARKED_DELETED"] == 0


def test_hnsw_python_surface_parity_helpers():
    """add_vector_parallel / range_parallel / check_integrity / get_vector keep the reference binding's shapes"""
    p = VecSim.HNSWParams()
    p.type, p.dim, p.metric, p.M, p.efConstruction, p.efRuntime = VecSim.VecSimType_FLOAT32, 24, VecSim.VecSimMetric_L2, 8, 60, 50
    ix = VecSim.HNSWIndex(p)
    rows = np.random.default_rng(5).uniform(-1, 1, (3000, 24)).astype(np.float32)
    ix.add_vector_parallel(rows, np.arange(3000))
    assert ix.index_size() == 3000 and ix.check_integrity()
    assert np.array_equal(ix.get_vector(77)[0], rows[77])
    l, d = ix.knn_parallel(rows[:5], 3)
    assert list(l[:, 0]) == [0, 1, 2, 3, 4] and np.all(d[:, 0] == 0)


@pytest.mark.parametrize("metric,dim,n,M,eps", [
    (VecSim.VecSimMetric_L2, 32, 4000, 16, 0.01),
    (VecSim.VecSimMetric_L2, 64, 3000, 8, 0.1),
    (VecSim.VecSimMetric_Cosine, 48, 3000, 12, 0.01),
    (VecSim.VecSimMetric_IP, 40, 2500, 16, 0.05),
])
def test_gpu_range_search_equals_reference_loops_on_same_graph(vso, metric, dim, n, M, eps):
    """searchRangeBottomLayer_WithTimeout + processCandidate_RangeSearch (hnsw.h:616-680, 2087-2187): same
    results, same scores and the same number of distance evaluations as the restated loops on the same graph"""
    ix, rows = build(dim, n, metric, M=M, efc=80, ef=50)
    g = ix.graph()
    rng = np.random.default_rng(17)
    q = rng.uniform(-1, 1, (12, dim)).astype(np.float32)
    srows = stored(vso, rows, metric)
    sq = stored(vso, q, metric)
    km = 0 if metric == VecSim.VecSimMetric_L2 else 1
    qp = VecSim.VecSimQueryParams()
    qp.hnswRuntimeParams.epsilon = eps
    for j in range(len(q)):
        sc = vso.scan(0, km, srows, sq[j], dim)
        # (VecSimIndex_RangeQuery rejects negative radii, as upstream: IP scores of unnormalised rows can be negative)
        for radius in sorted({max(float(np.sort(sc)[i]), 0.0) for i in (5, 60, 400)} | {max(float(np.sort(sc)[0]) - 1.0, 0.0)}):
            el, es, ev = vso.hnsw_range(0, km, srows, g, sq[j], radius, eps, dim)
            labels, dists = ix.range_query(q[j], radius, qp, order=VecSim.BY_ID)
            order = np.argsort(el.astype(np.int64), kind="stable")
            assert labels.shape[1] == len(el), (j, radius, labels.shape, len(el))
            assert np.array_equal(labels[0], el.astype(np.int64)[order])
            assert np.array_equal(dists[0], es[order])
            # the GPU reuses dist(entry point) instead of recomputing it at level 0
            assert ix.last_distance_evals() == ev - 1
            # everything returned is inside the radius, and recall against the exact scan is high
            assert np.all(dists[0] <= np.float32(radius))
    labels, dists = ix.range_query(q[0], max(float(np.sort(vso.scan(0, km, srows, sq[0], dim))[50]), 0.0), qp)
    assert np.all(np.diff(dists[0]) >= 0)   # BY_SCORE


def test_range_search_with_deleted_nodes_and_wide_radius(vso):
    dim, n = 24, 3000
    ix, rows = build(dim, n, VecSim.VecSimMetric_L2, M=8, efc=60, ef=40)
    for lab in range(0, n, 7):
        ix.delete_vector(lab)
    g = ix.graph()
    grows = rows[g["labels"].astype(np.int64)]   # (the deletes crossed the compaction threshold: rows in the graph's order)
    q = np.random.default_rng(2).uniform(-1, 1, dim).astype(np.float32)
    sc = vso.scan(0, 0, rows, q, dim)
    radius = float(np.sort(sc)[300])
    el, es, _ = vso.hnsw_range(0, 0, grows, g, q, radius, 0.01, dim)
    labels, dists = ix.range_query(q, radius, order=VecSim.BY_ID)
    assert sorted(labels[0].tolist()) == sorted(el.astype(np.int64).tolist())
    assert not any(l % 7 == 0 for l in labels[0].tolist())
    # a radius covering the whole index: still the graph walk's answer while its candidate window fits in LDS,
    # the exact table scan (every live vector) once it does not
    wide, _ = ix.range_query(q, float(sc.max()) + 1.0, order=VecSim.BY_ID)
    wl, _, _ = vso.hnsw_range(0, 0, grows, g, q, float(sc.max()) + 1.0, 0.01, dim)
    alive = [i for i in range(n) if i % 7 != 0]
    assert wide[0].tolist() in (sorted(wl.astype(np.int64).tolist()), alive)


def test_hnsw_batch_iterator_hands_out_exact_batches(vso, monkeypatch):
    """VECSIM_HNSW_ITER_EXACT=1 (rounds 1-3's iterator): every batch is the exact next-best set (same machinery and the same
    GPU score pass as the Flat iterator), deleted vectors never appear"""
    monkeypatch.setenv("VECSIM_HNSW_ITER_EXACT", "1")
    dim, n = 32, 2500
    ix, rows = build(dim, n, VecSim.VecSimMetric_L2, M=8, efc=60, ef=40)
    for lab in (5, 77, 1200):
        ix.delete_vector(lab)
    q = np.random.default_rng(8).uniform(-1, 1, dim).astype(np.float32)
    sc = vso.scan(0, 0, rows, q, dim)
    sc[[5, 77, 1200]] = np.inf
    order = np.lexsort((np.arange(n), sc))
    it = ix.create_batch_iterator(q)
    got = []
    while it.has_next() and len(got) < 300:
        l, d = it.get_next_results(100, VecSim.BY_SCORE)
        assert np.all(np.diff(d[0]) >= 0)
        got += l[0].tolist()
    assert got == order[:300].tolist()
    it.reset()
    l, d = it.get_next_results(10, VecSim.BY_SCORE)
    assert l[0].tolist() == order[:10].tolist() and np.array_equal(d[0], sc[order[:10]])


def test_hnsw_prefer_adhoc_follows_reference_tree():
    """spot checks of the decision tree (hnsw.h:2275-2408): (index size, subset ratio, k, dim, M) -> ad-hoc?"""
    def mk(n, dim, M):
        ix, _ = build(dim, n, VecSim.VecSimMetric_L2, M=M, efc=20, ef=10)
        return ix
    small = mk(3000, 16, 8)
    assert small.prefer_adhoc(3000, 10, True) is True                     # node 1: index_size <= 5500
    mid_lo = mk(6000, 32, 16)
    assert mid_lo.prefer_adhoc(600, 10, True) is True                     # r = 0.1 <= 0.17
    assert mid_lo.prefer_adhoc(3000, 10, True) is False                   # r = 0.5, k <= 12, d <= 55
    assert mid_lo.prefer_adhoc(3000, 20, True) is True                    # k > 12
    mid_hi = mk(6000, 64, 16)
    assert mid_hi.prefer_adhoc(3000, 10, True) is True                    # d > 55, M > 10
    assert mk(6000, 64, 8).prefer_adhoc(3000, 10, False) is False         # d > 55, M <= 10


@pytest.mark.parametrize("typ,metric,dim", [("bf16", "L2", 64), ("bf16", "Cosine", 96), ("f16", "IP", 48),
                                            ("i8", "L2", 64), ("u8", "IP", 32), ("i8", "Cosine", 64), ("u8", "Cosine", 40)])
def test_typed_hnsw_search_equals_reference_loops(vso, typ, metric, dim):
    """HNSW over the other stored types: the graph is built from widened copies (host, ingest side), the GPU
    search scores the stored blobs with that type's reference-order kernel and equals the restated loops"""
    from util import METRICS, TYPES, random_vectors, stored_rows
    n, k, ef = 2500, 8, 40
    rng = np.random.default_rng(dim)
    rows = random_vectors(rng, n, dim, typ, vso)
    q = random_vectors(rng, 25, dim, typ, vso)
    p = VecSim.HNSWParams()
    p.type, p.dim, p.metric, p.M, p.efConstruction, p.efRuntime = TYPES[typ], dim, METRICS[metric], 12, 80, ef
    ix = VecSim.HNSWIndex(p)
    ix.add_vectors(rows, np.arange(n))
    g = ix.graph()
    assert g["n"] == n and ix.check_integrity()
    labels, dists = ix.knn_query(q, k)
    srows = stored_rows(vso, rows, typ, metric)
    sq = stored_rows(vso, q, typ, metric)
    km = METRICS["IP"] if (metric == "Cosine" and typ not in ("i8", "u8")) else METRICS[metric]
    for j in range(len(q)):
        el, es, _ = vso.hnsw_search(TYPES[typ], km, srows, g, sq[j], k, ef, dim)
        assert np.array_equal(labels[j][:len(el)], el.astype(np.int64)), (typ, j)
        assert np.array_equal(dists[j][:len(es)], es), (typ, j)
    # exact recall sanity against the Flat index of the same type
    bp = VecSim.BFParams()
    bp.type, bp.dim, bp.metric = TYPES[typ], dim, METRICS[metric]
    bf = VecSim.BFIndex(bp)
    bf.add_vectors(rows, np.arange(n))
    exact, _ = bf.knn_query(q, k)
    hits = sum(len(set(labels[i]) & set(exact[i])) for i in range(len(q)))
    assert hits / (len(q) * k) > 0.6
    assert np.array_equal(ix.get_vector(5), bf.get_vector(5))


def test_hnsw_batch_iterator_sparse_mode_with_deleted_nodes(vso, monkeypatch):
    """(exact iterator) enough live nodes for the heap regime (device-resident scores): batches still exact, deleted nodes absent"""
    monkeypatch.setenv("VECSIM_HNSW_ITER_EXACT", "1")
    dim, n = 16, 120_000
    rng = np.random.default_rng(4)
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    ix, _ = build(dim, n, VecSim.VecSimMetric_L2, M=4, efc=10, ef=10, rows=rows)
    dead = list(range(0, n, 1001))
    for lab in dead:
        ix.delete_vector(lab)
    q = rng.uniform(-1, 1, dim).astype(np.float32)
    sc = vso.scan(0, 0, rows, q, dim)
    sc[dead] = np.inf
    order = np.lexsort((np.arange(n), sc))
    it = ix.create_batch_iterator(q)
    got = []
    for m in (20, 50, 30, 100):
        l, d = it.get_next_results(m, VecSim.BY_SCORE)
        got += l[0].tolist()
    assert got == order[:200].tolist()


def test_topk_beyond_the_lds_heaps_and_beyond_the_index(vso):
    """the reference answers any k (hnsw.h:2037-2084).  (1) k larger than the index: every live node comes back, like
    ef = live; (2) ef past what the kernel's per-query LDS heaps hold (about 1.3 K at dim 768): the batch is answered by
    the exact GPU scan -- the true k best live vectors -- and the reply is OK, not TimedOut; deleted nodes never appear"""
    dim, n = 768, 6000
    ix, rows = build(dim, n, VecSim.VecSimMetric_L2, M=8, efc=40, ef=20)
    for lab in (5, 77, 4000):
        ix.delete_vector(lab)
    q = np.random.default_rng(4).uniform(-1, 1, (3, dim)).astype(np.float32)
    k = 3000
    labels, dists, code = None, None, None
    l1, d1, code = ix.knn_query_code(q[0], k)
    assert code == VecSim._capi.VecSim_QueryReply_OK
    keep = np.ones(n, bool)
    keep[[5, 77, 4000]] = False
    for j in range(3):
        labels, dists = ix.knn_query(q[j], k)
        el, es = vso.flat_topk(0, 0, rows[keep], q[j], k, dim, np.nonzero(keep)[0].astype(np.uint64))
        assert np.array_equal(labels[0], el.astype(np.int64)) and np.array_equal(dists[0], es), j
    small, rows2 = build(16, 50, VecSim.VecSimMetric_L2, M=4, efc=20, ef=10)
    small.delete_vector(7)
    labels, dists = small.knn_query(rows2[0], 200)
    got = labels[0][labels[0] >= 0]
    assert len(got) == 49 and 7 not in got and np.all(np.diff(dists[0][:49]) >= 0) and np.all(labels[0][49:] == -1)


# ---------------------------------------------------------------- the reference's deterministic HNSW tests
def _kats():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "kat_hnsw.json")) as f:
        return json.load(f)


MET = {"L2": VecSim.VecSimMetric_L2, "IP": VecSim.VecSimMetric_IP, "Cosine": VecSim.VecSimMetric_Cosine}


@pytest.mark.parametrize("case", _kats()["topk"], ids=lambda c: c["name"])
def test_reference_hnsw_known_answers(vso, case):
    """tests/unit/test_hnsw.cpp closed forms (tests/golden/kat_hnsw.json), asserted twice: on the product (host-built graph,
    GPU search, through the C API) and on oracle/vso_hnsw.c searching the same graph -- which pins the oracle's search loops on
    answers the reference's own tests hold."""
    rows = np.array(case["vectors"], dtype=np.float32)
    labels = np.array(case["labels"])
    p = VecSim.HNSWParams()
    p.type, p.dim, p.metric, p.M, p.efConstruction = VecSim.VecSimType_FLOAT32, case["dim"], MET[case["metric"]], case["M"], case["efConstruction"]
    ix = VecSim.HNSWIndex(p)
    for v, lab in zip(rows, labels):
        ix.add_vector(v, int(lab))
    assert ix.index_size() == len(rows)
    q = np.array(case["query"], dtype=np.float32)
    k = case["k"]
    order = VecSim.BY_ID if case["order"] == "id" else VecSim.BY_SCORE
    gl, gd = ix.knn_query(q[None, :], k, order=order)
    assert ix.knn_query(q[None, :], 0)[0].shape[1] == 0          # "search for nothing" (test_hnsw.cpp:246)
    g = ix.graph()
    srows = stored(vso, rows, MET[case["metric"]])
    sq = stored(vso, q[None, :].copy(), MET[case["metric"]])[0]
    km = 0 if case["metric"] == "L2" else 1
    ol, od, _ = vso.hnsw_search(0, km, srows, g, sq, k, max(k, 10), case["dim"])   # default efRuntime 10 (hnsw.h:2073: max(ef, k))
    if case["order"] == "id":
        srt = np.argsort(ol, kind="stable")
        ol, od = ol[srt], od[srt]
    for got_l, got_d, who in ((gl[0], gd[0], "gpu"), (ol.astype(np.int64), od, "oracle")):
        assert len(got_l) == k, who
        if "expect_labels" in case:
            assert list(got_l) == case["expect_labels"], (who, got_l)
        if "expect_abs_diff" in case:
            assert [abs(int(x) - case["expect_labels_abs_diff_from"]) for x in got_l] == case["expect_abs_diff"], (who, got_l)
            assert list(got_d) == case["expect_scores"], (who, got_d)
        if "expect_label_range" in case:
            lo, hi = case["expect_label_range"]
            assert all(lo <= int(x) < hi for x in got_l) and all(float(s) <= case["expect_score_max"] for s in got_d), (who, got_l, got_d)
    if case["name"] == "testCosine":   # score == getDistanceFrom_Unsafe(id, normalised query) (test_hnsw.cpp:1606-1608)
        for lab, sc in zip(gl[0], gd[0]):
            assert sc == ix.get_distance_from(int(lab), sq)


def test_reference_hnsw_range_known_answers(vso):
    c = _kats()["range"]
    n, dim = c["n"], c["dim"]
    rows = np.repeat(np.arange(n, dtype=np.float32)[:, None], dim, axis=1)
    p = VecSim.HNSWParams()
    p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, dim, VecSim.VecSimMetric_L2
    ix = VecSim.HNSWIndex(p)
    for i in range(n):
        ix.add_vector(rows[i], i)
    q = np.full(dim, float(c["pivot"]), dtype=np.float32)
    g = ix.graph()
    for eps in c["epsilons"]:
        qp = VecSim.VecSimQueryParams()
        qp.hnswRuntimeParams.epsilon = eps
        l, d = ix.range_query(q, c["radius"], qp, VecSim.BY_SCORE)
        assert l.shape[1] == c["expect_count"]
        assert [abs(int(x) - c["pivot"]) for x in l[0]] == c["expect_abs_diff_by_score"] and list(d[0]) == c["expect_scores_by_score"]
        l, d = ix.range_query(q, c["radius"], qp, VecSim.BY_ID)
        assert list(l[0]) == c["expect_labels_by_id"]
        ol, od, _ = vso.hnsw_range(0, 0, rows, g, q, c["radius"], eps, dim)
        assert sorted(int(x) for x in ol) == c["expect_labels_by_id"]


# ---------------------------------------------------------------------------------------------------------------
# multi-value HNSW (hnsw_multi.h:16-247): a label owns several vectors, the search keeps one entry per label
def build_multi(dim, rows, labels, metric=VecSim.VecSimMetric_L2, M=16, efc=200, ef=10, typ=VecSim.VecSimType_FLOAT32):
    p = VecSim.HNSWParams()
    p.type, p.dim, p.metric, p.M, p.efConstruction, p.efRuntime, p.multi = typ, dim, metric, M, efc, ef, True
    ix = VecSim.HNSWIndex(p)
    for v, lab in zip(rows, labels):          # one at a time: the sequential insert path, repeated labels welcome
        assert ix.add_vector(v, int(lab)) == 1   # "we always add the vector, no overrides" (hnsw_multi.h:213-218)
    return ix


def test_multi_value_reference_known_answers(vso):
    """tests/unit/test_hnsw_multi.cpp closed forms (tests/golden/kat_hnsw.json, sections multi / multi_range) on the product (host-built
    graph, GPU search, C API) and on the oracle's multi-value search over the graph the index exports"""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "kat_hnsw.json")) as f:
        kat = json.load(f)
    for c in kat["multi"]:
        rows = np.array(c["vectors"], dtype=np.float32)
        ix = build_multi(c["dim"], rows, c["labels"])
        assert ix.index_size() == len(rows) and dict(ix.debug_info_fields())["INDEX_LABEL_COUNT"] == c["n_labels"]
        q = np.array(c["query"], dtype=np.float32)
        labels, dists = ix.knn_query(q, c["k"])
        el, es, _ = vso.hnsw_search(0, 0, rows, ix.graph(), q, c["k"], max(10, c["k"]), c["dim"], multi=True)
        for got_l, got_s in ((labels[0], dists[0]), (el.astype(np.int64), es)):
            assert len(set(got_l.tolist())) == len(got_l), c["name"]          # a label comes back once
            if "expect_labels" in c:
                assert got_l.tolist() == c["expect_labels"], c["name"]
            if "expect_abs_diff" in c:
                assert [abs(int(x) - c["expect_abs_diff_from"]) for x in got_l] == c["expect_abs_diff"], c["name"]
            if "expect_scores" in c:
                assert got_s.tolist() == c["expect_scores"], c["name"]
        assert np.array_equal(labels[0], el.astype(np.int64)) and np.array_equal(dists[0], es)
    r = kat["multi_range"]
    vecs, labs = [], []
    for i in range(r["n_labels"]):
        vecs.append([float(i)] * 4)
        labs.append(i)
        for _ in range(r["per_label"] - 1):
            vecs.append([float(i + r["n_labels"] * r["per_label"])] * 4)
            labs.append(i)
    ix = build_multi(4, np.array(vecs, dtype=np.float32), labs)
    q = np.full(4, float(r["pivot"]), dtype=np.float32)
    for eps in r["epsilons"]:
        qp = VecSim.VecSimQueryParams()
        qp.hnswRuntimeParams.epsilon = eps
        l, d = ix.range_query(q, r["radius"], qp)
        assert l.shape[1] == r["expect_count"]
        assert d[0].tolist() == r["expect_scores_by_score"]
        assert [abs(int(x) - r["pivot"]) for x in l[0]] == r["expect_abs_diff_by_score"]
    l, d = ix.range_query(q, r["radius"], order=VecSim.BY_ID)
    assert l[0].tolist() == r["expect_labels_by_id"]


@pytest.mark.parametrize("metric,dim,n,n_labels,ef,k", [
    (VecSim.VecSimMetric_L2, 32, 4000, 600, 60, 10),
    (VecSim.VecSimMetric_IP, 64, 3000, 200, 100, 20),
    (VecSim.VecSimMetric_Cosine, 48, 3000, 1000, 40, 10),
])
def test_multi_value_gpu_search_equals_reference_loops(vso, metric, dim, n, n_labels, ef, k):
    """random vectors, repeated vectors under different labels and repeated labels: labels, order, scores and the number of distance
    evaluations equal the oracle's label-keyed search (updatable_max_heap semantics) on the same graph"""
    rng = np.random.default_rng(n_labels)
    base = rng.uniform(-1, 1, (n // 2, dim)).astype(np.float32)
    rows = base[rng.integers(0, len(base), n)]
    labels = rng.integers(0, n_labels, n)
    p = VecSim.HNSWParams()
    p.type, p.dim, p.metric, p.M, p.efConstruction, p.efRuntime, p.multi = VecSim.VecSimType_FLOAT32, dim, metric, 12, 80, ef, True
    ix = VecSim.HNSWIndex(p)
    ix.add_vectors(rows, labels)                 # bulk path: repeated labels in one batch
    assert ix.index_size() == n
    g = ix.graph()
    q = rng.uniform(-1, 1, (40, dim)).astype(np.float32)
    got_l, got_s = ix.knn_query(q, k)
    evals = ix.last_distance_evals()
    srows, sq = stored(vso, rows, metric), stored(vso, q, metric)
    km = 0 if metric == VecSim.VecSimMetric_L2 else 1
    total = 0
    for j in range(len(q)):
        el, es, ev = vso.hnsw_search(0, km, srows, g, sq[j], k, ef, dim, multi=True)
        total += ev
        assert np.array_equal(got_l[j][:len(el)], el.astype(np.int64)), (j, got_l[j], el)
        assert np.array_equal(got_s[j][:len(es)], es), j
        assert len(set(el.tolist())) == len(el)
    assert evals == total - len(q)
    # get_distance_from = the minimum over the label's vectors (hnsw_multi.h:138-162); delete removes all of them
    lab = int(labels[0])
    mine = np.nonzero(labels == lab)[0]
    want = min(vso.distance(0, km, srows[i], sq[0], dim) for i in mine)
    assert ix.get_distance_from(lab, sq[0]) == want
    assert ix.delete_vector(lab) == len(mine) and ix.index_size() == n - len(mine)
    got_l, _ = ix.knn_query(q, k)
    assert lab not in set(got_l.ravel().tolist())
    # batch iterator (the graph walk; its batches against the oracle's: test_batch_iterator_walk_on_a_multi_value_index): a label
    # is handed out once, each batch ascending
    it = ix.create_batch_iterator(q[0])
    seen = []
    while it.has_next():
        l, d = it.get_next_results(50)
        live = l[0] >= 0
        assert np.all(np.diff(d[0][live]) >= 0)
        seen += l[0][live].tolist()
    assert len(seen) == len(set(seen)) and lab not in seen and set(seen) <= set(labels.tolist())


# ---------------------------------------------------------------------------------------------------------------
# FLOAT64 HNSW (index_factories/hnsw_factory.cpp:47: HNSWIndex<double, double>): rows scored in double in the reference's
# AVX-512F fp64 order, heaps and replies keep every bit of the double scores
@pytest.mark.parametrize("metric,dim,n,M,ef,k,multi", [
    (VecSim.VecSimMetric_L2, 32, 3000, 16, 50, 10, False),
    (VecSim.VecSimMetric_IP, 70, 2500, 8, 64, 10, False),
    (VecSim.VecSimMetric_Cosine, 96, 2500, 12, 40, 5, False),
    (VecSim.VecSimMetric_L2, 24, 3000, 12, 60, 10, True),
])
def test_fp64_gpu_search_equals_reference_loops_on_same_graph(vso, metric, dim, n, M, ef, k, multi):
    rng = np.random.default_rng(dim + n)
    rows = rng.uniform(-1, 1, (n, dim))
    labels = rng.integers(0, n // 5, n) if multi else np.arange(n)
    p = VecSim.HNSWParams()
    p.type, p.dim, p.metric, p.M, p.efConstruction, p.efRuntime, p.multi = VecSim.VecSimType_FLOAT64, dim, metric, M, 80, ef, multi
    ix = VecSim.HNSWIndex(p)
    ix.add_vectors(rows, labels)
    g = ix.graph()
    q = rng.uniform(-1, 1, (30, dim))
    got_l, got_s = ix.knn_query(q, k)
    evals = ix.last_distance_evals()

    def stored64(a):
        if metric != VecSim.VecSimMetric_Cosine:
            return a
        out = a.copy()
        for i in range(len(out)):
            vso.normalize(out[i], dim, vso.F64)
        return out
    srows, sq = stored64(rows), stored64(q)
    km = 0 if metric == VecSim.VecSimMetric_L2 else 1
    total = 0
    for j in range(len(q)):
        el, es, ev = vso.hnsw_search(vso.F64, km, srows, g, sq[j], k, ef, dim, multi=multi)
        total += ev
        assert np.array_equal(got_l[j][:len(el)], el.astype(np.int64)), (j, got_l[j], el)
        assert np.array_equal(got_s[j][:len(es)], es), (j, got_s[j], es)     # doubles, bit for bit
    assert evals == total - len(q)
    assert np.any(got_s != got_s.astype(np.float32).astype(np.float64))      # (the scores really are doubles)
    # range search in double as well
    rad = float(np.sort(got_s[0])[min(k, len(got_s[0])) - 1])
    if rad < 0:       # (IP scores of random vectors: the C API rejects a negative radius, as upstream)
        return
    l, d = ix.range_query(q[0], rad)
    el, es, _ = vso.hnsw_range(vso.F64, km, srows, g, sq[0], rad, 0.01, dim)
    if multi:
        best = {}
        for a, b in zip(el.tolist(), es.tolist()):
            best[a] = min(best.get(a, np.inf), b)
        want = sorted(best.items(), key=lambda t: (t[1], t[0]))
    else:
        want = sorted(zip(el.tolist(), es.tolist()), key=lambda t: (t[1], t[0]))
    assert [int(x) for x in l[0]] == [a for a, _ in want] and d[0].tolist() == [b for _, b in want]


def test_fp64_reference_known_answers():
    """the reference's typed HNSW tests run on double as well (tests/unit/test_hnsw.cpp: DataTypeSet): closed forms on fp64 rows"""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "kat_hnsw.json")) as f:
        kat = json.load(f)
    for c in kat["topk"]:
        if "expect_scores" not in c and "expect_labels" not in c:
            continue
        p = VecSim.HNSWParams()
        p.type, p.dim, p.metric, p.M, p.efConstruction = VecSim.VecSimType_FLOAT64, c["dim"], getattr(VecSim, "VecSimMetric_" + c["metric"]), c["M"], c["efConstruction"]
        ix = VecSim.HNSWIndex(p)
        for v, lab in zip(c["vectors"], c["labels"]):
            ix.add_vector(np.array(v, dtype=np.float64), lab)
        l, d = ix.knn_query(np.array(c["query"], dtype=np.float64), c["k"], order=VecSim.BY_ID if c["order"] == "id" else VecSim.BY_SCORE)
        if "expect_labels" in c:
            assert l[0].tolist() == c["expect_labels"], c["name"]
        if "expect_scores" in c:
            assert d[0].tolist() == c["expect_scores"], c["name"]
        if "expect_abs_diff" in c:
            assert [abs(int(x) - c["expect_labels_abs_diff_from"]) for x in l[0]] == c["expect_abs_diff"], c["name"]


def test_concurrent_readers_on_one_hnsw_index(vso):
    """several threads search (top-k and range) one HNSW index at once: a reader that finds the index's context busy runs on a
    reader lane (a view of the same snapshot with its own stream, staging and visited tags); every reply equals the serial one"""
    import threading
    dim, n, k = 48, 20_000, 10
    ix, rows = build(dim, n, VecSim.VecSimMetric_L2, M=12, efc=80, ef=64)
    rng = np.random.default_rng(4)
    qs = [rng.uniform(-1, 1, (1 + 37 * (i % 4), dim)).astype(np.float32) for i in range(16)]
    want = [ix.knn_query(q, k) for q in qs]
    want_r = [ix.range_query(q[0], float(w[1][0][4])) for q, w in zip(qs, want)]
    errors = []

    def worker(t):
        try:
            for rep in range(4):
                for i in range(t, len(qs), 4):
                    l, d = ix.knn_query(qs[i], k)
                    if not (np.array_equal(l, want[i][0]) and np.array_equal(d, want[i][1])):
                        errors.append((t, i, "knn"))
                    r = ix.range_query(qs[i][0], float(want[i][1][0][4]))
                    if not (np.array_equal(r[0], want_r[i][0]) and np.array_equal(r[1], want_r[i][1])):
                        errors.append((t, i, "range"))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]
    # a batch of deletes large enough to compact the graph (nodes renumbered, the device table's rows moved, the snapshot replaced):
    # the reader lanes' views must follow -- the same concurrent round again against fresh serial replies
    for lab in range(0, n, 9):
        assert ix.delete_vector(lab) == 1
    assert ix.graph()["n"] < n
    want[:] = [ix.knn_query(q, k) for q in qs]
    want_r[:] = [ix.range_query(q[0], float(w[1][0][4])) for q, w in zip(qs, want)]
    assert not any(l % 9 == 0 for w in want for l in w[0].ravel().tolist() if l >= 0)
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]


def test_debug_neighbours_dump_matches_the_exported_graph():
    """VecSimDebug_GetElementNeighborsInHNSWGraph (vec_sim_debug.h:30-44): per level {count, neighbour LABELS...}, NULL-terminated;
    BadIndex for a Flat index, LabelNotExists, MultiNotSupported -- as upstream"""
    import ctypes as C
    from vectorsimilarity_amd import _capi
    lib = _capi.load()
    fn = lib.VecSimDebug_GetElementNeighborsInHNSWGraph
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(C.POINTER(C.c_int)))]
    rel = lib.VecSimDebug_ReleaseElementNeighborsInHNSWGraph
    rel.restype = None
    rel.argtypes = [C.POINTER(C.POINTER(C.c_int))]
    ix, _ = build(16, 800, VecSim.VecSimMetric_L2, M=6, efc=40, ef=20, labels=np.arange(800) * 3 + 1)
    g = ix.graph()
    for node in (0, 17, 799, int(g["entry"])):
        out = C.POINTER(C.POINTER(C.c_int))()
        assert fn(ix._h, int(g["labels"][node]), C.byref(out)) == 0
        level = 0
        while out[level]:
            n = out[level][0]
            got = [out[level][1 + i] for i in range(n)]
            if level == 0:
                want = [int(g["labels"][j]) for j in g["links0"].reshape(-1, g["M0"])[node][: g["cnt0"][node]]]
                assert got == want, (node, got, want)
            assert n <= (g["M0"] if level == 0 else g["M"]) and all(x % 3 == 1 for x in got)
            level += 1
        assert level >= 1
        rel(out)
    out = C.POINTER(C.POINTER(C.c_int))()
    assert fn(ix._h, 2, C.byref(out)) == 2 and not out          # LabelNotExists
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, 4, VecSim.VecSimMetric_L2
    bf = VecSim.BFIndex(p)
    assert fn(bf._h, 0, C.byref(out)) == 1                       # BadIndex
    mp = VecSim.HNSWParams()
    mp.type, mp.dim, mp.metric, mp.multi = VecSim.VecSimType_FLOAT32, 4, VecSim.VecSimMetric_L2, True
    mx = VecSim.HNSWIndex(mp)
    mx.add_vector(np.zeros(4, dtype=np.float32), 5)
    assert fn(mx._h, 5, C.byref(out)) == 3                       # MultiNotSupported


# ---- the batch iterator's graph walk (hnsw_batch_iterator.h:96-230): product (host walk, GPU distances) == oracle twin ----
def _walk_batches(ix, q, sizes, qp=None, order=VecSim.BY_SCORE):
    it = ix.create_batch_iterator(q, qp)
    out = []
    for m in sizes:
        if not it.has_next():
            break
        l, d = it.get_next_results(m, order)
        out.append((l[0], d[0]))
    return it, out


@pytest.mark.parametrize("metric,dim,n,M,ef,sizes,dead", [
    (VecSim.VecSimMetric_L2, 32, 3000, 16, 10, [10] * 12, 0),
    (VecSim.VecSimMetric_L2, 24, 2500, 8, 20, [5, 7, 50, 1, 0, 33, 200, 3], 0),        # batches above and below ef, an empty one
    (VecSim.VecSimMetric_Cosine, 100, 2000, 12, 16, [16, 16, 40, 8], 0),
    (VecSim.VecSimMetric_IP, 64, 2000, 8, 12, [12, 30, 12, 12], 0),
    (VecSim.VecSimMetric_L2, 16, 1500, 6, 10, [25] * 70, 0),                              # to depletion: every label exactly once
    (VecSim.VecSimMetric_L2, 32, 3000, 16, 10, [10, 40, 10, 100], 7),                      # deleted nodes: traversed, never returned
    (VecSim.VecSimMetric_L2, 768, 6000, 16, 128, [10, 10, 100], 0),                        # config 5's shape
])
def test_batch_iterator_walk_equals_the_reference_walk_on_the_same_graph(vso, metric, dim, n, M, ef, sizes, dead):
    ix, rows = build(dim, n, metric, M=M, efc=200 if dim == 768 else 60, ef=ef)
    if dead:
        for lab in range(3, n, dead * 50):
            ix.delete_vector(lab)
    g = ix.graph()
    srows = stored(vso, rows, metric)
    km = 0 if metric == VecSim.VecSimMetric_L2 else 1
    rng = np.random.default_rng(1234)
    for trial in range(3):
        q = rng.uniform(-1, 1, dim).astype(np.float32)
        sq = stored(vso, q[None, :], metric)[0]
        want, want_depleted = vso.hnsw_iterate(0, km, srows, g, sq, ef, sizes, dim)
        it, got = _walk_batches(ix, q, sizes)
        assert len(got) == len(want), (trial, len(got), len(want))
        for b, ((gl, gd), (wl, wd)) in enumerate(zip(got, want)):
            assert np.array_equal(gl[:len(wl)], wl.astype(np.int64)) and np.all(gl[len(wl):] == -1), (trial, b, gl, wl)
            assert np.array_equal(gd[:len(wd)], wd), (trial, b)
        assert (not it.has_next()) == want_depleted
        if want_depleted:   # every live label exactly once
            allv = np.concatenate([l[l >= 0] for l, _ in got])
            live = np.asarray(g["labels"])[~np.asarray(g["deleted"], dtype=bool)]
            assert sorted(allv.tolist()) == sorted(live.astype(np.int64).tolist())
        # Reset: the same iteration again
        it.reset()
        l, d = it.get_next_results(sizes[0], VecSim.BY_SCORE)
        assert np.array_equal(l[0][:len(want[0][0])], want[0][0].astype(np.int64)) and np.array_equal(d[0][:len(want[0][1])], want[0][1])


def test_batch_iterator_walk_uses_the_query_params_ef_and_sorts_by_id(vso):
    dim, n = 32, 2000
    ix, rows = build(dim, n, VecSim.VecSimMetric_L2, M=8, efc=60, ef=10)
    g = ix.graph()
    q = np.random.default_rng(5).uniform(-1, 1, dim).astype(np.float32)
    qp = VecSim.VecSimQueryParams()
    qp.hnswRuntimeParams.efRuntime = 37
    want, _ = vso.hnsw_iterate(0, 0, rows, g, q, 37, [9, 9, 9], dim)
    _, got = _walk_batches(ix, q, [9, 9, 9], qp=qp, order=VecSim.BY_ID)
    for (gl, gd), (wl, wd) in zip(got, want):
        srt = np.argsort(wl, kind="stable")
        assert np.array_equal(gl, wl[srt].astype(np.int64)) and np.array_equal(gd, wd[srt])
    base, _ = vso.hnsw_iterate(0, 0, rows, g, q, 10, [9, 9, 9], dim)
    assert any(not np.array_equal(a[0], b[0]) for a, b in zip(want, base)) or True   # (ef may or may not change the batches)


def test_batch_iterator_walk_on_a_multi_value_index(vso):
    dim, n, n_labels = 24, 2400, 400
    rng = np.random.default_rng(31)
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    labels = rng.integers(0, n_labels, n)
    ix = build_multi(dim, rows, labels, M=8, efc=60, ef=10)
    g = ix.graph()
    for trial in range(3):
        q = rng.uniform(-1, 1, dim).astype(np.float32)
        sizes = [10, 25, 10, 3, 60] + [40] * 12
        want, want_depleted = vso.hnsw_iterate(0, 0, rows, g, q, 10, sizes, dim, multi=True)
        it, got = _walk_batches(ix, q, sizes)
        assert len(got) == len(want)
        seen = []
        for (gl, gd), (wl, wd) in zip(got, want):
            assert np.array_equal(gl[:len(wl)], wl.astype(np.int64)) and np.all(gl[len(wl):] == -1)
            assert np.array_equal(gd[:len(wd)], wd)
            seen += gl[gl >= 0].tolist()
        assert len(seen) == len(set(seen))   # a label is handed out once
        assert (not it.has_next()) == want_depleted


def test_reference_hnsw_batch_iterator_known_answers():
    """tests/unit/test_hnsw.cpp:912-1118 (hnsw_batch_iterator_basic / _reset / _batch_size_1 / _advanced), restated as data:
    vectors (i,i,i,i) under label i, query (n,n,n,n): batches come back from the largest id down"""
    def index(n, M, ef, labels=None):
        p = VecSim.HNSWParams()
        p.type, p.dim, p.metric, p.M, p.efConstruction, p.efRuntime = VecSim.VecSimType_FLOAT32, 4, VecSim.VecSimMetric_L2, M, ef, ef
        ix = VecSim.HNSWIndex(p)
        for i in range(n):
            ix.add_vector(np.full(4, float(i), dtype=np.float32), i if labels is None else labels[i])
        return ix
    # basic: n = 1000, ef 20, batches of 5
    n = 1000
    ix = index(n, 8, 20)
    q = np.full(4, float(n), dtype=np.float32)
    it = ix.create_batch_iterator(q)
    iters = 0
    while it.has_next():
        l, d = it.get_next_results(5, VecSim.BY_SCORE)
        assert l[0].tolist() == [n - iters * 5 - i - 1 for i in range(5)]
        iters += 1
    assert iters == n // 5
    # reset: batches of 100, three takes
    it = ix.create_batch_iterator(q)
    for take in range(3):
        iters = 0
        while it.has_next():
            l, d = it.get_next_results(100, VecSim.BY_SCORE)
            assert l[0].tolist() == [n - iters * 100 - i - 1 for i in range(100)]
            iters += 1
        assert iters == n // 100
        it.reset()
    # batch_size_1: labels n - i, ef 2: one result per batch, label == iteration number
    ix = index(n, 8, 2, labels=[n - i for i in range(n)])
    it = ix.create_batch_iterator(q)
    iters = 0
    while it.has_next():
        iters += 1
        l, d = it.get_next_results(1, VecSim.BY_SCORE)
        assert l[0].tolist() == [iters]
    assert iters == n
    # advanced: ef = n = 500; empty index, one vector, zero results, batches of 7 by id, nothing after depletion
    n = 500
    q = np.full(4, float(n), dtype=np.float32)
    ix = index(0, 8, n)
    it = ix.create_batch_iterator(q)
    l, d = it.get_next_results(10, VecSim.BY_SCORE)
    assert np.all(l[0] == -1) and not it.has_next()
    ix.add_vector(q, n)
    it = ix.create_batch_iterator(q)
    l, d = it.get_next_results(10, VecSim.BY_SCORE)
    assert (l[0] >= 0).sum() == 1 and not it.has_next()
    for i in range(1, n):
        ix.add_vector(np.full(4, float(i), dtype=np.float32), i)
    it = ix.create_batch_iterator(q)
    l, d = it.get_next_results(0, VecSim.BY_SCORE)
    assert l.shape[1] == 0 or np.all(l[0] == -1)
    iters = 0
    while it.has_next():
        iters += 1
        expect = [n - iters * 7 + i for i in range(1, 8)]
        if iters > n // 7:
            expect = expect[7 - n % 7:]
        l, d = it.get_next_results(7, VecSim.BY_ID)
        assert l[0][l[0] >= 0].tolist() == expect, (iters, l[0], expect)
    assert iters == n // 7 + 1
    l, d = it.get_next_results(1, VecSim.BY_SCORE)
    assert np.all(l[0] == -1)


def test_reference_hnsw_batch_iterator_timeouts():
    """tests/unit/test_hnsw.cpp:1728-1800: a timeout between batches, during the first scan, and in the descent to level 0"""
    p = VecSim.HNSWParams()
    p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, 4, VecSim.VecSimMetric_L2
    ix = VecSim.HNSWIndex(p)
    for i in range(2):
        ix.add_vector(np.full(4, 1.0, dtype=np.float32), 46 - i)
    q = np.full(4, 1.0, dtype=np.float32)
    lib = ix._lib
    try:
        it = ix.create_batch_iterator(q)
        l, d = it.get_next_results(1, VecSim.BY_ID)
        assert (l[0] >= 0).sum() == 1
        cb = VecSim.set_timeout_callback(lambda ctx: 1)
        rep = lib.VecSimBatchIterator_Next(it._h, 1, VecSim.BY_ID)
        assert lib.VecSimQueryReply_GetCode(rep) == 1 and lib.VecSimQueryReply_Len(rep) == 0
        lib.VecSimQueryReply_Free(rep)
        # fails on the second call of the callback: the first batch, while scanning
        calls = []
        cb = VecSim.set_timeout_callback(lambda ctx: (calls.append(1), 0 if len(calls) == 1 else 1)[1])
        it = ix.create_batch_iterator(q)
        rep = lib.VecSimBatchIterator_Next(it._h, 2, VecSim.BY_ID)
        assert lib.VecSimQueryReply_GetCode(rep) == 1 and lib.VecSimQueryReply_Len(rep) == 0
        lib.VecSimQueryReply_Free(rep)
        assert len(calls) == 2
        # in the descent: needs a node above level 0
        VecSim.set_timeout_callback(None)
        nxt = 0
        while ix.graph()["max_level"] == 0:
            ix.add_vector(np.full(4, 1.0, dtype=np.float32), nxt)
            nxt += 1
        cb = VecSim.set_timeout_callback(lambda ctx: 1)
        it = ix.create_batch_iterator(q)
        rep = lib.VecSimBatchIterator_Next(it._h, 2, VecSim.BY_ID)
        assert lib.VecSimQueryReply_GetCode(rep) == 1 and lib.VecSimQueryReply_Len(rep) == 0
        lib.VecSimQueryReply_Free(rep)
    finally:
        VecSim.set_timeout_callback(None)


@pytest.mark.parametrize("typ,metric,dim", [("bf16", "L2", 64), ("f16", "IP", 48), ("i8", "Cosine", 64), ("u8", "L2", 40), ("f64", "L2", 32)])
def test_batch_iterator_walk_over_the_other_stored_types(vso, typ, metric, dim):
    """the walk's distances are the stored type's reference-order kernel (fp64: doubles, heaps and replies keep every bit)"""
    from util import METRICS, TYPES, random_vectors, stored_rows
    n, ef = 2000, 12
    rng = np.random.default_rng(dim + 5)
    if typ == "f64":
        rows, q = rng.uniform(-1, 1, (n, dim)), rng.uniform(-1, 1, (6, dim))
        vt, srows, sq = VecSim.VecSimType_FLOAT64, rows, q
    else:
        rows = random_vectors(rng, n, dim, typ, vso)
        q = random_vectors(rng, 6, dim, typ, vso)
        vt = TYPES[typ]
        srows, sq = stored_rows(vso, rows, typ, metric), stored_rows(vso, q, typ, metric)
    p = VecSim.HNSWParams()
    p.type, p.dim, p.metric, p.M, p.efConstruction, p.efRuntime = vt, dim, METRICS[metric], 10, 60, ef
    ix = VecSim.HNSWIndex(p)
    ix.add_vectors(rows, np.arange(n))
    for lab in (7, 300):
        ix.delete_vector(lab)
    g = ix.graph()
    km = METRICS["IP"] if (metric == "Cosine" and typ not in ("i8", "u8")) else METRICS[metric]
    sizes = [12, 5, 40, 12]
    for j in range(len(q)):
        want, _ = vso.hnsw_iterate(vso.F64 if typ == "f64" else vt, km, srows, g, sq[j], ef, sizes, dim)
        _, got = _walk_batches(ix, q[j], sizes)
        assert len(got) == len(want)
        for (gl, gd), (wl, wd) in zip(got, want):
            assert np.array_equal(gl[:len(wl)], wl.astype(np.int64)) and np.all(gl[len(wl):] == -1), (typ, j)
            assert np.array_equal(gd[:len(wd)], wd), (typ, j)
