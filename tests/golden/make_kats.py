#!/usr/bin/env python3
"""Writes tests/golden/kat_spaces.json and kat_flat.json: the reference's OWN known-answer tests for
the hot path, restated as data (inputs + the expected values the reference asserts).

Nothing here is computed by our oracle or our kernels: every expected value is the closed form
the reference test states.  Sources (paths under /root/reference/tests/unit/):
  test_spaces.cpp:69-154    L2 "no optimisation" answers         -> l2_noopt
  test_spaces.cpp:157-283   IP answers, bf16/fp16 normalisation  -> ip_noopt, normalize
  test_spaces.cpp:286-322   int8/uint8 Cosine of a vector with itself ~ 0
  test_spaces.cpp:684-779, 878-879, 1073-1074, 1249-1250, 1404-1405
                            "every SIMD tier == scalar kernel" on v[i]=i, v2[i]=i+1.5 for every
                            residual; both sides are exact in fp arithmetic, so the expected value
                            is the exact real number: L2 = 2.25*d, IP = 1 - sum i(i+1.5)
  test_bruteforce.cpp:747-812   Flat top-k on {i,i,i,i} vectors (L2 closed form / IP id set)
  test_bruteforce.cpp:1389-1419 Cosine ranking: rank r <-> label n-r
  test_int8.cpp:366-386         tie order pinned: {50,49,51,48,52,...}
Run:  python tests/golden/make_kats.py
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def spaces():
    cases = []
    # --- no-optimisation answers ---
    a = [i + 1.5 for i in range(5)]
    cases.append(dict(name="float_l2_no_opt", type="f32", metric="L2", tier="scalar", a=a, b=a, expect=0.0,
                      src="test_spaces.cpp:69-81"))
    cases.append(dict(name="double_l2_no_opt", type="f64", metric="L2", tier="scalar", a=a, b=a, expect=0.0,
                      src="test_spaces.cpp:83-95"))
    sa = [0.5 + i * 0.25 for i in range(4)]
    sb = [i * 0.25 for i in range(4)]
    # bf16/fp16 of multiples of 0.25 are exact; expected = fp32 L2 of the same values = 4*0.25
    cases.append(dict(name="bf16_l2_no_opt", type="bf16", metric="L2", tier="scalar", a=sa, b=sb, expect=1.0,
                      src="test_spaces.cpp:97-113"))
    cases.append(dict(name="fp16_l2_no_opt", type="f16", metric="L2", tier="scalar", a=sa, b=sb, expect=1.0,
                      src="test_spaces.cpp:115-131"))
    ia = [i + 1 for i in range(5)]
    ib = [i + 2 for i in range(5)]
    cases.append(dict(name="int8_l2_no_opt", type="i8", metric="L2", tier="scalar", a=ia, b=ib, expect=5.0,
                      src="test_spaces.cpp:133-144"))
    cases.append(dict(name="uint8_l2_no_opt", type="u8", metric="L2", tier="scalar", a=ia, b=ib, expect=5.0,
                      src="test_spaces.cpp:146-154"))
    ip = 1.0 - sum(x * y for x, y in zip(sa, sb))
    cases.append(dict(name="bf16_ip_no_opt", type="bf16", metric="IP", tier="scalar", a=sa, b=sb, expect=ip,
                      src="test_spaces.cpp:233-249"))
    cases.append(dict(name="fp16_ip_no_opt", type="f16", metric="IP", tier="scalar", a=sa, b=sb, expect=ip,
                      src="test_spaces.cpp:251-267"))
    cases.append(dict(name="int8_ip_no_opt", type="i8", metric="IP", tier="scalar", a=[1, 0, 0, 0], b=[1, 0, 0, 0],
                      expect=0.0, src="test_spaces.cpp:269-275"))
    cases.append(dict(name="uint8_ip_no_opt", type="u8", metric="IP", tier="scalar", a=[1, 0, 0, 0], b=[1, 0, 0, 0],
                      expect=0.0, src="test_spaces.cpp:277-283"))
    # --- SIMD tier == scalar on exactly-summable inputs, every residual ---
    def exact(d):
        l2 = 2.25 * d
        ipv = 1.0 - sum(i * (i + 1.5) for i in range(d))
        return l2, ipv
    for typ, lo, hi in (("f32", 8, 64), ("f64", 4, 32), ("bf16", 32, 64), ("f16", 16, 64)):
        for d in range(lo, hi + 1):
            l2, ipv = exact(d)
            v = [float(i) for i in range(d)]
            w = [i + 1.5 for i in range(d)]
            for tier in ("avx512", "scalar"):
                cases.append(dict(name=f"{typ}_l2_simd_eq_scalar_d{d}_{tier}", type=typ, metric="L2", tier=tier,
                                  a=v, b=w, expect=l2, src="test_spaces.cpp:684-779 (param dims)"))
                cases.append(dict(name=f"{typ}_ip_simd_eq_scalar_d{d}_{tier}", type=typ, metric="IP", tier=tier,
                                  a=v, b=w, expect=ipv, src="test_spaces.cpp:781-876 (param dims)"))
    # bf16 IP through the vdpbf16ps tier as well (test_spaces.cpp:1184-1247)
    for d in range(32, 65):
        l2, ipv = exact(d)
        cases.append(dict(name=f"bf16_ip_dpbf16_d{d}", type="bf16", metric="IP", tier="avx512_bf16",
                          a=[float(i) for i in range(d)], b=[i + 1.5 for i in range(d)], expect=ipv,
                          src="test_spaces.cpp:1184-1247"))
    norm = [dict(name="bf16_normalize", type="bf16", input=[4.0] * 4, expect=[0.5] * 4, src="test_spaces.cpp:191-211"),
            dict(name="fp16_normalize", type="f16", input=[4.0] * 4, expect=[0.5] * 4, src="test_spaces.cpp:213-231")]
    return dict(distance=cases, normalize=norm)


def flat():
    cases = []
    n, k, dim = 100, 11, 4
    # L2: vectors {i,i,i,i}, query {50..}: rank r -> |id-50| = (r+1)//2, score 4*((r+1)//2)^2
    cases.append(dict(name="bf_search_l2", metric="L2", dim=dim, n=n, k=k, query_value=50,
                      vector_rule="label i -> [i]*dim", types=["f32", "f64", "bf16", "f16"],
                      block_sizes=[1, 12, 1024],
                      expect_absdiff=[(r + 1) // 2 for r in range(k)],
                      expect_scores=[4.0 * ((r + 1) // 2) ** 2 for r in range(k)],
                      src="test_bruteforce.cpp:781-812"))
    # IP: query {50..}: the 11 largest ids win (most negative 1 - 200*i)
    cases.append(dict(name="bf_search_ip", metric="IP", dim=dim, n=n, k=k, query_value=50,
                      vector_rule="label i -> [i]*dim", types=["f32", "f64"],
                      block_sizes=[1, 12, 1024],
                      expect_id_set=list(range(n - k, n)), src="test_bruteforce.cpp:747-779"))
    # int8 L2 with ties: ascending score, then ascending label
    cases.append(dict(name="int8_search_by_score", metric="L2", dim=dim, n=n, k=k, query_value=50,
                      vector_rule="label i -> [i]*dim", types=["i8", "u8"],
                      expect_labels=[50, 49, 51, 48, 52, 47, 53, 46, 54, 45, 55],
                      expect_scores=[4.0 * (50 - i) ** 2 for i in [50, 49, 51, 48, 52, 47, 53, 46, 54, 45, 55]],
                      src="test_int8.cpp:366-386"))
    cases.append(dict(name="int8_search_by_id", metric="L2", dim=dim, n=n, k=k, query_value=50,
                      vector_rule="label i -> [i]*dim", types=["i8", "u8"], order="BY_ID",
                      expect_labels=list(range(45, 56)),
                      expect_scores=[4.0 * (50 - i) ** 2 for i in range(45, 56)],
                      src="test_int8.cpp:333-347"))
    # Cosine: f[0] = i/n, rest 1.0, labels 1..n; query all ones: rank r -> label n-r
    cases.append(dict(name="bf_cosine", metric="Cosine", dim=128, n=100, k=10,
                      vector_rule="label i in 1..n -> [i/n, 1, 1, ...]", query_rule="[1.0]*dim",
                      types=["f32", "f64"], expect_labels=[100 - r for r in range(10)],
                      src="test_bruteforce.cpp:1389-1419"))
    # heap/tie semantics probed on the compiled reference during the survey:
    # scan order (label,dist) = (5,16),(9,16),(1,4),(2,16),(0,16)
    cases.append(dict(name="tie_probe", scan=[[5, 16.0], [9, 16.0], [1, 4.0], [2, 16.0], [0, 16.0]],
                      expect={"2": [1, 5], "3": [1, 5, 9]}, src="SURVEY.md §8(a) A10 [probed]"))
    return dict(cases=cases)


def flat_multi():
    """closed forms of tests/unit/test_bruteforce_multi.cpp: every vector is [value]*4, fp32 L2, query = 0; a case lists
    (label, value) per vector in insertion order, k, and the (label, score) reply the reference test asserts"""
    dim = 4
    cases = []
    # :251-288 5 vectors under 2 labels, k=3 -> only 2 results (unique labels), label i at rank i
    vec = [[i // 3, float(i)] for i in range(5)]
    cases.append(dict(name="search_more_than_there_is", dim=dim, vectors=vec, k=3, expect_labels=[0, 1],
                      expect_scores=[0.0, 4.0 * 9.0], src="test_bruteforce_multi.cpp:251-288"))
    # :290-324 100 vectors, 10 per label, value i: label i at rank i (its best vector is 10 i)
    vec = [[i // 10, float(i)] for i in range(100)]
    cases.append(dict(name="indexing_same_vector", dim=dim, vectors=vec, k=10, expect_labels=list(range(10)),
                      expect_scores=[4.0 * (10 * i) ** 2 for i in range(10)], src="test_bruteforce_multi.cpp:290-324"))
    # :326-371 each label gets ever better vectors, neighbours share scores; rank r -> label k-r-1, score el^2*dim
    n, nl, k = 100, 10, 10
    vec, best = [], {}
    for i in range(n):
        el = ((n - i - 1) % nl) + ((n - i - 1) // nl)
        vec.append([i // nl, float(el)])
        if i % nl == nl - 1:
            best[i // nl] = float(el * el * dim)
    cases.append(dict(name="find_better_score", dim=dim, vectors=vec, k=k, expect_labels=[k - r - 1 for r in range(k)],
                      expect_scores=[best[k - r - 1] for r in range(k)], src="test_bruteforce_multi.cpp:326-371"))
    # :373-403 12 vectors over 3 labels, each better than the previous: rank r -> label n_labels-r-1, for k = 2 and k = 3
    n, nl = 12, 3
    vec = [[i % nl, float(n - i)] for i in range(n)]
    for k in (nl - 1, nl):
        cases.append(dict(name="find_better_score_after_pop_k%d" % k, dim=dim, vectors=vec, k=k,
                          expect_labels=[nl - r - 1 for r in range(k)], src="test_bruteforce_multi.cpp:373-403"))
    return dict(cases=cases)


def sq8():
    """SQ8 known answers the reference's tests hold (tests/unit/test_components.cpp, test_spaces.cpp).  Expected bytes of the
    quantiser tests follow the test helper the reference compares with (unit_test_utils.h:240-277: round((x - min) / delta));
    metadata expectations are the values the tests assert (FLOAT_EQ = 4 ulp), distances the closed forms they state."""
    import math
    import struct

    def f32(x):
        return struct.unpack("f", struct.pack("f", x))[0]

    def helper_bytes(v):   # ComputeSQ8Quantization, in fp32 like the helper
        mn, mx = min(v), max(v)
        diff = f32(mx - mn)
        delta = 1.0 if diff == 0.0 else f32(diff / 255.0)
        return [int(math.floor(f32(f32(x - mn) / delta) + 0.5)) & 0xFF for x in v], mn, delta

    q = []
    for name, v, metric, src in (("quantization_test", [1, 2, 3, 4, 5, 6], "L2", "test_components.cpp:1047-1100"),
                                 ("metric_l2", [1, 2, 3, 4, 5], "L2", "test_components.cpp:1446-1596"),
                                 ("metric_ip", [1, 2, 3, 4, 5], "IP", "test_components.cpp:1446-1596"),
                                 ("metric_cosine", [1, 2, 3, 4, 5], "Cosine", "test_components.cpp:1446-1596")):
        b, mn, delta = helper_bytes([float(x) for x in v])
        qs, qq = sum(b), sum(x * x for x in b)
        sm = f32(len(v) * mn + delta * qs)
        sq = f32(len(v) * mn * mn + 2.0 * mn * delta * qs + delta * delta * qq)
        q.append(dict(name=name, input=[float(x) for x in v], metric=metric, bytes=b, min=mn, delta=delta, sum=sm,
                      sum_squares=sq if metric == "L2" else None, exact_meta=True,
                      query_sum=float(sum(v)), query_sum_squares=float(sum(x * x for x in v)) if metric == "L2" else None, src=src))
    q.append(dict(name="all_entries_equal", input=[3.5] * 5, metric="L2", bytes=[0] * 5, min=3.5, delta=1.0, sum=17.5,
                  sum_squares=61.25, exact_meta=False, src="test_components.cpp:1395-1443"))
    huge = 3.4028234663852886e38
    q.append(dict(name="non_representable_range", input=[-huge, 0.0, huge, 1.0], metric="L2", bytes=[0, 0, 0, 0],
                  src="test_components.cpp:1208-1239"))
    q.append(dict(name="delta_underflows", input=[0.0, 1e-44], metric="L2", bytes=[0, 255], src="test_components.cpp:1250-1270"))
    for name, v, b in (("all_equal_positive", [3.5] * 3, [0] * 3), ("all_equal_negative", [-2.5] * 3, [0] * 3),
                       ("all_zero", [0.0] * 3, [0] * 3), ("subnormal_delta_survives", [0.0, 7e-37], [0, 255]),
                       ("single_element", [7.5], [0])):
        q.append(dict(name=name, input=v, metric="L2", bytes=b, finite_positive_delta=True, src="test_components.cpp:1356-1393"))
    d = []
    # test_spaces.cpp:485-530 states this closed form for the FP16-query kernel; the same blobs through the FP32-query
    # kernel have the same exact answer (every term is a small integer): L2 = 55 + 90 - 2*70 = 5, IP = 1 - 70
    d.append(dict(name="l2_closed_form", dim=5, codes=[1, 2, 3, 4, 5], meta=[0.0, 1.0, 15.0, 55.0],
                  query=[2.0, 3.0, 4.0, 5.0, 6.0], qmeta=[20.0, 90.0], metric="L2", expect=5.0, src="test_spaces.cpp:485-530"))
    d.append(dict(name="ip_closed_form", dim=5, codes=[1, 2, 3, 4, 5], meta=[0.0, 1.0, 15.0, 55.0],
                  query=[2.0, 3.0, 4.0, 5.0, 6.0], qmeta=[20.0, 90.0], metric="IP", expect=-69.0, src="test_spaces.cpp:485-530"))
    # the FP16-query kernel's own closed form (the only exact distance the reference's SQ8 tests state)
    d.append(dict(name="l2_closed_form_fp16_query", dim=5, codes=[1, 2, 3, 4, 5], meta=[0.0, 1.0, 15.0, 55.0],
                  query=[2.0, 3.0, 4.0, 5.0, 6.0], qmeta=[20.0, 90.0], metric="L2", expect=5.0, query_type="f16",
                  src="test_spaces.cpp:485-530"))
    # mean-centred blobs (QuantPreprocessor<..., WithNorm = true>, DistanceCalculatorWithNorm): test_components.cpp:1820-2020
    # (blob layout + metadata, FLOAT_EQ = 4 ulp; storage = the plain quantiser on x - mean) and :2095-2366 (calculator
    # results within 0.05 / 0.001 of the brute-force distance of the ORIGINAL vectors; zero mean == the base kernels)
    def seqsum(v):
        a = 0.0
        for x in v:
            a = f32(a + f32(x))
        return a
    src5, mean5 = [1.0, 2.0, 3.0, 4.0, 5.0], [f32(0.1), f32(0.2), f32(0.3), f32(0.4), f32(0.5)]
    wn = []
    for metric in ("L2", "IP"):
        centred = [f32(x - m) for x, m in zip(src5, mean5)]
        body = centred if metric == "L2" else src5
        wn.append(dict(name="with_norm_blob_" + metric.lower(), metric=metric, input=src5, mean=mean5, centred=centred,
                       storage_bytes=5 + 16, query_bytes=5 * 4 + 8, query_body=body,
                       x_mean_ip=seqsum(f32(x * m) for x, m in zip(src5, mean5)) if metric == "IP" else None,
                       y_sum=seqsum(body), y_sum_squares=seqsum(f32(b * b) for b in body) if metric == "L2" else None,
                       y_mean_ip=seqsum(f32(x * m) for x, m in zip(src5, mean5)) if metric == "IP" else None,
                       src="test_components.cpp:1820-2020"))
    x8 = [1.0, 2.0, 3.0, 4.0, 1.0, 2.0, 3.0, 4.0]
    y8 = [0.5, 1.5, 2.5, 3.5, 0.5, 1.5, 2.5, 3.5]
    m8 = [0.5, 1.0, 1.5, 2.0, 0.5, 1.0, 1.5, 2.0]
    calc = []
    for metric in ("L2", "IP"):
        exp = sum((a - b) ** 2 for a, b in zip(x8, y8)) if metric == "L2" else 1.0 - sum(a * b for a, b in zip(x8, y8))
        for mode, inp in (("asymmetric", "f32"), ("symmetric", "f32"), ("asymmetric", "f16")):
            calc.append(dict(name="calc_%s_%s_%s" % (metric.lower(), mode, inp), metric=metric, mode=mode, input_type=inp,
                             x=x8, y=y8, mean=m8, mean_sum_squares=sum(m * m for m in m8), expect=exp, tol=0.05,
                             src="test_components.cpp:2095-2366"))
    calc.append(dict(name="l2_large_offset_small_distance", metric="L2", mode="asymmetric", input_type="f32",
                     x=[1001.0] * 128, y=[f32(1001.1)] * 128, mean=[1000.0] * 128, mean_sum_squares=128 * 1e6,
                     expect=seqsum([f32(f32(1001.0 - f32(1001.1)) ** 2)] * 128), tol=0.001, nonnegative=True,
                     src="test_components.cpp:2164-2200"))
    zero = dict(name="zero_mean_matches_base", x=[1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0], y=[0.5, 1.5, 2.5, 3.5, 4.5, 5.5, 6.5, 7.5],
                src="test_components.cpp:2369-2415")
    return dict(quantize=q, distance=d, with_norm=dict(blobs=wn, calculator=calc, zero_mean=zero),
                tolerance_property=dict(abs=0.01, src="test_spaces.cpp:326-410, 2330-2700: every tier within 0.01 of the "
                                        "reconstruct-then-dot baseline (tests/utils/tests_utils.h:76-170, 244-270)"))


def hnsw():
    """The reference's deterministic HNSW unit tests (tests/unit/test_hnsw.cpp) as data: how the index is filled, the query,
    and the closed form every result must satisfy.  Vectors {v,v,v,v} (GenerateAndAddVector: every element = the value)."""
    cases = []
    cases.append(dict(name="hnsw_vector_search_test", src="test_hnsw.cpp:225-248", dim=4, metric="L2", M=16, efConstruction=200,
                      vectors=[[float(i)] * 4 for i in range(100)], labels=list(range(100)), query=[50.0] * 4, k=11, order="score",
                      expect_labels_abs_diff_from=50, expect_scores=[4.0 * ((r + 1) // 2) ** 2 for r in range(11)],
                      expect_abs_diff=[(r + 1) // 2 for r in range(11)]))
    cases.append(dict(name="hnsw_vector_search_by_id_test", src="test_hnsw.cpp:250-269", dim=4, metric="L2", M=16, efConstruction=200,
                      vectors=[[float(i)] * 4 for i in range(100)], labels=list(range(100)), query=[50.0] * 4, k=11, order="id",
                      expect_labels=[r + 45 for r in range(11)]))
    cases.append(dict(name="hnsw_indexing_same_vector", src="test_hnsw.cpp:271-293", dim=4, metric="L2", M=16, efConstruction=200,
                      vectors=[[float(i // 10)] * 4 for i in range(100)], labels=list(range(100)),
                      query=[4.9, 4.95, 5.05, 5.1], k=10, order="score", expect_label_range=[50, 60], expect_score_max=1.0))
    n = 100
    cases.append(dict(name="testCosine", src="test_hnsw.cpp:1580-1632", dim=4, metric="Cosine", M=16, efConstruction=200,
                      vectors=[[i / n, 1.0, 1.0, 1.0] for i in range(1, n + 1)], labels=list(range(1, n + 1)),
                      query=[1.0] * 4, k=10, order="score", expect_labels=[n - r for r in range(10)]))
    rq = dict(name="rangeQuery", src="test_hnsw.cpp:1801-1845", dim=4, metric="L2", M=16, efConstruction=200, n=5000, pivot=2500,
              radius=4.0 * 5 ** 2, expect_count=11, expect_scores_by_score=[4.0 * ((r + 1) // 2) ** 2 for r in range(11)],
              expect_abs_diff_by_score=[(r + 1) // 2 for r in range(11)], expect_labels_by_id=[2500 - 5 + r for r in range(11)],
              epsilons=[0.01, 1.0])
    # multi-value HNSW (tests/unit/test_hnsw_multi.cpp): labels own several vectors, a label is reported once with its best score
    multi = []
    multi.append(dict(name="multi_vector_search_test", src="test_hnsw_multi.cpp:106-132", dim=4, metric="L2",
                      vectors=[[float(i)] * 4 for i in range(1000)], labels=[i % 100 for i in range(1000)], n_labels=100,
                      query=[50.0] * 4, k=11, expect_abs_diff_from=50, expect_abs_diff=[(r + 1) // 2 for r in range(11)],
                      expect_scores=[4.0 * ((r + 1) // 2) ** 2 for r in range(11)]))
    n, nl = 100, 10
    els = [((n - i - 1) % nl) + ((n - i - 1) // nl) for i in range(n)]
    best = {i // nl: float(els[i] * els[i] * 4) for i in range(n) if i % nl == nl - 1}
    multi.append(dict(name="multi_find_better_score", src="test_hnsw_multi.cpp:217-259", dim=4, metric="L2",
                      vectors=[[float(e)] * 4 for e in els], labels=[i // nl for i in range(n)], n_labels=nl, query=[0.0] * 4, k=10,
                      expect_labels=[10 - r - 1 for r in range(10)], expect_scores=[best[10 - r - 1] for r in range(10)]))
    multi.append(dict(name="multi_find_better_score_after_pop", src="test_hnsw_multi.cpp:261-287", dim=4, metric="L2",
                      vectors=[[float(12 - i)] * 4 for i in range(12)], labels=[i % 3 for i in range(12)], n_labels=3, query=[0.0] * 4,
                      k=3, expect_labels=[3 - r - 1 for r in range(3)]))
    vecs, labs = [], []
    for i in range(1000):
        vecs.append([float(i)] * 4)
        labs.append(i)
        for _ in range(4):
            vecs.append([float(i + 5000)] * 4)
            labs.append(i)
    mrq = dict(name="multi_rangeQuery", src="test_hnsw_multi.cpp:1461-1510", dim=4, metric="L2", n_labels=1000, per_label=5, pivot=500,
               radius=4.0 * 5 ** 2, expect_count=11, expect_scores_by_score=[4.0 * ((r + 1) // 2) ** 2 for r in range(11)],
               expect_abs_diff_by_score=[(r + 1) // 2 for r in range(11)], expect_labels_by_id=[500 - 5 + r for r in range(11)],
               epsilons=[0.01, 1.0])
    return dict(topk=cases, range=rq, multi=multi, multi_range=mrq)


if __name__ == "__main__":
    with open(os.path.join(HERE, "kat_hnsw.json"), "w") as f:
        json.dump(hnsw(), f, indent=0)
    with open(os.path.join(HERE, "kat_sq8.json"), "w") as f:
        json.dump(sq8(), f, indent=1)
    with open(os.path.join(HERE, "kat_flat_multi.json"), "w") as f:
        json.dump(flat_multi(), f, indent=1)
    with open(os.path.join(HERE, "kat_spaces.json"), "w") as f:
        json.dump(spaces(), f, indent=0)
    with open(os.path.join(HERE, "kat_flat.json"), "w") as f:
        json.dump(flat(), f, indent=1)
    print("wrote kat_spaces.json, kat_flat.json")
