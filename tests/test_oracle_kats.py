"""CPU: pins the oracle against the reference's own known-answer tests (tests/golden/kat_*.json,
made by tests/golden/make_kats.py from the reference's unit tests) and cross-checks the portable
lane emulation against an independent AVX-512 intrinsics implementation on the host CPU."""
import json
import os

import numpy as np
import pytest

from util import METRICS, TIERS, TYPES, encode

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def test_distance_kats(vso):
    kats = _load("kat_spaces.json")["distance"]
    assert len(kats) > 500
    for c in kats:
        a = encode(vso, c["a"], c["type"])
        b = encode(vso, c["b"], c["type"])
        got = vso.distance(TYPES[c["type"]], METRICS[c["metric"]], a, b, dim=len(c["a"]), tier=TIERS[c["tier"]])
        assert got == c["expect"], (c["name"], got, c["expect"], c["src"])


def test_normalize_kats(vso):
    for c in _load("kat_spaces.json")["normalize"]:
        v = encode(vso, c["input"], c["type"])
        vso.normalize(v, len(c["input"]), TYPES[c["type"]])
        wide = [vso.lib().vso_bf16_to_f32(int(x)) if c["type"] == "bf16" else vso.lib().vso_f16_to_f32(int(x)) for x in v]
        assert wide == c["expect"], c["name"]


def test_int_cosine_self_distance(vso):
    # test_spaces.cpp:286-322: Cosine(v, v) ~ 0 with the norm stored after the elements
    rng = np.random.default_rng(123)
    for typ, t in (("i8", 4), ("u8", 5)):
        v = np.zeros(8, dtype=np.uint8)
        v[:4] = rng.integers(0, 256, 4, dtype=np.uint8)
        vso.normalize(v, 4, t)
        assert abs(vso.distance(t, 2, v, v, dim=4)) < 1e-6


def test_tie_semantics_probe(vso):
    c = [x for x in _load("kat_flat.json")["cases"] if x["name"] == "tie_probe"][0]
    labels = np.array([p[0] for p in c["scan"]], dtype=np.uint64)
    scores = np.array([p[1] for p in c["scan"]], dtype=np.float64)
    for k, exp in c["expect"].items():
        got, _ = vso.topk_replay(scores, int(k), labels)
        assert list(got) == exp


def _flat_case(vso, c, typ):
    n, dim, k = c["n"], c["dim"], c["k"]
    t, m = TYPES[typ], METRICS[c["metric"]]
    if c["name"] == "bf_cosine":
        rows = np.ones((n, dim))
        rows[:, 0] = np.arange(1, n + 1) / n
        labels = np.arange(1, n + 1, dtype=np.uint64)
        q = np.ones(dim)
    else:
        rows = np.repeat(np.arange(n)[:, None], dim, axis=1).astype(np.float64)
        labels = np.arange(n, dtype=np.uint64)
        q = np.full(dim, c["query_value"], dtype=np.float64)
    enc = np.stack([encode(vso, r, typ) for r in rows])
    qe = encode(vso, q, typ)
    if c["metric"] == "Cosine":
        for i in range(n):
            vso.normalize(enc[i], dim, t)
        vso.normalize(qe, dim, t)
    got_l, got_s = vso.flat_topk(t, 1 if c["metric"] == "Cosine" else m, enc, qe, k, dim, labels)
    return got_l, got_s


def test_flat_kats(vso):
    for c in _load("kat_flat.json")["cases"]:
        if c["name"] == "tie_probe":
            continue
        for typ in c["types"]:
            got_l, got_s = _flat_case(vso, c, typ)
            if c.get("order") == "BY_ID":
                o = np.argsort(got_l, kind="stable")
                got_l, got_s = got_l[o], got_s[o]
            if "expect_labels" in c:
                assert list(got_l) == c["expect_labels"], (c["name"], typ)
            if "expect_scores" in c:
                assert list(got_s) == c["expect_scores"], (c["name"], typ)
            if "expect_absdiff" in c:
                assert [abs(int(x) - 50) for x in got_l] == c["expect_absdiff"], (c["name"], typ)
            if "expect_id_set" in c:
                assert sorted(int(x) for x in got_l) == c["expect_id_set"], (c["name"], typ)


def test_multi_value_replay_kats(vso):
    """the label-keyed replay (vso_topk_replay_multi, restating utils/updatable_heap.h:20-113 +
    brute_force_multi.h:108-277) against the closed forms of tests/unit/test_bruteforce_multi.cpp"""
    for c in _load("kat_flat_multi.json")["cases"]:
        dim = c["dim"]
        labels = np.array([v[0] for v in c["vectors"]], dtype=np.uint64)
        rows = np.array([[v[1]] * dim for v in c["vectors"]], dtype=np.float32)
        scores = vso.scan(0, 0, rows, np.zeros(dim, np.float32), dim)
        got_l, got_s = vso.topk_replay_multi(scores, c["k"], labels)
        assert [int(x) for x in got_l] == c["expect_labels"], c["name"]
        if "expect_scores" in c:
            assert list(got_s) == c["expect_scores"], c["name"]


def test_lanes_match_avx512_intrinsics(vso):
    """portable 32-lane emulation == hand-written AVX-512 intrinsics (vso_fast.c), bit for bit"""
    if not vso.lib().vso_has_avx512():
        pytest.skip("host CPU has no AVX-512F")
    rng = np.random.default_rng(7)
    for d in list(range(8, 200)) + [512, 768, 1000, 1024, 1536]:
        a = rng.uniform(-1, 1, d).astype(np.float32)
        b = rng.uniform(-1, 1, d).astype(np.float32)
        for m in (0, 1):
            assert vso.distance(0, m, a, b) == vso.distance_fast(0, m, a, b), (d, m)


@pytest.mark.parametrize("typ,tier", [("f64", "avx512"), ("f16", "avx512"), ("bf16", "avx512"), ("bf16", "avx512_bf16"),
                                      ("i8", "avx512"), ("u8", "avx512")])
def test_every_tier_matches_its_intrinsics_twin(vso, typ, tier):
    """the restatement of each AVX-512 tier (vso.c: lanes, association, reduce tree) against the same published algorithm run
    on the host CPU's own vector unit (vso_fast.c: vcvtph2ps, vpexpandw / vpunpck, vdpbf16ps, vpdpwssd), bit for bit on random
    data, every residual class of dim from the tier's minimum up, plus the BASELINE widths.  Order-sensitive: a wrong lane map
    or reduce tree changes last bits on these inputs (the reference's own v[i] = i tests cannot see that)."""
    from util import METRICS, TIERS, TYPES, random_vectors, stored_rows
    t = TYPES[typ]
    lo = {"f64": 4, "f16": 16, "bf16": 32, "i8": 32, "u8": 32}[typ]
    step = {"f64": 16, "f16": 32, "bf16": 32, "i8": 64, "u8": 64}[typ]
    if not vso.fast_available(t, 1, lo + step, TIERS[tier]):
        pytest.skip("host CPU lacks the instructions of this tier")
    rng = np.random.default_rng(sum(map(ord, typ + tier)))
    checked = 0
    for d in list(range(lo, lo + 2 * step + 1)) + [200, 513, 768, 1000, 1024, 1536]:
        rows = random_vectors(rng, 4, d, typ, vso)
        for metric in ("L2", "IP", "Cosine"):
            if metric == "Cosine" and typ not in ("i8", "u8"):
                continue   # (fp Cosine is the IP kernel on normalised blobs)
            st = stored_rows(vso, rows, typ, metric)
            for j in (1, 2, 3):
                want = vso.distance(t, METRICS[metric], st[0], st[j], d, TIERS[tier])
                got = vso.distance_fast(t, METRICS[metric], st[0], st[j], d, TIERS[tier])
                assert want == got or (np.isnan(want) and np.isnan(got)), (typ, tier, metric, d, want, got)
                checked += 1
    assert checked > 100


def test_f16c_restatement_matches_the_hosts_f16c_unit(vso):
    """fp16 F16C tier (the reference's kernel for dims 8..15 on an AVX-512 host, L2_space.cpp:404-409): the portable
    restatement equals the same published algorithm run on this host's vcvtph2ps/fmadd units, bit for bit, at every
    residual class; and the chooser sends dims 8..15 there, 16+ to the AVX512F order, < 8 to the scalar order"""
    L = vso.lib()
    if not L.vso_has_f16c():
        pytest.skip("host CPU has no F16C/FMA/AVX2")
    import ctypes as C
    rng = np.random.default_rng(11)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for d in list(range(8, 140)) + [768]:
        for rep in range(3):
            a = rng.uniform(-1, 1, d).astype(np.float16).view(np.uint16)
            b = rng.uniform(-1, 1, d).astype(np.float16).view(np.uint16)
            for m in (0, 1):
                sw, hw = L.vso_f16c_distance(m, d, p(a), p(b)), L.vso_f16c_distance_hw(m, d, p(a), p(b))
                assert sw == hw, (d, m, sw, hw)
                if d < 16:
                    assert vso.distance(3, m, a, b) == hw, (d, m)
    differs = 0
    for _ in range(200):   # ... and that order is NOT the scalar one the previous round used there
        a = rng.uniform(-1, 1, 15).astype(np.float16).view(np.uint16)
        b = rng.uniform(-1, 1, 15).astype(np.float16).view(np.uint16)
        differs += vso.distance(3, 1, a, b, tier=0) != vso.distance(3, 1, a, b, tier=1)
    assert differs > 0


def test_scalar_and_lane_orders_differ_in_last_bits(vso):
    # documents why the oracle must fix a tier (SURVEY.md hard part 1)
    rng = np.random.default_rng(3)
    diff = 0
    for _ in range(50):
        a = rng.uniform(-1, 1, 768).astype(np.float32)
        b = rng.uniform(-1, 1, 768).astype(np.float32)
        diff += vso.distance(0, 0, a, b, tier=0) != vso.distance(0, 0, a, b, tier=1)
    assert diff > 0


def test_dpbf16_emulation_matches_hardware(vso):
    """vdpbf16ps characterisation: acc += odd product (rounded), then even product (rounded)"""
    L = vso.lib()
    if not L.vso_has_avx512_bf16():
        pytest.skip("host CPU has no avx512_bf16")
    import ctypes as C
    rng = np.random.default_rng(11)
    for d in (32, 40, 64, 96, 127, 768):
        for _ in range(20):
            x = vso.f32_to_bf16(rng.uniform(-1, 1, d).astype(np.float32))
            y = vso.f32_to_bf16(rng.uniform(-1, 1, d).astype(np.float32))
            # hardware: emulate the kernel structure with the probe primitive
            acc = np.zeros(16, dtype=np.float32)
            r = d % 32
            pos = 0
            if r:
                xx = np.zeros(32, dtype=np.uint16); yy = np.zeros(32, dtype=np.uint16)
                xx[:r] = x[:r]; yy[:r] = y[:r]
                L.vso_probe_dpbf16(acc.ctypes.data_as(C.c_void_p), xx.ctypes.data_as(C.c_void_p), yy.ctypes.data_as(C.c_void_p))
                pos = r
            while pos < d:
                xx = np.ascontiguousarray(x[pos:pos + 32]); yy = np.ascontiguousarray(y[pos:pos + 32])
                L.vso_probe_dpbf16(acc.ctypes.data_as(C.c_void_p), xx.ctypes.data_as(C.c_void_p), yy.ctypes.data_as(C.c_void_p))
                pos += 32
            t = acc[8:] + acc[:8]
            u = t[4:] + t[:4]
            hw = np.float32(1.0) - ((u[0] + u[2]) + (u[1] + u[3]))
            assert vso.distance(2, 1, x, y, dim=d, tier=2) == float(hw), d


def test_synthetic_generator_is_uniform_and_exact(vso):
    rows = vso.synth_rows_f32(47, 0, 64, 768)
    assert rows.min() >= -1.0 and rows.max() < 1.0
    assert abs(rows.mean()) < 0.01 and abs(rows.var() - 1 / 3) < 0.01
    again = vso.synth_rows_f32(47, 10, 2, 768)
    assert np.array_equal(again, rows[10:12])


# ---------------------------------------------------------------- SQ8 (oracle/vso_sq8.c)
SQ8_METRIC = {"L2": 0, "IP": 1, "Cosine": 2}


def _ulps(a, b):
    ia = np.array([a], dtype=np.float32).view(np.int32)[0]
    ib = np.array([b], dtype=np.float32).view(np.int32)[0]
    return abs(int(ia) - int(ib))


def test_sq8_quantizer_kats(vso):
    """QuantPreprocessor known answers (test_components.cpp:1047-1596): code bytes, metadata, query metadata."""
    kats = _load("kat_sq8.json")["quantize"]
    assert len(kats) >= 12
    for c in kats:
        m = SQ8_METRIC[c["metric"]]
        x = np.array(c["input"], dtype=np.float32)
        blob = vso.sq8_quantize(x, m)
        dim = x.size
        assert blob.size == dim + (16 if c["metric"] == "L2" else 12), c["name"]
        assert blob[:dim].tolist() == c["bytes"], (c["name"], blob[:dim].tolist())
        meta = blob[dim:].view(np.float32)
        if c.get("finite_positive_delta"):
            assert np.isfinite(meta[0]) and np.isfinite(meta[1]) and meta[1] > 0, c["name"]
        if "min" in c:
            names = ["min", "delta", "sum"] + (["sum_squares"] if c["metric"] == "L2" else [])
            for i, n in enumerate(names):
                exp = np.float32(c[n])
                if c["exact_meta"]:   # the reference compares the whole blob bytewise with its helper
                    assert meta[i] == exp, (c["name"], n, meta[i], exp)
                else:                 # ASSERT_FLOAT_EQ
                    assert _ulps(meta[i], exp) <= 4, (c["name"], n, meta[i], exp)
        if "query_sum" in c:
            qb = vso.sq8_query_blob(x, m)
            assert qb.size == dim + (2 if c["metric"] == "L2" else 1)
            assert np.array_equal(qb[:dim], x)
            assert _ulps(qb[dim], np.float32(c["query_sum"])) <= 4
            if c["metric"] == "L2":
                assert _ulps(qb[dim + 1], np.float32(c["query_sum_squares"])) <= 4


def test_sq8_distance_kats(vso):
    for c in _load("kat_sq8.json")["distance"]:
        dim = c["dim"]
        st = np.zeros(dim + 16, dtype=np.uint8)
        st[:dim] = c["codes"]
        st[dim:].view(np.float32)[:] = c["meta"]
        q = np.array(c["query"] + c["qmeta"], dtype=np.float32)
        if c.get("query_type") == "f16":   # fp16 values, then the FP32 metadata at an unaligned offset
            qb = np.zeros(dim * 2 + 8, dtype=np.uint8)
            qb[:dim * 2].view(np.uint16)[:] = vso.f32_to_f16(np.array(c["query"], dtype=np.float32))
            qb[dim * 2:] = np.array(c["qmeta"], dtype=np.float32).view(np.uint8)
            for tier in (vso.TIER_AVX512, vso.TIER_SCALAR):
                assert vso.sq8_fp16_distance(SQ8_METRIC[c["metric"]], st, qb, dim, tier=tier) == c["expect"], (c["name"], tier)
            continue
        for tier in (vso.TIER_AVX512, vso.TIER_SCALAR):
            got = vso.sq8_fp32_distance(SQ8_METRIC[c["metric"]], st, q, dim, tier=tier)
            assert got == c["expect"], (c["name"], tier, got)


def test_sq8_mean_centred_blob_kats(vso):
    """QuantPreprocessor<..., WithNorm = true> (test_components.cpp:1820-2020): blob sizes, storage == the plain quantiser
    on x - mean (+ x_mean_ip for IP), query body (centred for L2), query metadata; fp32 and fp16 inputs."""
    for c in _load("kat_sq8.json")["with_norm"]["blobs"]:
        m = SQ8_METRIC[c["metric"]]
        x = np.array(c["input"], dtype=np.float32)
        mean = np.array(c["mean"], dtype=np.float32)
        dim = x.size
        for f16 in (False, True):
            xin = vso.f32_to_f16(x) if f16 else x
            blob = vso.sq8_quantize_norm(xin, mean, m, f16=f16)
            assert blob.size == c["storage_bytes"]
            base = vso.sq8_quantize(np.array(c["centred"], dtype=np.float32), m)
            assert np.array_equal(blob[:base.size], base), c["name"]          # CompareVectors on the common prefix
            meta = blob[dim:].view(np.float32)
            if c["metric"] == "IP":
                assert _ulps(meta[3], np.float32(c["x_mean_ip"])) <= 4
            qb = vso.sq8_query_blob_norm(xin, mean, m, f16=f16)
            eb = 2 if f16 else 4
            assert qb.size == dim * eb + 8 and (f16 or qb.size == c["query_bytes"])
            body = qb[:dim * eb].view(np.uint16 if f16 else np.float32)
            want = np.array(c["query_body"], dtype=np.float32)
            if f16:
                assert np.array_equal(body, vso.f32_to_f16(want)), c["name"]
                continue                                                            # (metadata of the re-rounded body: below, fp32 only)
            assert np.array_equal(body, want), c["name"]
            qm = qb[dim * eb:].view(np.float32)
            assert _ulps(qm[0], np.float32(c["y_sum"])) <= 4
            other = c["y_sum_squares"] if c["metric"] == "L2" else c["y_mean_ip"]
            assert _ulps(qm[1], np.float32(other)) <= 4


def test_sq8_mean_centred_calculator_kats(vso):
    """DistanceCalculatorWithNorm (test_components.cpp:2095-2415): asymmetric and symmetric distances land within the
    tolerance the reference states of the brute-force distance between the ORIGINAL vectors; with a zero mean the
    results are the base kernels' bit for bit."""
    kats = _load("kat_sq8.json")["with_norm"]
    for c in kats["calculator"]:
        m = SQ8_METRIC[c["metric"]]
        x, y, mean = (np.array(c[k], dtype=np.float32) for k in ("x", "y", "mean"))
        dim = x.size
        f16 = c["input_type"] == "f16"
        xin, yin = (vso.f32_to_f16(x), vso.f32_to_f16(y)) if f16 else (x, y)
        sb = vso.sq8_quantize_norm(xin, mean, m, f16=f16)
        for tier in (vso.TIER_AVX512, vso.TIER_SCALAR):
            if c["mode"] == "asymmetric":
                got = vso.sq8_distance_norm(m, sb, vso.sq8_query_blob_norm(yin, mean, m, f16=f16), dim, f16=f16, tier=tier)
            else:
                got = vso.sq8_sq8_distance_norm(m, sb, vso.sq8_quantize_norm(yin, mean, m, f16=f16), dim, c["mean_sum_squares"], tier=tier)
            assert abs(got - c["expect"]) <= c["tol"], (c["name"], tier, got, c["expect"])
            assert not c.get("nonnegative") or got >= 0.0
    z = kats["zero_mean"]
    x, y = np.array(z["x"], dtype=np.float32), np.array(z["y"], dtype=np.float32)
    zero = np.zeros(x.size, dtype=np.float32)
    sb, sy = vso.sq8_quantize_norm(x, zero, 1), vso.sq8_quantize_norm(y, zero, 1)
    qb = vso.sq8_query_blob_norm(y, zero, 1)
    assert vso.sq8_distance_norm(1, sb, qb, x.size) == vso.sq8_fp32_distance(1, sb, qb.view(np.float32), x.size)
    assert vso.sq8_sq8_distance_norm(1, sb, sy, x.size, 0.0) == vso.sq8_sq8_distance(1, sb, sy, x.size)


@pytest.mark.parametrize("dim", [1, 5, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 128, 777])
def test_sq8_tiers_within_the_reference_tolerance(vso, dim):
    """The reference's SQ8 tests (test_spaces.cpp:326-410, 2330-4111) demand every tier within 0.01 of the
    reconstruct-then-dot baseline on random data; the restated scalar and AVX-512 orders must satisfy the same."""
    tol = _load("kat_sq8.json")["tolerance_property"]["abs"]
    rng = np.random.default_rng(dim)
    for metric in (0, 1, 2):
        x = rng.uniform(-1, 1, dim).astype(np.float32)
        y = rng.uniform(-1, 1, dim).astype(np.float32)
        z = rng.uniform(-1, 1, dim).astype(np.float32)
        if metric == 2:
            x /= np.linalg.norm(x)
            y /= np.linalg.norm(y)
            z /= np.linalg.norm(z)
        st, st2 = vso.sq8_quantize(x, metric), vso.sq8_quantize(z, metric)
        qb = vso.sq8_query_blob(y, metric)
        meta, meta2 = st[dim:].view(np.float32), st2[dim:].view(np.float32)
        xr = meta[0] + meta[1] * st[:dim].astype(np.float64)
        zr = meta2[0] + meta2[1] * st2[:dim].astype(np.float64)
        base = float(np.sum((xr - y) ** 2)) if metric == 0 else 1.0 - float(np.dot(xr, y))
        base2 = float(np.sum((xr - zr) ** 2)) if metric == 0 else 1.0 - float(np.dot(xr, zr))
        for tier in (vso.TIER_AVX512, vso.TIER_SCALAR):
            assert abs(vso.sq8_fp32_distance(metric, st, qb, dim, tier=tier) - base) < tol, (metric, tier)
            assert abs(vso.sq8_sq8_distance(metric, st, st2, dim, tier=tier) - base2) < tol, (metric, tier)
        # the stored sums describe the reconstruction (preprocessors.h:369-381)
        assert abs(meta[2] - xr.sum()) <= 1e-4 * max(1.0, abs(xr.sum()))


def _hnsw_graph(z, name):
    g = {}
    for key in ("n", "M", "M0", "entry", "max_level"):
        g[key] = int(z["%s/%s" % (name, key)])
    for key in ("links0", "cnt0", "upper_off", "upper", "deleted", "labels"):
        g[key] = np.ascontiguousarray(z["%s/%s" % (name, key)])
    return g, np.ascontiguousarray(z["%s/stored" % name])


def test_hnsw_search_oracle_on_the_reference_known_answers(vso):
    """oracle/vso_hnsw.c searching the graphs of tests/golden/kat_hnsw_graphs.npz (the product's host builder on the inputs of
    tests/unit/test_hnsw.cpp, exported by tests/golden/make_hnsw_graphs.py on an MI355X) must give the closed forms the
    reference's own tests assert (kat_hnsw.json) -- the same pin tests/test_gpu_hnsw.py applies, here without a GPU."""
    z = np.load(os.path.join(GOLD, "kat_hnsw_graphs.npz"))
    kats = _load("kat_hnsw.json")
    assert len(kats["topk"]) >= 4
    for case in kats["topk"]:
        g, srows = _hnsw_graph(z, case["name"])
        assert g["n"] == len(case["vectors"]) and list(g["labels"]) == case["labels"]
        q = np.array(case["query"], dtype=np.float32)
        rows = np.array(case["vectors"], dtype=np.float32)
        if case["metric"] == "Cosine":      # stored blobs are the oracle's own normalisation of the inputs, bit for bit
            vso.normalize(q, case["dim"], 0)
            for r in rows:
                vso.normalize(r, case["dim"], 0)
        assert srows.tobytes() == rows.tobytes(), case["name"]
        k = case["k"]
        ol, od, _ = vso.hnsw_search(0, 0 if case["metric"] == "L2" else 1, srows, g, q, k, max(k, 10), case["dim"])
        if case["order"] == "id":
            srt = np.argsort(ol, kind="stable")
            ol, od = ol[srt], od[srt]
        assert len(ol) == k, case["name"]
        if "expect_labels" in case:
            assert [int(x) for x in ol] == case["expect_labels"], case["name"]
        if "expect_abs_diff" in case:
            assert [abs(int(x) - case["expect_labels_abs_diff_from"]) for x in ol] == case["expect_abs_diff"], case["name"]
            assert list(od) == case["expect_scores"], case["name"]
        if "expect_label_range" in case:
            lo, hi = case["expect_label_range"]
            assert all(lo <= int(x) < hi for x in ol) and all(float(s) <= case["expect_score_max"] for s in od), case["name"]


def test_hnsw_batch_iterator_oracle_on_the_fixture_graphs(vso):
    """oracle/vso_hnsw.c's twin of the batch iterator's walk (hnsw_batch_iterator.h:96-230) on the fixture graphs: a first batch
    of k <= ef equals the top-k search (same admission rule, same stop), batches never repeat a label and come in non-decreasing
    score order on these line-shaped inputs, and the Cosine case -- no tied distances -- hands out all 100 vectors in exact order.
    (With tied distances the walk LOSES entries, upstream as here: a popped candidate whose distance equals lower_bound while
    top_candidates is full goes neither to the heap nor to the extras, hnsw_single_batch_iterator.h:62-80.)"""
    z = np.load(os.path.join(GOLD, "kat_hnsw_graphs.npz"))
    for case in _load("kat_hnsw.json")["topk"]:
        g, srows = _hnsw_graph(z, case["name"])
        q = np.array(case["query"], dtype=np.float32)
        km = 0 if case["metric"] == "L2" else 1
        if case["metric"] == "Cosine":
            vso.normalize(q, case["dim"], 0)
        for k in (1, 5, 10):
            out, _ = vso.hnsw_iterate(0, km, srows, g, q, 10, [k], case["dim"])
            ol, od, _ = vso.hnsw_search(0, km, srows, g, q, k, 10, case["dim"])
            assert np.array_equal(out[0][0], ol) and np.array_equal(out[0][1], od), (case["name"], k)
        out, depleted = vso.hnsw_iterate(0, km, srows, g, q, 10, [7] * (g["n"] // 7 + 3), case["dim"])
        labs = np.concatenate([l for l, _ in out])
        sc = np.concatenate([s for _, s in out])
        assert depleted and len(set(labs.tolist())) == len(labs) and np.all(np.diff(sc) >= 0), case["name"]
        if case["name"] == "testCosine":
            exact = vso.scan(0, km, srows, q, case["dim"])
            assert np.array_equal(labs, g["labels"][np.lexsort((g["labels"], exact))])
        else:
            assert len(labs) <= g["n"]


def test_hnsw_range_oracle_on_the_reference_known_answers(vso):
    z = np.load(os.path.join(GOLD, "kat_hnsw_graphs.npz"))
    c = _load("kat_hnsw.json")["range"]
    g, srows = _hnsw_graph(z, "rangeQuery")
    assert g["n"] == c["n"]
    q = np.full(c["dim"], float(c["pivot"]), dtype=np.float32)
    for eps in c["epsilons"]:
        ol, od, _ = vso.hnsw_range(0, 0, srows, g, q, c["radius"], eps, c["dim"])
        assert len(ol) == c["expect_count"]
        assert sorted(int(x) for x in ol) == c["expect_labels_by_id"]
        srt = np.lexsort((ol, od))
        assert [abs(int(x) - c["pivot"]) for x in ol[srt]] == c["expect_abs_diff_by_score"]
        assert list(od[srt]) == c["expect_scores_by_score"]


def test_topk_replay_follows_libstdcxx_heap_moves_with_nan_scores(vso):
    """The reference keeps its top-k in std::priority_queue<pair<score, label>> (utils/vecsim_stl.h:66-72).  With NaN scores
    the pair order is no strict weak order and the reply depends on the heap algorithm's own moves, so the oracle restates
    libstdc++'s (oracle/vso.c heap_push_hole / heap_pop).  Checked here against the real container on scores with NaNs, +-Inf
    and ties; labels both ascending and shuffled."""
    import ctypes as C
    import subprocess
    helpers = os.path.join(os.path.dirname(__file__), "helpers")
    so, src = os.path.join(helpers, "libheap_probe.so"), os.path.join(helpers, "heap_probe.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["g++", "-O1", "-std=gnu++20", "-fPIC", "-shared", "-o", so, src], check=True)
    L = C.CDLL(so)
    L.heap_probe_topk.restype = C.c_size_t
    L.heap_probe_topk.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(77)
    cases = 0
    for trial in range(400):
        n = int(rng.integers(1, 120))
        k = int(rng.integers(1, 40))
        sc = rng.integers(-6, 7, n).astype(np.float64) if trial % 2 else rng.normal(size=n)
        for frac, val in ((0.15, np.nan), (0.05, np.inf), (0.05, -np.inf)):
            sc[rng.random(n) < frac * (trial % 5)] = val
        labels = np.arange(n, dtype=np.uint64) if trial % 3 else rng.permutation(n).astype(np.uint64) * 7
        ol, osc = np.zeros(max(k, 1), dtype=np.uint64), np.zeros(max(k, 1), dtype=np.float64)
        c = L.heap_probe_topk(sc.ctypes.data, labels.ctypes.data, n, k, ol.ctypes.data, osc.ctypes.data)
        gl, gs = vso.topk_replay(sc, k, labels)
        assert len(gl) == c == min(n, k)
        assert np.array_equal(gl, ol[:c]), (trial, sc, gl, ol[:c])
        assert np.array_equal(gs, osc[:c], equal_nan=True)
        cases += bool(np.isnan(sc).any())
    assert cases > 200


def test_reference_random_int_vectors_exact_answers(vso):
    """the reference's own random int8 / uint8 test inputs (tests/unit/test_spaces.cpp:1574-2050: mt19937 seeds 123 / 1234 through
    tests/utils/tests_utils.h:25-49, regenerated by tests/golden/make_ref_random_kats.py) against answers computed in exact
    integer arithmetic: every tier the oracle models (scalar, AVX-512 VNNI) and the host's own VNNI unit give exactly those"""
    import importlib.util
    import json
    import os
    from util import TIERS
    here = os.path.join(os.path.dirname(__file__), "golden")
    spec = importlib.util.spec_from_file_location("make_ref_random_kats", os.path.join(here, "make_ref_random_kats.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    with open(os.path.join(here, "kat_ref_random_ints.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) == 202
    for c in cases:
        signed = c["type"] == "i8"
        t = 4 if signed else 5
        dim = c["dim"]
        a = gen.populate(c["seed_a"], dim, signed).astype(np.int8 if signed else np.uint8)
        b = gen.populate(c["seed_b"], dim, signed).astype(np.int8 if signed else np.uint8)
        assert [int(x) for x in a[:4]] == c["head_a"] and [int(x) for x in b[:4]] == c["head_b"]
        na = np.float32(np.sqrt(np.float64(c["sum_sq_a"])))   # compute_norm.h:18-31: sqrt(uint64) in double, stored as float
        nb = np.float32(np.sqrt(np.float64(c["sum_sq_b"])))
        cos = np.float32(1.0) - np.float32(c["dot"]) / (na * nb)   # IP.cpp:264-271, float32 operations
        sa = np.zeros(dim + 4, dtype=np.uint8)
        sb = np.zeros(dim + 4, dtype=np.uint8)
        sa[:dim], sb[:dim] = a.view(np.uint8), b.view(np.uint8)
        vso.normalize(sa, dim, t)
        vso.normalize(sb, dim, t)
        assert sa[dim:].view(np.float32)[0] == na and sb[dim:].view(np.float32)[0] == nb
        for tier in ("scalar", "avx512"):
            assert vso.distance(t, 0, a, b, dim, TIERS[tier]) == float(np.float32(c["l2"])), (c["type"], dim, tier)
            assert vso.distance(t, 1, a, b, dim, TIERS[tier]) == float(np.float32(c["ip"])), (c["type"], dim, tier)
            assert vso.distance(t, 2, sa, sb, dim, TIERS[tier]) == float(cos), (c["type"], dim, tier)
        if vso.fast_available(t, 0, dim):
            assert vso.distance_fast(t, 0, a, b, dim) == float(np.float32(c["l2"]))
            assert vso.distance_fast(t, 1, a, b, dim) == float(np.float32(c["ip"]))
            assert vso.distance_fast(t, 2, sa, sb, dim) == float(cos)
