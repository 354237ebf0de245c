"""CPU: the oracle's scalar tier against tests/golden/ref_scalar_random.json -- numbers the REFERENCE'S OWN compiled code
produced on seeded random inputs (tests/golden/make_ref_scalar_random.py over oracle/_ref, built by oracle/build_ref.sh
from the reference's L2.cpp / IP.cpp / vecsim_malloc.cpp with plain g++).  Random inputs are summation-order sensitive,
so unlike the reference's `v[i] = i` known answers these pin the order as well.  Tolerance: none (bit patterns).

Which parts of the oracle this makes reference-executed: the scalar kernels (A3), normalisation (A9), the bf16 / fp16
conversions (every fp16 / bf16 input), scalar SQ8, and the sequential top-k containers (A10, multi-value included).
The AVX-512 tiers stay twin-checked (tests/test_oracle_kats.py: vso.c against intrinsics on the host's vector unit).
"""
import json
import os
import sys

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)
import refgen  # noqa: E402


@pytest.fixture(scope="module")
def fixture():
    with open(os.path.join(GOLD, "ref_scalar_random.json")) as f:
        return json.load(f)


class OracleScalar:
    """refgen backend over oracle/vso.py, scalar tier"""

    def __init__(self, vso):
        self.vso = vso
        self.f32_to_bf16, self.f32_to_f16 = vso.f32_to_bf16, vso.f32_to_f16
        self.bf16_to_f32, self.f16_to_f32 = vso.bf16_to_f32, vso.f16_to_f32

    def distance(self, t, m, a, b, dim):
        return self.vso.distance(t, m, a, b, dim, tier=self.vso.TIER_SCALAR)

    def normalize(self, blob, dim, t):
        return self.vso.normalize(blob, dim, t)

    def sq8_distance(self, kind, m, storage, query, dim):
        f = {"fp32": self.vso.sq8_fp32_distance, "fp16": self.vso.sq8_fp16_distance, "sq8": self.vso.sq8_sq8_distance}[kind]
        return f(m, storage, query, dim, tier=self.vso.TIER_SCALAR)

    def topk(self, scores, k, labels, multi, wide):
        if not wide:
            scores = np.asarray(scores, dtype=np.float64).astype(np.float32).astype(np.float64)
        return self.vso.topk_replay_multi(scores, k, labels) if multi else self.vso.topk_replay(scores, k, labels)


def _diff(name, got, want, key=lambda e: e):
    bad = [(key(w), g, w) for g, w in zip(got, want) if g != w]
    assert len(got) == len(want) and not bad, "%s: %d of %d differ, first: %r" % (name, len(bad), len(want), bad[:3])


def test_scalar_distance_kernels_equal_the_compiled_reference(vso, fixture):
    got = refgen.compute_distances(OracleScalar(vso))
    _diff("distances", got, fixture["distances"], key=lambda e: (e["type"], e["metric"], e["dim"]))
    assert len(got) == len(refgen.TYPES) * len(refgen.METRICS) * len(refgen.DIMS)


def test_conversions_equal_the_compiled_reference(vso, fixture):
    assert refgen.compute_conversions(OracleScalar(vso)) == fixture["conversions"]


def test_normalisation_equals_the_compiled_reference(vso, fixture):
    _diff("normalize", refgen.compute_normalize(OracleScalar(vso)), fixture["normalize"], key=lambda e: (e["type"], e["dim"]))


def test_scalar_sq8_kernels_equal_the_compiled_reference(vso, fixture):
    _diff("sq8", refgen.compute_sq8(OracleScalar(vso)), fixture["sq8"], key=lambda e: (e["kind"], e["metric"], e["dim"]))


def test_topk_replay_equals_the_reference_containers(vso, fixture):
    """brute_force.h:257-288 over vecsim_stl::max_priority_queue / updatable_max_heap, ties and NaNs included"""
    _diff("topk", refgen.compute_topk(OracleScalar(vso)), fixture["topk"], key=lambda e: e["case"])


def test_uint8_beyond_the_32_bit_bound_equals_the_compiled_reference(vso, fixture):
    """dim > 33 025: the reference's choosers hand back the scalar kernel with its 64-bit total (spaces.h:57-66)"""
    _diff("wide_u8", refgen.compute_wide_u8(OracleScalar(vso)), fixture["wide_u8"], key=lambda e: (e["dim"], e["kind"], e["metric"]))
    # the oracle's default tier resolves to the same exact integer there
    class Avx(OracleScalar):
        def distance(self, t, m, a, b, dim):
            return self.vso.distance(t, m, a, b, dim, tier=self.vso.TIER_AVX512)
    _diff("wide_u8 (avx512 tier)", refgen.compute_wide_u8(Avx(vso)), fixture["wide_u8"], key=lambda e: (e["dim"], e["kind"], e["metric"]))


def test_fixture_is_what_the_reference_build_produces_here(fixture):
    """where oracle/_ref exists (the build container; the .so also travels to the GPU box), re-run the reference's code on
    the seeds: the committed fixture must be reproducible, and a wider random sweep must agree with the oracle too"""
    from oracle import vsref
    if not vsref.available():
        pytest.skip("oracle/_ref/libvsref.so not built (needs /root/reference; sh oracle/build_ref.sh)")
    try:
        vsref.lib()
    except OSError as e:     # a library built against another libstdc++
        pytest.skip("oracle/_ref/libvsref.so does not load here: %s" % e)
    live = refgen.compute_all(vsref)
    for section in ("distances", "conversions", "normalize", "sq8", "topk", "wide_u8"):
        assert live[section] == fixture[section], section


def test_oracle_scalar_tier_equals_the_live_reference_on_more_dims(vso):
    from oracle import vsref
    if not vsref.available():
        pytest.skip("oracle/_ref/libvsref.so not built")
    try:
        vsref.lib()
    except OSError as e:
        pytest.skip(str(e))
    dims = (41, 63, 64, 65, 100, 255, 256, 257, 513, 1000, 1536, 2048, 4099)
    a = refgen.compute_distances(vsref, dims)
    b = refgen.compute_distances(OracleScalar(vso), dims)
    _diff("live distances", b, a, key=lambda e: (e["type"], e["metric"], e["dim"]))
    a = refgen.compute_sq8(vsref, (9, 100, 257, 1536))
    b = refgen.compute_sq8(OracleScalar(vso), (9, 100, 257, 1536))
    _diff("live sq8", b, a, key=lambda e: (e["kind"], e["metric"], e["dim"]))


def test_product_merge_replays_like_the_reference_containers(fixture):
    """the PRODUCT's host replay (libvecsim_amd.so: VecSimGpu_MergeTopK -> merge_topk over RefMaxHeap, host/ref_heap.h; no GPU
    involved) on the fixture's tie-heavy score streams: labels, order and scores equal what the reference's own
    max_priority_queue produced.  (NaN streams take the every-row replay of a live index: tests/test_gpu_flat_parity.py.)"""
    from vectorsimilarity_amd.sharded import merge_topk
    n_checked = 0
    for e in fixture["topk"]:
        n, k, levels, multi, wide, nan_every, distinct = e["case"]
        if multi or nan_every:
            continue
        scores, labels = refgen.topk_inputs(tuple(e["case"]))
        if not wide:
            scores = scores.astype(np.float32).astype(np.float64)
        kk = min(k, n)
        # two "shards": even and odd ids, as a sharded index would hold them; gid = id
        for parts in (1, 2):
            cap = n
            gids = np.zeros((parts, 1, cap), dtype=np.uint64)
            labs = np.zeros((parts, 1, cap), dtype=np.uint64)
            scs = np.zeros((parts, 1, cap), dtype=np.float64)
            counts = np.zeros((parts, 1), dtype=np.uint32)
            for p in range(parts):
                ids = np.arange(p, n, parts)
                counts[p, 0] = len(ids)
                gids[p, 0, :len(ids)] = ids
                labs[p, 0, :len(ids)] = labels[ids]
                scs[p, 0, :len(ids)] = scores[ids]
            ol, osc = merge_topk(counts, gids, labs, scs, kk)
            assert [int(x) for x in ol[0]] == e["labels"], (e["case"], parts)
            assert refgen.hexbits(osc[0]) == e["scores"], (e["case"], parts)
        n_checked += 1
    assert n_checked >= 8


def test_product_heaps_equal_the_reference_containers_on_scripts():
    """vectorsimilarity_amd/csrc/host/ref_heap.h -- the min-heap order, the max-heap order and the label-keyed updatable heap the HNSW
    batch iterator's walk, the Flat replay and the sharded merge keep their state in -- against the reference's OWN containers
    (utils/vecsim_stl.h:63-89, utils/updatable_heap.h:20-113, compiled as gnu++20 into oracle/_ref): random scripts of emplace / pop
    with ties, +-inf and (std::priority_queue kinds) NaN scores; size and top after every operation
    (tests/golden/ref_heap_scripts.json <- make_ref_heap_scripts.py).  With the reference build present the live library is run too."""
    import ctypes as C
    import importlib.util
    import subprocess
    helpers = os.path.join(os.path.dirname(__file__), "helpers")
    so, src = os.path.join(helpers, "libref_heap_probe.so"), os.path.join(helpers, "ref_heap_probe.cpp")
    hdr = os.path.join(os.path.dirname(__file__), "..", "vectorsimilarity_amd", "csrc", "host", "ref_heap.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", so, src], check=True)
    probe = C.CDLL(so).heap_script
    spec = importlib.util.spec_from_file_location("make_ref_heap_scripts", os.path.join(GOLD, "make_ref_heap_scripts.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    with open(os.path.join(GOLD, "ref_heap_scripts.json")) as f:
        cases = json.load(f)["cases"]
    assert [(c["seed"], c["kind"], c["n"]) for c in cases] == gen.CASES
    live = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libvsref.so")
    ref = C.CDLL(live).vsref_heap_script if os.path.exists(live) else None
    nan_tops = 0
    for c in cases:
        op, score, label = gen.script(c["seed"], c["kind"], c["n"])
        sz, ts, tl = gen.run(probe, c["kind"], op, score, label)
        want_ts = np.array(c["top_score_bits"], dtype=np.uint64).view(np.float64)
        assert sz.tolist() == c["size"], (c["seed"], c["kind"])
        assert np.array_equal(ts, want_ts, equal_nan=True), (c["seed"], c["kind"], np.nonzero(~((ts == want_ts) | (np.isnan(ts) & np.isnan(want_ts))))[0][:5])
        assert tl.tolist() == c["top_label"], (c["seed"], c["kind"])
        nan_tops += int(np.isnan(want_ts[np.array(c["size"]) > 0]).sum())
        if ref is not None:
            rsz, rts, rtl = gen.run(ref, c["kind"], op, score, label)
            assert rsz.tolist() == c["size"] and rtl.tolist() == c["top_label"] and np.array_equal(rts, want_ts, equal_nan=True)
    assert nan_tops > 0   # (NaN scores did reach the top of a heap: the gnu++20 pair order was exercised)
