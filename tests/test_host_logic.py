"""CPU: host-side logic of the product that needs no GPU -- blob preprocessing (VecSim_Normalize),
blob sizes, runtime-parameter resolution helpers -- against the oracle."""
import ctypes as C

import numpy as np
import pytest

from util import random_vectors
from vectorsimilarity_amd import VecSim, _capi

TCODE = {"f32": 0, "f64": 1, "bf16": 2, "f16": 3, "i8": 4, "u8": 5}


def test_normalize_matches_oracle_bitwise(vso):
    rng = np.random.default_rng(21)
    for typ, t in TCODE.items():
        for d in (1, 3, 4, 7, 16, 33, 128, 768):
            v = random_vectors(rng, 1, d, typ, vso)[0]
            ours = VecSim.normalize(v, t)
            if typ in ("i8", "u8"):
                ref = np.zeros(d + 4, dtype=np.uint8)
                ref[:d] = v.view(np.uint8)
            else:
                ref = v.copy()
            vso.normalize(ref, d, t)
            assert np.array_equal(np.asarray(ours).view(np.uint8), ref.view(np.uint8)), (typ, d)


def test_fp16_narrowing_is_the_reference_rounding_not_numpy(vso):
    # the reference's FP32_to_FP16 is truncate-12-bits-then-round, which differs from IEEE RNE on
    # some inputs; the host code must follow the reference
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, 20000).astype(np.float32)
    ref = vso.f32_to_f16(x)
    assert (ref != x.astype(np.float16).view(np.uint16)).sum() >= 0  # informational
    # normalise a 1-element fp16 vector: result is +-1.0 exactly
    for v in (0.37, -2.5):
        h = vso.f32_to_f16(np.array([v], dtype=np.float32))
        out = VecSim.normalize(h, 3)
        assert out[0] == (0x3C00 if v > 0 else 0xBC00)


def test_query_blob_size():
    L = _capi.load()
    assert L.VecSimParams_GetQueryBlobSize(0, 128, 0) == 512
    assert L.VecSimParams_GetQueryBlobSize(1, 128, 2) == 1024
    assert L.VecSimParams_GetQueryBlobSize(2, 128, 1) == 256
    assert L.VecSimParams_GetQueryBlobSize(4, 128, 2) == 132  # norm appended (vec_sim.cpp:256-266)
    assert L.VecSimParams_GetQueryBlobSize(5, 128, 0) == 128


def test_null_result_accessors():
    L = _capi.load()
    assert L.VecSimQueryResult_GetId(None) == 0xFFFFFFFF  # INVALID_ID widened (query_results.cpp:54-59)
    assert np.isnan(L.VecSimQueryResult_GetScore(None))


def test_sq8_host_preprocessor_matches_the_oracle_and_the_reference_kats(vso):
    """The product's QuantPreprocessor restatement (csrc/host/sq8_prep.h, through the C ABI: VecSimGpu_SQ8_Quantize /
    _QueryBlob) against oracle/vso_sq8.c on random vectors, and against the reference's known answers directly."""
    import json
    import os
    from vectorsimilarity_amd import VecSim
    rng = np.random.default_rng(3)
    for dim in [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 33, 64, 100, 128, 777, 1024]:
        for metric in (0, 1, 2):
            for scale in (1.0, 1e-3, 50.0):
                x = (rng.uniform(-1, 1, dim) * scale).astype(np.float32)
                assert np.array_equal(VecSim.sq8_quantize(x, metric), vso.sq8_quantize(x, metric)), (dim, metric)
                assert np.array_equal(VecSim.sq8_query_blob(x, metric).view(np.uint32),
                                      vso.sq8_query_blob(x, metric).view(np.uint32)), (dim, metric)
                if metric != 2:   # mean-centred blobs (WithNorm = true; L2 and IP)
                    mean = (rng.uniform(-1, 1, dim) * scale).astype(np.float32)
                    assert np.array_equal(VecSim.sq8_quantize_centred(x, mean, metric), vso.sq8_quantize_norm(x, mean, metric)), (dim, metric)
                    assert np.array_equal(VecSim.sq8_query_blob_centred(x, mean, metric).view(np.uint8),
                                          vso.sq8_query_blob_norm(x, mean, metric)), (dim, metric)
    with open(os.path.join(os.path.dirname(__file__), "golden", "kat_sq8.json")) as f:
        kats = json.load(f)["quantize"]
    mcode = {"L2": 0, "IP": 1, "Cosine": 2}
    for c in kats:
        x = np.array(c["input"], dtype=np.float32)
        blob = VecSim.sq8_quantize(x, mcode[c["metric"]])
        assert blob[:x.size].tolist() == c["bytes"], c["name"]
        if c.get("exact_meta"):
            meta = blob[x.size:].view(np.float32)
            assert meta[0] == np.float32(c["min"]) and meta[1] == np.float32(c["delta"]) and meta[2] == np.float32(c["sum"]), c["name"]


def test_sq8_mfma_filter_bound_holds(vso):
    """The SQ8 MFMA filter (csrc/mfma_lowp_kernels.hpp epilogue_sq8, host half in csrc/vsgpu_lowp.hip) replaces the fp32 query
    by one int8 piece per element and must bracket the reference's score: |score_ref - score| <= E.  Restated here in numpy
    (Cauchy-Schwarz on the centred codes: |sum (c_i - 128) e_i| <= |c - 128|_2 |e|_2) with the kernel's fp32 operation order and checked against the oracle on random and hostile inputs (wide row scales,
    a dominant query component, near-constant rows)."""
    rng = np.random.default_rng(8)
    u = np.float32(2.0 ** -24)
    kU = np.float32(64.0) * u
    worst = 0.0
    for trial in range(400):
        dim = int(rng.choice([8, 33, 100, 128, 320, 768, 1024]))
        metric = int(rng.integers(0, 3))
        x = (rng.uniform(-1, 1, dim) * np.exp(rng.uniform(-4, 6))).astype(np.float32)
        if trial % 7 == 0:
            x = (np.float32(3.0) + rng.uniform(-1e-4, 1e-4, dim)).astype(np.float32)
        y = rng.uniform(-1, 1, dim).astype(np.float32)
        if trial % 3 == 0:
            y[rng.integers(0, dim)] *= np.float32(500.0)
        if metric == 2:
            x /= np.float32(np.linalg.norm(x))
            y /= np.float32(np.linalg.norm(y))
        st = vso.sq8_quantize(x, metric)
        qb = vso.sq8_query_blob(y, metric)
        ref = vso.sq8_fp32_distance(metric, st, qb, dim)
        # host half: s, Y, W
        ymax = float(np.max(np.abs(y.astype(np.float64))))
        sf = np.float32(ymax / 127.0)
        if not (sf > 0) or not np.isfinite(sf):
            sf = np.float32(1.0)
        Y = np.clip(np.rint(y.astype(np.float64) / float(sf)), -127, 127)
        e = y.astype(np.float64) - float(sf) * Y
        yabs = float(np.sum(np.abs(y.astype(np.float64))))
        se = float(np.sum(e))
        ce = np.float32(128.0 * se)
        Wref = np.nextafter(np.float32((2.0 * (dim / 32.0 + 8.0) * 2.0 ** -24 * 255.0 * yabs + 2.0 ** -22 * abs(128.0 * se)) * (1 + 1e-6)), np.float32(np.inf))
        ne = np.nextafter(np.float32(np.sqrt(np.sum(e * e)) * (1 + 1e-6)), np.float32(np.inf))
        K = int(128 * np.sum(Y))
        c = st[:dim].astype(np.int64)
        D = int(np.sum((c - 128) * Y.astype(np.int64)))
        meta = st[dim:].view(np.float32)
        mn, dl = meta[0], meta[1]
        ysum = qb[dim]
        nc = np.nextafter(np.float32(np.sqrt(float(np.sum((c - 128) ** 2))) * 1.000001), np.float32(np.inf))
        f = np.float32(D + K)
        dq = np.float32(dl * np.float32(np.float32(sf * f) + ce))
        my = np.float32(mn * ysum)
        ip = np.float32(my + dq)
        if metric == 0:
            C = np.float32(meta[3] + qb[dim + 1])
            sc = np.float32(C - np.float32(np.float32(2.0) * ip))
        else:
            C = np.float32(1.0)
            sc = np.float32(np.float32(1.0) - ip)
        g = np.float32(2.0 if metric == 0 else 1.0)
        E = np.float32(np.float32(np.float32(g * dl) * np.float32(np.float32(nc * ne) + Wref)) + np.float32(kU * np.float32(np.float32(2.0) * np.float32(abs(my) + abs(dq)) + C)))
        assert float(sc) - float(E) <= ref <= float(sc) + float(E), (trial, dim, metric, ref, float(sc), float(E))
        worst = max(worst, abs(ref - float(sc)) / max(float(E), 1e-30))
    assert worst <= 1.0


@pytest.mark.parametrize("flags,override,expect", [
    ("avx512f,avx512bw,avx512vl,avx512vbmi2,avx512vnni,avx512_bf16", None, "AVX512_BF16"),
    ("avx512f,avx512bw,avx512vbmi2,avx512_bf16", None, "AVX512"),          # avx512_bf16 without avx512vl: IP_space.cpp:585
    ("avx512f", None, "AVX512"),
    ("avx,fma3,f16c", None, "AVX512"),                                     # no AVX-512: still the AVX-512 order (host_tier.h says why)
    ("", None, "AVX512"),
    ("avx,fma3,f16c", "scalar", "SCALAR"),                                 # the scalar order is an explicit choice
    ("avx512f,avx512vl,avx512_bf16", "avx512", "AVX512"),                  # the override wins
    ("avx,f16c", "avx512_bf16", "AVX512_BF16"),
    ("avx512f,avx512vl,avx512_bf16,avx512_fp16", None, "AVX512_BF16"),     # IP_space.cpp:649-658's half-accumulating fp16 kernels are OPT-IN
    ("avx512f,avx512_fp16", None, "AVX512"),                               # (unpinned order, gcc >= 12 builds only: host_tier.h)
    ("avx512f,avx512vl,avx512_bf16,avx512_fp16", "avx512_bf16", "AVX512_BF16"),   # a gcc-11 build of the reference has no such kernels
    ("avx512f", "avx512_fp16", "AVX512_FP16"),
])
def test_host_tier_rule(monkeypatch, flags, override, expect):
    """host_tier.h: the tier comes from the host's CPU features the way the reference's choosers pick their kernels
    (spaces.h:68-78, IP_space.cpp:554-615), $VECSIM_GPU_TIER overrides; no GPU involved"""
    from vectorsimilarity_amd import _capi
    lib = _capi.load()
    monkeypatch.setenv("VECSIM_GPU_HOST_FLAGS", flags)
    if override is None:
        monkeypatch.delenv("VECSIM_GPU_TIER", raising=False)
    else:
        monkeypatch.setenv("VECSIM_GPU_TIER", override)
    assert lib.VecSimGpu_HostTier().decode() == expect


@pytest.mark.parametrize("flags,typ,expect", [
    ("avx512f,avx512bw,avx512vl,avx512vbmi2", 2, ""),                      # bf16: L2_space.cpp:332-337 satisfied
    ("avx512f", 2, "avx512bw,avx512vbmi2"),                                # avx512f alone: the reference's bf16 chooser falls lower
    ("avx512f,avx512bw", 2, "avx512vbmi2"),
    ("avx512f,avx512vl", 3, "avx512bw"),                                   # fp16 needs avx512bw && avx512vl
    ("avx512f", 0, ""), ("avx512f", 4, ""), ("avx512f", 5, ""),            # fp32; integers: every tier gives the same number
    ("avx,fma3", 0, "avx512f"),
])
def test_host_tier_note_names_missing_features(monkeypatch, flags, typ, expect):
    """round-4 advisor finding: a host with avx512f but without avx512bw / avx512vbmi2 reports AVX512 for bf16 / fp16 tables although
    its own reference build would run lower-tier kernels: the library says which feature is missing (stderr at index creation,
    VecSimGpu_HostTierNote to callers)"""
    from vectorsimilarity_amd import _capi
    monkeypatch.setenv("VECSIM_GPU_HOST_FLAGS", flags)
    assert _capi.load().VecSimGpu_HostTierNote(typ).decode() == expect


def test_host_tier_probe_matches_proc_cpuinfo(monkeypatch):
    """without the test hook the probe is the CPU's own feature list"""
    from vectorsimilarity_amd import _capi
    lib = _capi.load()
    monkeypatch.delenv("VECSIM_GPU_HOST_FLAGS", raising=False)
    monkeypatch.delenv("VECSIM_GPU_TIER", raising=False)
    flags = set()
    with open("/proc/cpuinfo") as f:
        for line in f:
            if line.startswith("flags"):
                flags = set(line.split(":", 1)[1].split())
                break
    expect = "AVX512_BF16" if {"avx512f", "avx512_bf16", "avx512vl"} <= flags else "AVX512"
    assert lib.VecSimGpu_HostTier().decode() == expect


HL_DIMS = {"f32": list(range(1, 100)) + [128, 500, 768, 1024], "f64": list(range(1, 50)) + [128, 768], "f16": list(range(1, 100)) + [128, 768],
           "bf16": list(range(1, 130)) + [768, 1024], "i8": list(range(1, 140)) + [768, 1024], "u8": list(range(1, 140)) + [1024]}


@pytest.mark.parametrize("typ", list(HL_DIMS))
def test_host_lane_walker_equals_the_oracle_for_every_residual_class(vso, typ):
    """csrc/host/host_lane_eval.h -- what the reference-order HNSW insert path ranks its candidates by -- against the pinned kernel
    oracle, bit for bit: every dim's residual class, L2 / IP / Cosine, the AVX-512 tier (+ the vdpbf16ps tier for bf16), stored blobs
    (Cosine: normalised, int8 / uint8 with the norm behind the elements); no GPU involved"""
    import ctypes as C
    from util import METRICS, TYPES, random_vectors, stored_rows
    lib = _capi.load()
    rng = np.random.default_rng(11)
    t = TYPES[typ]
    for d in HL_DIMS[typ]:
        v = random_vectors(rng, 2, d, typ, vso)
        for metric in ("L2", "IP", "Cosine"):
            st = stored_rows(vso, v, typ, metric)
            a, b = np.ascontiguousarray(st[0]), np.ascontiguousarray(st[1])
            km = METRICS["IP"] if (metric == "Cosine" and typ not in ("i8", "u8")) else METRICS[metric]
            for tier in [0] + ([2] if typ == "bf16" else []):
                got = lib.VecSimGpu_HostLaneDistance(t, METRICS[metric], tier, d, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
                want = vso.distance(t, km, a, b, dim=d, tier=tier)
                assert got == want or (got != got and want != want), (typ, metric, d, tier, got, want)
    assert np.isnan(lib.VecSimGpu_HostLaneDistance(TYPES["f16"], 0, 3, 64, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))) or typ != "f16"
