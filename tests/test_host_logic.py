"""CPU: host-side logic of the product that needs no GPU -- blob preprocessing (VecSim_Normalize),
blob sizes, runtime-parameter resolution helpers -- against the oracle."""
import ctypes as C

import numpy as np

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
