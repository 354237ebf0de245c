"""CPU: the lane tables that drive the exact HIP kernels (vectorsimilarity_amd/csrc/lane_program.h)
reproduce the oracle bit for bit when walked the way the kernel walks them (tests/helpers/lane_emul.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from util import random_vectors

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELP = os.path.join(ROOT, "tests", "helpers")


@pytest.fixture(scope="module")
def emul():
    so = os.path.join(HELP, "liblane_emul.so")
    src = os.path.join(HELP, "lane_emul.cpp")
    hdr = os.path.join(ROOT, "vectorsimilarity_amd", "csrc", "lane_program.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma",
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "vectorsimilarity_amd", "csrc"),
                        "-o", so, src], check=True)
    L = C.CDLL(so)
    L.lane_emul.restype = C.c_double
    L.lane_emul.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
    L.lane_emul_sq8.restype = C.c_double
    L.lane_emul_sq8.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
    L.lane_emul_sq8h.restype = C.c_double
    L.lane_emul_sq8h.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
    return L


DIMS = {"f32": list(range(1, 100)) + [128, 500, 768, 1024],
        "f64": list(range(1, 50)) + [128, 768],
        "f16": list(range(1, 100)) + [128, 768],
        "bf16": list(range(1, 130)) + [768, 1024],
        "i8": list(range(1, 140)) + [768, 1024],
        "u8": list(range(1, 140)) + [1024]}
TCODE = {"f32": 0, "f64": 1, "bf16": 2, "f16": 3, "i8": 4, "u8": 5}


@pytest.mark.parametrize("typ", list(DIMS))
def test_lane_tables_reproduce_oracle(emul, vso, typ):
    rng = np.random.default_rng(5)
    t = TCODE[typ]
    tiers = [0, 1] + ([2] if typ == "bf16" else [])
    for d in DIMS[typ]:
        v = random_vectors(rng, 2, d, typ, vso)
        a, b = np.ascontiguousarray(v[0]), np.ascontiguousarray(v[1])
        for metric in (0, 1):
            for tier in tiers:
                acc = emul.lane_emul(t, metric, tier, d, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
                want = vso.distance(t, metric, a, b, dim=d, tier=tier)
                if typ in ("i8", "u8"):
                    got = float(np.float32(acc)) if metric == 0 else float(np.float32(1 - acc))
                elif typ == "f64":
                    got = acc if metric == 0 else 1.0 - acc
                else:
                    got = acc if metric == 0 else float(np.float32(1.0) - np.float32(acc))
                assert got == want, (typ, d, metric, tier, got, want)


def test_sq8_lane_tables_reproduce_oracle(emul, vso):
    """SQ8 storage x FP32 query: the table + epilogue the GPU kernel runs (k_exact_scan<EK_SQ8>, sq8_score) against
    oracle/vso_sq8.c, scalar and AVX-512 tier, every residual class of dim, all three metrics."""
    rng = np.random.default_rng(11)
    for d in list(range(1, 140)) + [768, 1000, 1024]:
        for metric in (0, 1, 2):
            x = rng.uniform(-1, 1, d).astype(np.float32)
            y = rng.uniform(-1, 1, d).astype(np.float32)
            if metric == 2:
                x /= max(np.linalg.norm(x), 1e-6)
                y /= max(np.linalg.norm(y), 1e-6)
            st = vso.sq8_quantize(x, metric)
            qb = vso.sq8_query_blob(y, metric)
            for tier in (0, 1):
                got = emul.lane_emul_sq8(0 if metric == 0 else 1, tier, d, st.ctypes.data_as(C.c_void_p), qb.ctypes.data_as(C.c_void_p))
                want = vso.sq8_fp32_distance(metric, st, qb, d, tier=tier)
                assert got == want, (d, metric, tier, got, want)


def test_sq8_fp16_query_lane_tables_reproduce_oracle(emul, vso):
    """SQ8 storage x FP16 query (four 16-lane accumulators on 64 virtual lanes; scalar below dim 16)"""
    rng = np.random.default_rng(12)
    for d in list(range(1, 200)) + [768, 1000, 1024]:
        for metric in (0, 1):
            x = rng.uniform(-1, 1, d).astype(np.float32)
            y = rng.uniform(-1, 1, d).astype(np.float16).view(np.uint16)
            st = vso.sq8_quantize(x, metric)
            qb = vso.sq8_query_blob_f16(y, metric)
            for tier in (0, 1):
                got = emul.lane_emul_sq8h(metric, tier, d, st.ctypes.data_as(C.c_void_p), qb.ctypes.data_as(C.c_void_p))
                want = vso.sq8_fp16_distance(metric, st, qb, d, tier=tier)
                assert got == want, (d, metric, tier, got, want)
