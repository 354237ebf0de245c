#!/usr/bin/env python3
"""Flat top-10, batch 64, across dims at a fixed table size in bytes (~6 GB): scan kernel rate.
    python tools/bench_dims.py [--type f32|bf16|f16|i8|u8] [--metric L2|IP|Cosine] [--batch 64] dims..."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--type", default="f32")
ap.add_argument("--metric", default="L2")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--opt", action="append", default=[], help="NAME=VALUE index option (repeatable)")
ap.add_argument("dims", nargs="*", type=int, default=[768, 1536, 2048, 3072])
a = ap.parse_args()
T = {"f32": (VecSim.VecSimType_FLOAT32, 4, synth.rows_f32), "bf16": (VecSim.VecSimType_BFLOAT16, 2, synth.rows_bf16),
     "f16": (VecSim.VecSimType_FLOAT16, 2, synth.rows_f16), "i8": (VecSim.VecSimType_INT8, 1, synth.rows_i8),
     "u8": (VecSim.VecSimType_UINT8, 1, lambda s, f, n, d: synth.rows_i8(s, f, n, d).view("uint8"))}[a.type]
for dim in a.dims:
    n = int(6e9 / (dim * T[1]))
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = T[0], dim, getattr(VecSim, "VecSimMetric_" + a.metric)
    ix = VecSim.BFIndex(p)
    try:
        ix.add_synthetic(n, 47)
    except RuntimeError:   # (no device-side generator for this type: uint8) host rows, uploaded in chunks; a smaller table
        import numpy as np
        n = n // 4
        rng = np.random.default_rng(47)
        for r0 in range(0, n, 65536):
            cnt = min(65536, n - r0)
            ix.add_vectors(rng.integers(0, 256, (cnt, dim), dtype=np.uint8), np.arange(r0, r0 + cnt))
    for o in a.opt:
        ix.set_option(o.split("=")[0], int(o.split("=")[1]))
    q = T[2](48, 0, a.batch, dim)
    ix.knn_query(q, 10)
    ix.reset_stats()
    t0 = time.perf_counter()
    for _ in range(5):
        ix.knn_query(q, 10)
    dt = (time.perf_counter() - t0) / 5
    st = ix.stats()
    kms = st["scan_ms"] / st["scan_launches"]
    print("%s %s dim %5d rows %8d batch %d: %.3f ms per batch, %s %.3f ms = %.0f GB/s, cand/query %.0f" % (
        a.type, a.metric, dim, n, a.batch, dt * 1e3, st["scan_kernel"], kms, st["scan_bytes"] / st["scan_launches"] / kms / 1e6,
        st["candidates"] / (5 * a.batch)), flush=True)
    del ix
