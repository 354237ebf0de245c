#!/usr/bin/env python3
"""Flat fp32 L2 batch-64 top-10 across dims at a fixed table size in bytes (~6 GB): filter kernel rate."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim, synth  # noqa: E402

for dim in [int(x) for x in (sys.argv[1:] or ["768", "1536", "2048", "3072"])]:
    n = int(6e9 / (dim * 4))
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, dim, VecSim.VecSimMetric_L2
    ix = VecSim.BFIndex(p)
    ix.add_synthetic(n, 47)
    q = synth.rows_f32(48, 0, 64, dim)
    ix.knn_query(q, 10)
    ix.reset_stats()
    t0 = time.perf_counter()
    for _ in range(5):
        ix.knn_query(q, 10)
    dt = (time.perf_counter() - t0) / 5
    st = ix.stats()
    kms = st["scan_ms"] / st["scan_launches"]
    print("dim %5d rows %8d: batch %.3f ms, %s %.3f ms = %.0f GB/s, cand/query %.0f" % (
        dim, n, dt * 1e3, st["scan_kernel"], kms, n * dim * 4 / kms / 1e6, st["candidates"] / (5 * 64)), flush=True)
    del ix
