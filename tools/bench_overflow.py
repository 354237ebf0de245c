#!/usr/bin/env python3
"""The candidate-list overflow cliff: fp32 L2, N x 768, batch 64, top-10 on a table with a cluster of
near-duplicate rows -- a query next to the cluster collects more candidates than its list holds (cand_cap), and used to be
answered by a dense exact pass over all rows.  Prints the batch time with 0 / 1 / 8 such queries in the batch and the number of
fallbacks taken.      python tools/bench_overflow.py [--rows 10000000] [--cluster 30000] [--noise 0.02]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--cluster", type=int, default=30_000)
ap.add_argument("--noise", type=float, default=0.02)
ap.add_argument("--check", type=int, default=1)
a = ap.parse_args()
p = VecSim.BFParams()
p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, a.dim, VecSim.VecSimMetric_L2
ix = VecSim.BFIndex(p)
ix.add_synthetic(a.rows - a.cluster, 47)
rng = np.random.default_rng(3)
centre = rng.uniform(-1, 1, a.dim).astype(np.float32)
cl = (centre[None, :] + a.noise * rng.standard_normal((a.cluster, a.dim))).astype(np.float32)
for i in range(0, a.cluster, 10_000):
    ix.add_vectors(cl[i:i + 10_000], np.arange(a.rows - a.cluster + i, a.rows - a.cluster + min(a.cluster, i + 10_000)))
base_q = synth.rows_f32(48, 0, 64, a.dim)
for hot in (0, 1, 8):
    q = base_q.copy()
    for j in range(hot):
        q[j] = centre + a.noise * rng.standard_normal(a.dim).astype(np.float32)
    ix.knn_query(q, 10)
    ix.reset_stats()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        l, d = ix.knn_query(q, 10)
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None else min(best, dt)
    st = ix.stats()
    ok = ""
    if a.check and hot:
        ix.set_option("mfma", 0)
        l0, d0 = ix.knn_query(q[:hot], 10)
        ix.set_option("mfma", 1)
        ok = "  replies == exact path: %s" % bool(np.array_equal(l[:hot], l0) and np.array_equal(d[:hot], d0))
    print("%d of 64 queries next to a %d-row cluster: batch %.2f ms (best of 3), fallbacks per batch %.1f, retries per batch %.1f, "
          "candidates/query %.0f%s" % (hot, a.cluster, best, st["fallbacks"] / 3.0, st.get("retries", 0) / 3.0,
                                       st["candidates"] / (3 * 64.0), ok), flush=True)
