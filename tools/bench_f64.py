#!/usr/bin/env python3
"""fp64 Flat L2 top-10 latency (single query and batch 64) on a ~3 GB table."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim  # noqa: E402

dim, n = 256, 1_500_000
rng = np.random.default_rng(0)
rows = rng.uniform(-1, 1, (n, dim))
p = VecSim.BFParams()
p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT64, dim, VecSim.VecSimMetric_L2
ix = VecSim.BFIndex(p)
ix.add_vectors(rows, np.arange(n))
for nq in (1, 8, 64):
    q = rng.uniform(-1, 1, (nq, dim))
    ix.knn_query(q, 10)
    ix.reset_stats()
    t0 = time.perf_counter()
    for _ in range(3):
        ix.knn_query(q, 10)
    dt = (time.perf_counter() - t0) / 3
    print("fp64 %d x %d, %d queries: %.2f ms per batch (%s)" % (n, dim, nq, dt * 1e3, ix.stats()["scan_kernel"]), flush=True)
