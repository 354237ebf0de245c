#!/usr/bin/env python3
"""fp64 Flat index, batch of 64 queries, top-10: MFMA filter + double re-rank against the dense exact fp64 scan."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=2_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--k", type=int, default=10)
a = ap.parse_args()
p = VecSim.BFParams()
p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT64, a.dim, VecSim.VecSimMetric_L2
ix = VecSim.BFIndex(p)
rng = np.random.default_rng(1)
chunk = 100_000
for r0 in range(0, a.rows, chunk):
    m = min(chunk, a.rows - r0)
    ix.add_vectors(rng.uniform(-1, 1, (m, a.dim)), np.arange(r0, r0 + m))
q = rng.uniform(-1, 1, (a.batch, a.dim))
res = {}
for mfma in (1, 0):
    ix.set_option("mfma", mfma)
    res[mfma] = ix.knn_query(q, a.k)
    ix.reset_stats()
    ts = []
    for _ in range(5 if mfma else 2):
        t0 = time.perf_counter()
        ix.knn_query(q, a.k)
        ts.append((time.perf_counter() - t0) * 1e3)
    st = ix.stats()
    gb = a.rows * a.dim * 8 / 1e9
    print("mfma %d: batch min %.2f ms mean %.2f ms; scan kernel %s %.3f ms/launch = %.0f GB/s of fp64 rows; candidates/query %.0f"
          % (mfma, min(ts), sum(ts) / len(ts), st["scan_kernel"], st["scan_ms"] / max(1, st["scan_launches"]),
             gb / (st["scan_ms"] / max(1, st["scan_launches"]) / 1e3), st["candidates"] / max(1, a.batch * st["scan_launches"])), flush=True)
assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
print("identical replies: True")
