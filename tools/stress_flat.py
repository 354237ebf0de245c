#!/usr/bin/env python3
"""Repeat the MFMA-filter parity cases many times in one process, looking for run-to-run differences."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
CASES = [(256, 100_003, 40, 100), (128, 150_000, 64, 10), (768, 60_000, 64, 10), (384, 50_000, 130, 10)]
bad = 0
for dim, n, nq, k in CASES:
    rng = np.random.default_rng(dim + n)
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    q = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    ref = None
    for r in range(reps):
        p = VecSim.BFParams()
        p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, dim, VecSim.VecSimMetric_L2
        ix = VecSim.BFIndex(p)
        ix.add_vectors(rows, np.arange(n))
        ix.set_option("dense_pairs", 0)
        ix.reset_stats()
        l, d = ix.knn_query(q, k)
        st = ix.stats()
        if ref is None:
            ix.set_option("mfma", 0)
            ref = ix.knn_query(q, k)
        if not (np.array_equal(l, ref[0]) and np.array_equal(d, ref[1])) or st["fallbacks"] or st["scan_kernel"] != "k_mfma_filter":
            bad += 1
            w = np.argwhere(l != ref[0])
            print("MISMATCH dim", dim, "rep", r, "where", w[:6].tolist(), "stats", st, flush=True)
        del ix
    print("case", (dim, n, nq, k), "done", flush=True)
print("bad =", bad)
