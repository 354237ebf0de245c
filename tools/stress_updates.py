#!/usr/bin/env python3
"""Deletes (swap-delete = device row move + norm update) and overwrites followed at once by an MFMA-path query,
checked against the exact path of the same index: hunts for ordering bugs between row writes and norm kernels."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(3)
dim, n = 128, 40_000
bad = 0
for r in range(reps):
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32) * np.float32(1 + (r % 5))
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, dim, VecSim.VecSimMetric_L2
    ix = VecSim.BFIndex(p)
    ix.add_vectors(rows, np.arange(n))
    q = rng.uniform(-1, 1, (32, dim)).astype(np.float32)
    for lab in rng.choice(n, 200, replace=False):
        ix.delete_vector(int(lab))
    for lab in rng.choice(n, 50, replace=False):
        ix.add_vector(rng.uniform(-3, 3, dim).astype(np.float32), int(lab))
    ix.set_option("dense_pairs", 0)
    l1, d1 = ix.knn_query(q, 10)
    ix.set_option("mfma", 0)
    l2, d2 = ix.knn_query(q, 10)
    if not (np.array_equal(l1, l2) and np.array_equal(d1, d2)):
        bad += 1
        print("MISMATCH rep", r, np.argwhere(l1 != l2)[:4].tolist(), flush=True)
    del ix
print("reps", reps, "bad", bad)
