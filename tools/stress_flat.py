#!/usr/bin/env python3
"""Hunt for run-to-run differences on the MFMA filter path.  Two data sets of the same shape alternate: every
repetition frees the previous index and builds a fresh one over the OTHER data set (the allocator hands back the
same device addresses), then compares the first query pass with that data set's exact-path answer."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
kind = sys.argv[2] if len(sys.argv) > 2 else "f32"      # f32 | f32odd | f64 | f64odd | bf16 | i8
opts = dict(kv.split("=") for kv in sys.argv[3:])
dim, n, nq, k = {"f32": (128, 150_000, 64, 10), "f32odd": (100, 150_000, 64, 10), "bf16": (256, 120_000, 100, 10),
                 "i8": (512, 100_000, 200, 10), "f64": (256, 80_000, 64, 10), "f64odd": (100, 100_000, 40, 10)}[kind]
VT = {"f32": VecSim.VecSimType_FLOAT32, "f32odd": VecSim.VecSimType_FLOAT32, "bf16": VecSim.VecSimType_BFLOAT16,
      "i8": VecSim.VecSimType_INT8, "f64": VecSim.VecSimType_FLOAT64, "f64odd": VecSim.VecSimType_FLOAT64}[kind]
rng = np.random.default_rng(1)
sets = []
for s in range(2):
    if kind == "i8":
        rows = rng.integers(-128, 128, (n, dim)).astype(np.int8)
        q = rng.integers(-128, 128, (nq, dim)).astype(np.int8)
    else:
        ft = np.float64 if kind.startswith("f64") else np.float32
        rows = rng.uniform(-1, 1, (n, dim)).astype(ft)
        q = rng.uniform(-1, 1, (nq, dim)).astype(ft)
        if kind == "bf16":
            rows = (rows.view(np.uint32) >> 16).astype(np.uint16)
            q = (q.view(np.uint32) >> 16).astype(np.uint16)
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = VT, dim, VecSim.VecSimMetric_L2
    ix = VecSim.BFIndex(p)
    ix.add_vectors(rows, np.arange(n))
    ix.set_option("mfma", 0)
    sets.append((rows, q, ix.knn_query(q, k)))
    del ix
bad = 0
for r in range(reps):
    rows, q, ref = sets[r % 2]
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = VT, dim, VecSim.VecSimMetric_L2
    ix = VecSim.BFIndex(p)
    ix.add_vectors(rows, np.arange(n))
    ix.set_option("dense_pairs", 0)
    for key, v in opts.items():
        ix.set_option(key, int(v))
    l, d = ix.knn_query(q, k)
    if not (np.array_equal(l, ref[0]) and np.array_equal(d, ref[1])):
        bad += 1
        w = np.argwhere(l != ref[0])
        qb = int(w[0][0])
        print("MISMATCH rep", r, "query", qb, "missing", [int(x) for x in ref[0][qb] if x not in set(l[qb].tolist())], flush=True)
    del ix
print("kind", kind, "reps", reps, "bad", bad, "opts", opts)
