#!/usr/bin/env python3
"""Differential soak of the HNSW batch iterator: random indexes (type, metric, dim, M, ef, single / multi-value, deleted labels,
tied distances from coarse integer coordinates) and random batch-size sequences; the product's walk (host heaps, GPU distances)
against oracle/vso_hnsw.c's twin on the graph the index exports.      python tools/fuzz_hnsw_iter.py [--seconds 120] [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from vectorsimilarity_amd import VecSim  # noqa: E402
from oracle import vso  # noqa: E402  (the checker)

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t_end = time.time() + a.seconds
runs = mism = batches = 0
while time.time() < t_end:
    dim = int(rng.integers(2, 160))
    n = int(rng.integers(50, 4000))
    M = int(rng.integers(2, 17))
    ef = int(rng.integers(1, 64))
    metric = int(rng.integers(0, 3))
    multi = bool(rng.integers(0, 2))
    coarse = bool(rng.integers(0, 3) == 0)   # integer coordinates in a small range: many tied distances, duplicate vectors
    if coarse:
        rows = rng.integers(-2, 3, (n, dim)).astype(np.float32)
        rows[np.all(rows == 0, axis=1)] = 1.0
    else:
        rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    labels = rng.integers(0, max(2, n // int(rng.integers(1, 6))), n) if multi else rng.permutation(n)
    p = VecSim.HNSWParams()
    p.type, p.dim, p.metric, p.M, p.efConstruction, p.efRuntime, p.multi = VecSim.VecSimType_FLOAT32, dim, metric, M, int(rng.integers(8, 80)), ef, multi
    ix = VecSim.HNSWIndex(p)
    ix.add_vectors(rows, labels)
    for lab in rng.choice(np.unique(labels), size=int(rng.integers(0, 6)), replace=False):
        ix.delete_vector(int(lab))
    g = ix.graph()
    # the rows in the graph's own node order, as stored (Cosine: normalised at ingest) -- a batch of deletes may have removed dead
    # nodes and moved the last live ones into the holes (round 5), so insertion order is not node order
    srows = ix.stored_rows(0, g["n"]).view(np.float32).reshape(g["n"], dim).copy() if g["n"] else rows[:0]
    km = 0 if metric == VecSim.VecSimMetric_L2 else 1
    for _ in range(3):
        q = (rng.integers(-2, 3, dim).astype(np.float32) if coarse else rng.uniform(-1, 1, dim).astype(np.float32))
        if not np.any(q):
            q[0] = 1.0
        sq = q.copy()
        if metric == VecSim.VecSimMetric_Cosine:
            vso.normalize(sq, dim, 0)
        sizes = [int(x) for x in rng.integers(0, 2 * ef + 8, int(rng.integers(1, 30)))]
        want, want_dep = vso.hnsw_iterate(0, km, srows, g, sq, ef, sizes, dim, multi=multi)
        it = ix.create_batch_iterator(q)
        got = []
        for m in sizes:
            if not it.has_next():
                break
            l, d = it.get_next_results(m, VecSim.BY_SCORE)
            got.append((l[0], d[0]))
        ok = len(got) == len(want) and (not it.has_next()) == want_dep
        for (gl, gd), (wl, wd) in zip(got, want):
            ok = ok and np.array_equal(gl[:len(wl)], wl.astype(np.int64)) and np.all(gl[len(wl):] == -1) and np.array_equal(gd[:len(wd)], wd)
        runs += 1
        batches += len(want)
        if not ok:
            mism += 1
            print("MISMATCH seed %d run %d: dim %d n %d M %d ef %d metric %d multi %s coarse %s sizes %s" % (a.seed, runs, dim, n, M, ef, metric, multi, coarse, sizes), flush=True)
print("iterations %d batches %d mismatches %d" % (runs, batches, mism))
sys.exit(1 if mism else 0)
