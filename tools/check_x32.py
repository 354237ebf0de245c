#!/usr/bin/env python3
"""32x32x32 int8 filter (option lowp_x32 = VAR + 1) against the 16x16x64 filter and the exact path: python tools/check_x32.py [--vars 1,2,4]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import METRICS, TYPES, random_vectors  # noqa: E402
from vectorsimilarity_amd import VecSim  # noqa: E402
from oracle import vso  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--vars", default="2")
a = ap.parse_args()
vso.build()
vso.lib()
CASES = [("i8", "Cosine", 1024, 40_000, 256, 100), ("i8", "L2", 1024, 30_011, 300, 10), ("i8", "IP", 900, 25_013, 200, 10),
         ("u8", "L2", 1024, 30_000, 256, 10), ("u8", "IP", 800, 20_005, 130, 100), ("i8", "Cosine", 1024, 150_000, 256, 20)]
bad = 0
for typ, metric, dim, n, nq, k in CASES:
    rng = np.random.default_rng(dim * 7 + n + nq)
    if n == 150_000:   # repeated rows: exact ties, every workgroup walks several tiles
        base = random_vectors(rng, 5_000, dim, typ, vso)
        rows = base[rng.integers(0, 5_000, n)]
    else:
        rows = random_vectors(rng, n, dim, typ, vso)
    q = random_vectors(rng, nq, dim, typ, vso)
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = TYPES[typ], dim, METRICS[metric]
    ix = VecSim.BFIndex(p)
    ix.add_vectors(rows, np.arange(n))
    ix.set_option("dense_pairs", 0)
    ix.set_option("lowp_x32", 0)
    l0, d0 = ix.knn_query(q, k)
    for v in [int(x) for x in a.vars.split(",")]:
        ix.set_option("lowp_x32", v)
        ix.reset_stats()
        l1, d1 = ix.knn_query(q, k)
        st = ix.stats()
        ok = bool(np.array_equal(l0, l1) and np.array_equal(d0, d1))
        bad += 0 if ok else 1
        nbad = int((l0 != l1).any(axis=1).sum())
        print("%s %s d=%d n=%d nq=%d k=%d  lowp_x32=%d  kernel %s  same as 16x16x64: %s (%d queries differ)" % (
            typ, metric, dim, n, nq, k, v, st["scan_kernel"], ok, nbad), flush=True)
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
