#!/usr/bin/env python3
"""Differential soak of the reference-order HNSW insert path: random (type, metric, dim, n, M, efConstruction), vectors with planted
duplicates and coarse grids (exact distance ties), VecSimIndex_AddVector x n through the C API against the oracle's insert path
(oracle/vso_hnsw.c, pinned on the reference-built graphs) -- levels, entry point and every link list in order.
    python tools/fuzz_hnsw_build.py --seconds 240 [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("VECSIM_GPU_TIER", "avx512")
from oracle import vso  # noqa: E402
from util import METRICS, TYPES, random_vectors, stored_rows  # noqa: E402
from vectorsimilarity_amd import VecSim  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
vso.build()
t0, shapes, bad = time.time(), 0, 0
while time.time() - t0 < a.seconds:
    typ = str(rng.choice(["f32", "f32", "bf16", "f16", "f64", "i8", "u8"]))
    metric = str(rng.choice(["L2", "IP", "Cosine"]))
    dim = int(rng.choice([3, 4, 7, 8, 9, 15, 16, 17, 31, 32, 33, 48, 64, 100, 128, 200]))
    n = int(rng.integers(50, 2500))
    M = int(rng.choice([2, 3, 4, 8, 16, 24, 32]))
    efc = int(rng.choice([1, 5, 10, 40, 100, 200]))
    rows = random_vectors(rng, n, dim, typ, vso)
    mode = int(rng.integers(0, 3))
    if mode == 1 and typ in ("f32", "f64"):          # coarse grid: many exactly equal distances
        rows = np.round(rows * 2).astype(rows.dtype) / 2
    if mode == 2:                                    # duplicates of earlier rows
        src = rng.integers(0, n, n // 5)
        rows[rng.integers(0, n, n // 5)] = rows[src]
    if metric == "Cosine":
        # no zero vectors (-0.0 included): their normalisation is 0 / 0, every distance to them NaN, and which of two pairs with NaN
        # scores std::priority_queue moves is a property of the language standard the reference is built with (docs/HISTORY.md 3)
        if typ in ("bf16", "f16"):
            zero = np.all((rows & 0x7FFF) == 0, axis=1)
        else:
            zero = np.all(rows == 0, axis=1)
        for i in np.nonzero(zero)[0]:
            rows[i] = random_vectors(rng, 1, dim, typ, vso)[0]
            if typ in ("f32", "f64"):
                rows[i] += 3.0
    p = VecSim.HNSWParams()
    p.type, p.dim, p.metric, p.M, p.efConstruction, p.efRuntime = TYPES[typ], dim, METRICS[metric], M, efc, 10
    ix = VecSim.HNSWIndex(p)
    for i in range(n):
        ix.add_vector(rows[i], i)
    got = ix.graph()
    st = stored_rows(vso, rows, typ, metric)
    km = METRICS["IP"] if (metric == "Cosine" and typ not in ("i8", "u8")) else METRICS[metric]
    ref = vso.hnsw_build(TYPES[typ], km, st, dim, M, efc)
    ok = got["reference_order_build"] and got["entry"] == ref["entry"] and got["max_level"] == ref["max_level"] and \
        np.array_equal(got["levels"], ref["levels"]) and vso.graph_lists(got) == vso.graph_lists(ref)
    shapes += 1
    if not ok:
        bad += 1
        print("MISMATCH", typ, metric, dim, n, M, efc, mode, flush=True)
print("fuzz_hnsw_build: %d shapes, %d mismatches, %.0f s" % (shapes, bad, time.time() - t0))
sys.exit(1 if bad else 0)
