#!/usr/bin/env python3
"""Differential soak of the sharded Flat index (G shards driven from one process, all on GPU 0: the partition, the per-shard
candidate rule and the gid-order merge are the distributed index's) against the single index over the same vectors: random type,
metric, dim, rows, shards, block size, batch, k, with duplicate rows (ties across shards) and deletes.
    python tools/fuzz_sharded.py --seconds 120 [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim  # noqa: E402
from vectorsimilarity_amd.sharded import ShardedFlatIndex  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=60)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
T = {"f32": VecSim.VecSimType_FLOAT32, "bf16": VecSim.VecSimType_BFLOAT16, "i8": VecSim.VecSimType_INT8, "u8": VecSim.VecSimType_UINT8}
M = {"L2": VecSim.VecSimMetric_L2, "IP": VecSim.VecSimMetric_IP, "Cosine": VecSim.VecSimMetric_Cosine}


def vectors(typ, n, dim):
    if typ in ("i8", "u8"):
        lo, hi = (-128, 128) if typ == "i8" else (0, 256)
        return rng.integers(lo, hi, (n, dim)).astype(np.int8 if typ == "i8" else np.uint8)
    x = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    if typ == "f32":
        return x
    u = x.view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


t_end = time.time() + a.seconds
runs = bad = 0
while time.time() < t_end:
    typ = rng.choice(list(T))
    metric = rng.choice(list(M))
    dim = int(rng.choice([rng.integers(8, 130), rng.integers(130, 800)]))
    n = int(rng.integers(500, 40_000))
    G = int(rng.choice([2, 3, 4, 8]))
    block = int(rng.choice([64, 256, 1024]))
    nq = int(rng.choice([1, 5, 33, 64, 130]))
    k = int(rng.choice([1, 10, 50]))
    p = VecSim.BFParams()
    p.type, p.dim, p.metric, p.blockSize = T[typ], dim, M[metric], block
    rows = vectors(typ, n, dim)
    if rng.random() < 0.5:   # ties: copies of a few rows scattered over the blocks (and so over the shards)
        src = rng.integers(0, n, 20)
        for s in src:
            rows[rng.integers(0, n, 8)] = rows[s]
    one = VecSim.BFIndex(p)
    sh = ShardedFlatIndex(p, shards=G, devices=[0] * G)
    labels = np.arange(n) * 3 + 1
    for r0 in range(0, n, 10_000):
        one.add_vectors(rows[r0:r0 + 10_000], labels[r0:r0 + 10_000])
        sh.add_vectors(rows[r0:r0 + 10_000], labels[r0:r0 + 10_000])
    if rng.random() < 0.4:
        for lab in rng.choice(labels, 25, replace=False):
            one.delete_vector(int(lab))
            sh.delete_vector(int(lab))
    q = vectors(typ, nq, dim)
    l1, d1 = one.knn_query(q, k)
    l2, d2 = sh.knn_query(q, k)
    runs += 1
    if not (np.array_equal(l1, l2) and np.array_equal(d1, d2, equal_nan=True)):
        bad += 1
        print("MISMATCH", typ, metric, "dim", dim, "n", n, "shards", G, "block", block, "nq", nq, "k", k, flush=True)
    del one, sh
print("runs %d mismatches %d" % (runs, bad))
sys.exit(1 if bad else 0)
