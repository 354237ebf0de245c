#!/usr/bin/env python3
"""Differential soak: random (type, metric, dim, rows, batch, k) on the filter paths against the exact path of the same index
(which the -m gpu suite pins against the oracle, 0 ulp).  Runs until --seconds are used up; prints every mismatch and a summary.
    python tools/fuzz_parity.py --seconds 300 [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from vectorsimilarity_amd import VecSim  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--readers", type=int, default=0, help="> 1: every shape is also queried by this many threads at once (reader lanes)")
ap.add_argument("--stream", action="store_true", help="fp32 rows up to 3072 elements with the opt-in streaming threshold (option stream_tau) at random re-read paces")
ap.add_argument("--verbose", action="store_true", help="print every shape before it runs (the last line names the shape a crash happened in)")
ap.add_argument("--wide", action="store_true", help="only the wide-row kernels' shapes: 16-bit rows of 2049 .. 8192 elements, 8-bit rows of 4097 .. 16384, "
                                                     "fp32 3073 .. 8192, batches above and below the 16 / 32 / 64 queries a workgroup holds")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
T = {"f32": VecSim.VecSimType_FLOAT32, "f64": VecSim.VecSimType_FLOAT64, "bf16": VecSim.VecSimType_BFLOAT16,
     "f16": VecSim.VecSimType_FLOAT16, "i8": VecSim.VecSimType_INT8, "u8": VecSim.VecSimType_UINT8}
M = {"L2": VecSim.VecSimMetric_L2, "IP": VecSim.VecSimMetric_IP, "Cosine": VecSim.VecSimMetric_Cosine}


def vectors(typ, n, dim, scale):
    if typ in ("i8", "u8"):
        lo, hi = (-128, 128) if typ == "i8" else (0, 256)
        return rng.integers(lo, hi, (n, dim)).astype(np.int8 if typ == "i8" else np.uint8)
    x = (rng.uniform(-1, 1, (n, dim)) * scale).astype(np.float32)
    if typ == "f32" or typ == "sq8":
        return x
    if typ == "f64":
        return x.astype(np.float64)
    if typ == "f16":
        return x.astype(np.float16).view(np.uint16)
    u = x.view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)   # bf16, RNE


t_end = time.time() + a.seconds
runs = bad = 0
kernels = {}
while time.time() < t_end:
    typ = rng.choice(["f32", "f32", "bf16", "f16", "i8", "u8", "f64", "sq8", "sq8"])
    metric = rng.choice(["L2", "IP", "Cosine"])
    dmax = {"f32": 8192, "bf16": 8192, "f16": 8192, "i8": 16384, "u8": 16384, "f64": 2048, "sq8": 1024}[typ]   # (round 4: int8 / uint8 wide rows)
    dim = int(rng.choice([rng.integers(8, 200), rng.integers(200, 1100), rng.integers(min(1100, dmax), dmax + 1)], p=[0.3, 0.45, 0.25]))
    dim = min(dim, dmax)
    if a.wide:
        typ = rng.choice(["f32", "bf16", "f16", "i8", "u8"])
        lo, hi = {"f32": (3073, 8192), "bf16": (2049, 8192), "f16": (2049, 8192), "i8": (4097, 16384), "u8": (4097, 16384)}[typ]
        dim = int(rng.choice([rng.integers(lo, lo + (hi - lo) // 6), rng.integers(lo, hi + 1)]))   # (half of them in the narrowest kernel width)
    if a.stream:
        typ = "f32"
        dim = int(rng.choice([rng.integers(33, 600), rng.integers(600, 3073)]))
    eb = {"f32": 4, "f64": 8, "bf16": 2, "f16": 2, "i8": 1, "u8": 1, "sq8": 1}[typ]
    budget = int(rng.choice([3e7, 1.5e8, 4e8]))                      # bytes of rows: one slab .. several
    n = max(300, min(400_000, budget // (dim * eb)))
    if a.stream:
        n = max(20_000, min(1_000_000, int(rng.choice([4e8, 1.2e9, 2.5e9])) // (dim * eb)))
    nq = int(rng.choice([1, 3, 16, 17, 40, 64, 100, 128, 200, 256]))
    if a.wide:
        nq = int(rng.choice([16, 31, 33, 48, 64, 65, 100, 128]))
    k = int(rng.choice([1, 10, 10, 37, 100]))
    scale = float(rng.choice([1.0, 1.0, 30.0, 1e-3]))
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = (T["f32"] if typ == "sq8" else T[typ]), dim, M[metric]
    ix = VecSim.SQ8Index(p) if typ == "sq8" else VecSim.BFIndex(p)
    rows = vectors(typ, n, dim, scale)
    if rng.random() < 0.3 and typ not in ("i8", "u8"):   # a cluster of near-duplicates: ties and crowded thresholds
        rows[: n // 50] = rows[0]
    for r0 in range(0, n, 100_000):
        ix.add_vectors(rows[r0:r0 + 100_000], np.arange(r0, min(n, r0 + 100_000)))
    q = vectors(typ, nq, dim, scale)
    if a.verbose:
        print("shape", runs, typ, metric, "dim", dim, "n", n, "nq", nq, "k", k, "scale", scale, flush=True)
    ix.set_option("dense_pairs", 0)
    if a.stream:
        ix.set_option("stream_tau", 1)
        ix.set_option("stream_probe_tiles", int(rng.choice([16, 64, 512])))
        ix.set_option("stream_refresh", int(rng.choice([1, 2, 8, 32])))
        ix.set_option("stream_early", int(rng.choice([0, 0, 16])))
    ix.reset_stats()
    l1, d1 = ix.knn_query(q, k)
    kern = ix.stats()["scan_kernel"]
    if a.readers > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(a.readers) as pool:
            outs = list(pool.map(lambda t: [ix.knn_query(q, k) for _ in range(3)], range(a.readers)))
        for o in outs:
            for (lc, dc) in o:
                if not (np.array_equal(lc, l1) and np.array_equal(dc, d1, equal_nan=True)):
                    bad += 1
                    print("MISMATCH (concurrent readers)", typ, metric, "dim", dim, "n", n, "nq", nq, "k", k, "kernel", kern, flush=True)
                    break
    kernels[kern] = kernels.get(kern, 0) + 1
    ix.set_option("mfma", 0)
    nchk = min(nq, 6)
    l2, d2 = ix.knn_query(q[:nchk], k)
    runs += 1
    if not (np.array_equal(l1[:nchk], l2) and np.array_equal(d1[:nchk], d2, equal_nan=True)):
        bad += 1
        print("MISMATCH", typ, metric, "dim", dim, "n", n, "nq", nq, "k", k, "scale", scale, "kernel", kern, flush=True)
    del ix
print("runs %d mismatches %d kernels %s" % (runs, bad, kernels))
sys.exit(1 if bad else 0)
