#!/usr/bin/env python3
"""SQ8 Flat index at the headline shape (10 M x 768, top-10): batch time, distances/s and scan-kernel bandwidth over the
quarter-size rows, next to the fp32 index on the same vectors.
    python tools/bench_sq8.py [--rows 10000000] [--batches 64,128] [--metric L2]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--batches", default="64,128")
ap.add_argument("--metric", default="L2")
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--readers", type=int, default=1, help="threads submitting the batches (2: the steady state bench.py measures)")
ap.add_argument("--sweep", default="", help="NAME=V1,V2: repeat every MFMA measurement with this index option at each value")
a = ap.parse_args()
p = VecSim.BFParams()
p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, a.dim, getattr(VecSim, "VecSimMetric_" + a.metric)
ix = VecSim.SQ8Index(p)
for o in a.opt:
    ix.set_option(o.split("=")[0], int(o.split("=")[1]))
t0 = time.perf_counter()
for r0 in range(0, a.rows, 500_000):
    r1 = min(a.rows, r0 + 500_000)
    ix.add_vectors(synth.rows_f32(47, r0, r1 - r0, a.dim), np.arange(r0, r1))
print("ingest (host quantiser + upload): %.1f s" % (time.perf_counter() - t0), flush=True)
row_bytes = a.dim + (16 if a.metric == "L2" else 12)
for b in [int(x) for x in a.batches.split(",")]:
    qs = [synth.rows_f32(48 + i, 0, b, a.dim) for i in range(3)]
    sweep = [(a.sweep.split("=")[0], int(v)) for v in a.sweep.split("=")[1].split(",")] if a.sweep else [None]
    for mf, sw in [(1, x) for x in sweep] + ([(0, None)] if not a.sweep else []):
        if sw:
            ix.set_option(sw[0], sw[1])
            print("option %s = %d:" % sw, end=" ")
        ix.set_option("mfma", mf)
        ix.knn_query(qs[0], a.k)
        ix.reset_stats()
        steps = a.steps if mf else 2
        if a.readers > 1 and mf:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(a.readers) as pool:
                list(pool.map(lambda t: [ix.knn_query(qs[j % 3], a.k) for j in range(t, 2 * a.readers, a.readers)], range(a.readers)))   # lanes' scratch
                ix.reset_stats()
                t0 = time.perf_counter()
                list(pool.map(lambda t: [ix.knn_query(qs[j % 3], a.k) for j in range(t, steps, a.readers)], range(a.readers)))
                dt = (time.perf_counter() - t0) / steps
        else:
            t0 = time.perf_counter()
            for s in range(steps):
                ix.knn_query(qs[s % 3], a.k)
            dt = (time.perf_counter() - t0) / steps
        st = ix.stats()
        kms = st["scan_ms"] / max(1, st["scan_launches"])
        print("batch %d mfma %d: %.2f ms per batch = %.1f G distances/s; %s %.3f ms per launch, %d launches per batch = %.0f GB/s "
              "of SQ8 rows; candidates/query %.0f" % (b, mf, dt * 1e3, a.rows * b / dt / 1e9, st["scan_kernel"], kms,
                                                      st["scan_launches"] / steps, a.rows * row_bytes / kms / 1e6,
                                                      st["candidates"] / max(1, steps * b)), flush=True)
