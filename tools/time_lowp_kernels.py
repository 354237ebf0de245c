#!/usr/bin/env python3
"""Filter-kernel time of the int8 config-3 shape (Cosine d=1024, 256 queries, top-100) under the kernel options:
    python tools/time_lowp_kernels.py [--rows 10000000] [--opts lowp_ksplit=0,lowp_ksplit=1,lowp_ksplit=2]
Every option set must reproduce the first one's reply."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--opts", default="lowp_ksplit=0,lowp_ksplit=1,lowp_ksplit=2")
ap.add_argument("--reps", type=int, default=8)
a = ap.parse_args()
p = VecSim.BFParams()
p.type, p.dim, p.metric = VecSim.VecSimType_INT8, 1024, VecSim.VecSimMetric_Cosine
ix = VecSim.BFIndex(p)
ix.add_synthetic(a.rows, 42)
q = synth.rows_i8(48, 0, 256, 1024)
base = None
for spec in a.opts.split(","):
    for kv in spec.split("+"):
        k, v = kv.split("=")
        ix.set_option(k, int(v))
    r = ix.knn_query(q, 100)
    if base is None:
        base = r
    same = bool((r[0] == base[0]).all() and (r[1] == base[1]).all())
    ix.reset_stats()
    best = None
    for _ in range(a.reps):
        t0 = time.perf_counter()
        ix.knn_query(q, 100)
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None else min(best, dt)
    st = ix.stats()
    kms = st["scan_ms"] / max(1, st["scan_launches"])
    print("%-28s %-24s kernel %.3f ms = %.0f GB/s, %.0f TOP/s; batch %.2f ms (best of %d); same reply: %s" % (
        spec, st["scan_kernel"], kms, a.rows * 1028 / kms / 1e6, 2.0 * a.rows * 1024 * 256 / kms / 1e9, best, a.reps, same), flush=True)
