#!/usr/bin/env python3
"""int8 Cosine d=1024, 256 queries: filter-kernel time on all-zero rows / queries against random ones (is the kernel bound by the
chip's power budget?  data-dependent switching power shows up as a clock difference):  python tools/time_i8_zero.py [--rows N]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=2_000_000)
ap.add_argument("--opts", default="lowp_x32=0,lowp_x32=1")
a = ap.parse_args()
for kind in ("random", "zero", "ones"):
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = VecSim.VecSimType_INT8, 1024, VecSim.VecSimMetric_L2
    ix = VecSim.BFIndex(p)
    if kind == "random":
        rows = synth.rows_i8(42, 0, a.rows, 1024)
        q = synth.rows_i8(48, 0, 256, 1024)
    else:
        v = 0 if kind == "zero" else 1
        rows = np.full((a.rows, 1024), v, dtype=np.int8)
        q = np.full((256, 1024), v, dtype=np.int8)
        rows[:, 0] = (np.arange(a.rows) % 100).astype(np.int8)   # (keeps the candidate lists short: scores differ)
    step = 250_000
    for i in range(0, a.rows, step):
        ix.add_vectors(rows[i:i + step], np.arange(i, min(i + step, a.rows)))
    for spec in a.opts.split(","):
        k, v = spec.split("=")
        ix.set_option(k, int(v))
        ix.knn_query(q, 10)
        ix.reset_stats()
        for _ in range(6):
            ix.knn_query(q, 10)
        st = ix.stats()
        kms = st["scan_ms"] / max(1, st["scan_launches"])
        print("%-7s %-14s %-24s kernel %.3f ms = %.0f GB/s, %.0f TOP/s" % (kind, spec, st["scan_kernel"], kms, a.rows * 1024 / kms / 1e6,
                                                                          2.0 * a.rows * 1024 * 256 / kms / 1e9), flush=True)
    del ix
