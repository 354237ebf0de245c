#!/usr/bin/env python3
"""Latency of small query batches on the config-2 table (10M x 768 fp32 L2): exact scan path vs MFMA filter path."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim, synth  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
p = VecSim.BFParams()
p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, dim, VecSim.VecSimMetric_L2
ix = VecSim.BFIndex(p)
ix.add_synthetic(rows, 47)
for nq in (1, 2, 4, 8, 16):
    q = synth.rows_f32(48, 0, nq, dim)
    ref = None
    for minq in (9, 1):
        ix.set_option("mfma_min_q", minq)
        ix.knn_query(q, 10)
        ix.reset_stats()
        t0 = time.perf_counter()
        for _ in range(5):
            l, d = ix.knn_query(q, 10)
        dt = (time.perf_counter() - t0) / 5
        st = ix.stats()
        if ref is None:
            ref = (l, d)
        same = np.array_equal(l, ref[0]) and np.array_equal(d, ref[1])
        print("rows %d dim %d nq %2d mfma_min_q %d: %.3f ms/call  kernel %s  identical %s" % (rows, dim, nq, minq, dt * 1e3, st["scan_kernel"], same), flush=True)
