#!/usr/bin/env python3
"""int8 Cosine 10 M x 1024, batch 128, top-100: 8-wave workgroups (two per CU) against the 16-wave 256-query workgroup"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim, synth  # noqa: E402
p = VecSim.BFParams()
p.type, p.dim, p.metric = VecSim.VecSimType_INT8, 1024, VecSim.VecSimMetric_Cosine
ix = VecSim.BFIndex(p)
ix.add_synthetic(10_000_000, 42)
for b in (64, 128, 256):
    q = synth.rows_i8(43, 0, b, 1024)
    ref = None
    for narrow in (1, 0):
        ix.set_option("lowp_narrow", narrow)
        r = ix.knn_query(q, 100)
        if ref is None:
            ref = r
        assert np.array_equal(ref[0], r[0]) and np.array_equal(ref[1], r[1])
        ix.reset_stats()
        t0 = time.perf_counter()
        for _ in range(5):
            ix.knn_query(q, 100)
        dt = (time.perf_counter() - t0) / 5
        st = ix.stats()
        print("batch %d lowp_narrow %d: %.2f ms per batch, scan kernel %.3f ms" % (b, narrow, dt * 1e3, st["scan_ms"] / st["scan_launches"]), flush=True)
