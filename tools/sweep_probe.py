#!/usr/bin/env python3
"""Headline shape (fp32 L2, 10 M x 768, batch 64, top-10): batch wall time against the probe size."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim, synth  # noqa: E402

p = VecSim.BFParams()
p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, 768, VecSim.VecSimMetric_L2
ix = VecSim.BFIndex(p)
ix.add_synthetic(10_000_000, 47)
q = synth.rows_f32(48, 0, 64, 768)
base = ix.knn_query(q, 10)
for rnd in range(2):
    for div in (16, 24, 32, 48, 64, 96, 128, 192):
        ix.set_option("probe_div", div)
        r = ix.knn_query(q, 10)
        assert (r[0] == base[0]).all() and (r[1] == base[1]).all()
        ix.reset_stats()
        ts = []
        for _ in range(12):
            t0 = time.perf_counter()
            ix.knn_query(q, 10)
            ts.append((time.perf_counter() - t0) * 1e3)
        st = ix.stats()
        print("probe_div %3d: batch mean %.3f ms  min %.3f ms  candidates/query %.0f" %
              (div, sum(ts) / len(ts), min(ts), st["candidates"] / (64.0 * st["scan_launches"])), flush=True)
