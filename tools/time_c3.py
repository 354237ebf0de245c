#!/usr/bin/env python3
"""Config-3 shape (int8 Cosine d=1024, batch 256, top-100): batch wall time against the probe size.
    python tools/time_c3.py [--rows 50000000] [--caps 8192,16384]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=50_000_000)
ap.add_argument("--caps", default="8192,16384")
ap.add_argument("--divs", default="32")
a = ap.parse_args()
p = VecSim.BFParams()
p.type, p.dim, p.metric = VecSim.VecSimType_INT8, 1024, VecSim.VecSimMetric_Cosine
ix = VecSim.BFIndex(p)
ix.add_synthetic(a.rows, 42)
q = synth.rows_i8(48, 0, 256, 1024)
base = ix.knn_query(q, 100)
for cap in [int(x) for x in a.caps.split(",")]:
    for div in [int(x) for x in a.divs.split(",")]:
        ix.set_option("probe_cap", cap)
        ix.set_option("probe_div", div)
        r = ix.knn_query(q, 100)
        assert (r[0] == base[0]).all() and (r[1] == base[1]).all()
        ix.reset_stats()
        best = None
        for _ in range(5):
            t0 = time.perf_counter()
            ix.knn_query(q, 100)
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        st = ix.stats()
        print("probe_cap %d probe_div %d: batch %.2f ms (best of 5), filter kernel %.2f ms, candidates/query %.0f"
              % (cap, div, best, st["scan_ms"] / max(1, st["scan_launches"]), st["candidates"] / (256.0 * max(1, st["scan_launches"]))), flush=True)
