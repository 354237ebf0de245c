#!/usr/bin/env python3
"""In-process A/B of the MFMA filter kernel's tuning variants (ring depth, cache policy, occupancy,
grid size) on the bench workload.  Interleaved rounds, reports mean/min HIP-event kernel time.
    python tools/tune_filter.py [--rows 10000000] [--rounds 4]"""
import argparse
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--variants", default="0,1,5,6,7,8,9")
ap.add_argument("--wgs", default="2,3,4")
a = ap.parse_args()

p = VecSim.BFParams()
p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, a.dim, VecSim.VecSimMetric_L2
ix = VecSim.BFIndex(p)
ix.add_synthetic(a.rows, 47)
q = synth.rows_f32(48, 0, a.batch * 4, a.dim).reshape(4, a.batch, a.dim)
res = {}
combos = list(itertools.product([int(v) for v in a.variants.split(",")], [int(w) for w in a.wgs.split(",")]))
base = [ix.knn_query(q[b], 10) for b in range(4)]
for r in range(a.rounds):
    for v, w in combos:
        ix.set_option("mfma_variant", v)
        ix.set_option("wg_per_cu", w)
        ix.reset_stats()
        for b in range(3):
            l, d = ix.knn_query(q[(r + b) % 4], 10)
            bl, bd = base[(r + b) % 4]
            assert np.array_equal(l, bl) and np.array_equal(d, bd), ("variant result differs", v, w)
        assert ix.stats()["fallbacks"] == 0
        st = ix.stats()
        res.setdefault((v, w), []).append(st["scan_ms"] / st["scan_launches"])
bytes_ = a.rows * a.dim * 4
print("variant wg/cu   mean_ms   min_ms   GB/s(min)")
for (v, w), ts in sorted(res.items(), key=lambda kv: min(kv[1])):
    print("%7d %5d %9.3f %8.3f %10.0f" % (v, w, np.mean(ts), min(ts), bytes_ / min(ts) / 1e6))
