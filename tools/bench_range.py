#!/usr/bin/env python3
"""Flat range query / batch-iterator first batch latency on ~6 GB tables."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim, synth  # noqa: E402

for dim in [int(x) for x in (sys.argv[1:] or ["768", "100"])]:
    n = int(6e9 / (dim * 4))
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, dim, VecSim.VecSimMetric_L2
    ix = VecSim.BFIndex(p)
    ix.add_synthetic(n, 47)
    q = synth.rows_f32(48, 0, 1, dim)
    l, d = ix.knn_query(q, 100)
    radius = float(d[0][-1])
    ix.range_query(q[0], radius)
    t0 = time.perf_counter()
    for _ in range(3):
        rl, rd = ix.range_query(q[0], radius)
    dt = (time.perf_counter() - t0) / 3
    st = ix.stats()
    t0 = time.perf_counter()
    it = ix.create_batch_iterator(q[0])
    bl, bd = it.get_next_results(100, VecSim.BY_SCORE)
    dtb = time.perf_counter() - t0
    print("dim %d rows %d: range (%d hits) %.3f ms [%s], iterator first batch %.3f ms, same as knn: %s %s" % (
        dim, n, rl.shape[1], dt * 1e3, st["scan_kernel"], dtb * 1e3, sorted(rl[0].tolist()) == sorted(l[0].tolist()),
        bl[0].tolist() == l[0].tolist()), flush=True)
    del ix
