#!/usr/bin/env python3
"""smallest run of the streaming-threshold filter (option stream_tau): one index, one batch, compared with the plain filter"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vectorsimilarity_amd import VecSim
n, dim, nq, k = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000, 768, 64, 10
from vectorsimilarity_amd import synth
q = synth.rows_f32(48, 0, nq, dim)
p = VecSim.BFParams(); p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, dim, VecSim.VecSimMetric_L2
ix = VecSim.BFIndex(p); ix.add_synthetic(n, 47); ix.set_option("dense_pairs", 0)
a = ix.knn_query(q, k)
print("plain done", flush=True)
ix.set_option("stream_tau", 1)
for i in range(3):
    ix.reset_stats(); t0 = time.time(); b = ix.knn_query(q, k); st = ix.stats()
    print("stream pass", i, "same", bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])), "cand", st["candidates"] / nq, "fallbacks", st["fallbacks"], "retries", st["retries"], "%.1f ms" % ((time.time() - t0) * 1e3), flush=True)
